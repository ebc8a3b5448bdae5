#!/usr/bin/env python3
"""TEST INFRASTRUCTURE - runs ONLY in the build container.  Records how the REFERENCE's Sigma (module/vae_layers/
layers.py:73-213) prints, lists its parameters and moves under update() for a set of constructor arguments, as DATA:
tests/golden/sigma_forms.json.  These strings end up in job-directory names and train_params.json, so the drop-in class
must reproduce them exactly (tests/test_abi_and_host.py::test_sigma_matches_reference_forms)."""
import json
import os
import sys
import warnings

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
from oracle.gen_golden import import_reference      # noqa: E402

CASES = [dict(value=0.5), dict(value=1.0, learned=True), dict(is_rmse=True),
         dict(value=1.0, decay=0.1, reach=2, max_step=0.05), dict(value=2.0, decay=0.3), dict(input_dim=[3, 32, 32]),
         dict(input_dim=[3, 32, 32], sdim=[3, 32, 32]), dict(value=0.7, sdim=[3, 4, 4], learned=True),
         dict(value=0.3, is_log=True), dict(value=1.5, sigma0=3.0, decay=0.2, reach=0.5)]
RMSES = [0.8, 0.2, 3.0]


def snapshot(s):
    p = s.params
    return {'str': str(s), 'repr': repr(s), 'formats': {f: format(s, f) for f in ('', 'g', '.3f', 'i', 'x')},
            'param_keys': list(p), 'params': {k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in p.items()
                                              if not (isinstance(v, float) and v != v)},
            'data': s.data.flatten()[:4].tolist(), 'shape': list(s.shape), 'requires_grad': s.requires_grad}


def main():
    warnings.simplefilter('ignore')
    import_reference()
    from module.vae_layers.layers import Sigma
    out = []
    for kw in CASES:
        k = {a: (tuple(b) if isinstance(b, list) else b) for a, b in kw.items()}
        s = Sigma(**k)
        rec = {'kwargs': kw, 'initial': snapshot(s), 'after': []}
        if s.coded:
            v = torch.linspace(-1, 1, 5 * int(torch.tensor(s.output_dim).prod())).reshape(5, *s.output_dim)
            s.update(v=v)
            rec['after'].append(snapshot(s))
        else:
            for r in RMSES:
                s.update(rmse=torch.tensor(r))
                rec['after'].append(snapshot(s))
        out.append(rec)
    path = os.path.join(REPO, 'tests', 'golden', 'sigma_forms.json')
    json.dump({'rmses': RMSES, 'cases': out}, open(path, 'w'), indent=1)
    print('wrote', path, len(out), 'cases')


if __name__ == '__main__':
    main()
