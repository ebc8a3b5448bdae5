"""TEST INFRASTRUCTURE - writes tests/golden/c5_n256_bf16emu.npz: one training step of BASELINE configs[4]'s geometry at its
per-rank batch (N = 256) computed by the ORACLE in its bf16-emulating mode (oracle.jvae_oracle.bf16_convs: bf16 operands and
stored activations / activation gradients around the 5x5 convolutions, fp32 everywhere else - the arithmetic of the product's
`set_compute_dtype('bf16')`; the reference has no such mode, so this fixture is oracle data, the fp32 reference golden
c5_n256.npz stays the pin of the oracle itself).  Deterministic weights / inputs as everywhere (oracle/det_init.py).

    python oracle/gen_bf16_golden.py        (about two minutes on 8 CPU threads)
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import jvae_oracle as O                      # noqa: E402
from oracle.cases import get_case                        # noqa: E402
from oracle.det_init import det_inputs                   # noqa: E402


def main():
    case = get_case('c5_n256')
    kw, N = case['net'], case['N']
    sp = O.make_spec(**kw)
    x, y, eps = det_inputs(N, kw['input_shape'], kw['num_labels'], 1, kw['latent_dim'])
    out = {}
    for mode in ('bf16', 'fp32'):
        P = O.init_state(sp, seed=0)
        if mode == 'bf16':
            with O.bf16_convs():
                o, grads, gn = O.train_step(sp, P, O.AdamState(sp), x, y, eps, kl_var_weighting=case['kl_var_weighting'],
                                            gamma_weighting=case['gamma_weighting'])
        else:
            o, grads, gn = O.train_step(sp, P, O.AdamState(sp), x, y, eps, kl_var_weighting=case['kl_var_weighting'],
                                        gamma_weighting=case['gamma_weighting'])
        sfx = '' if mode == 'bf16' else '.fp32'
        for k, v in o[2].items():
            out['loss.' + k + sfx] = v.detach().numpy().astype(np.float32)
        out['mu' + sfx] = o[4].detach().numpy().astype(np.float32)
        out['log_var' + sfx] = o[5].detach().numpy().astype(np.float32)
        xr = o[0].detach().double().flatten(2)
        out['x_reco_mean' + sfx] = xr.mean(-1).numpy()
        out['x_reco_norm' + sfx] = xr.norm(dim=-1).numpy()
        out['total_grad_norm' + sfx] = np.float64(gn)
        for n_, g in grads.items():
            out['gnorm.' + n_ + sfx] = np.float64(g.double().norm())
            if mode == 'bf16' and g.numel() <= 8192:
                out['grad.' + n_] = g.detach().numpy().astype(np.float32)
    out['grad_names'] = np.array(sorted(grads))
    dst = os.path.join(REPO, 'tests', 'golden', 'c5_n256_bf16emu.npz')
    np.savez_compressed(dst, **out)
    print('wrote', dst, os.path.getsize(dst), 'bytes')


if __name__ == '__main__':
    main()
