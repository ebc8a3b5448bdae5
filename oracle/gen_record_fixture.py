#!/usr/bin/env python3
"""TEST INFRASTRUCTURE - build container only.  Lets the REFERENCE's LossRecorder (utils/save_load/recorders.py:13-400)
write tests/golden/record_ref/record-demo.pth (cut) and record-demo-uncut.pth from deterministic batches shaped like the
all-class evaluation's outputs ((C, N) losses, (N,) losses, logits.T, y_true; the last batch partial), and stores what its
own accessors return (get_batch, __getitem__, recorded_samples, merge) in expected.npz.  The drop-in recorder must read
these files, return the same values and write files with the same dictionary (tests/test_abi_and_host.py)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
from oracle.gen_golden import import_reference      # noqa: E402

C, B = 3, 4
SIZES = [4, 4, 4, 2]


def batch(i, n):
    g = torch.Generator().manual_seed(50 + i)
    return dict(total=torch.randn(C, n, generator=g), kl=torch.randn(C, n, generator=g), cross_x=torch.randn(n, generator=g),
                logits=torch.randn(C, n, generator=g), y_true=torch.randint(0, C, (n,), generator=g))


def main():
    import_reference()
    from utils.save_load import LossRecorder
    out = os.path.join(REPO, 'tests', 'golden', 'record_ref')
    os.makedirs(out, exist_ok=True)
    exp = {}
    r = LossRecorder(B, **batch(0, B))
    for i, n in enumerate(SIZES):
        r.append_batch(**batch(i, n))
    exp['len'] = np.int64(len(r))
    exp['recorded_samples'] = np.int64(r.recorded_samples)
    exp['num_batch_before_save'] = np.int64(r.num_batch)
    for k in r:
        exp['all.' + k] = r[k].numpy()
        exp['b3.' + k] = r.get_batch(3, k).numpy()
        exp['b1.' + k] = r.get_batch(1, k).numpy()
    r.save(os.path.join(out, 'record-demo-uncut.pth'), cut=False)
    r.save(os.path.join(out, 'record-demo.pth'))
    exp['num_batch_after_cut'] = np.int64(r.num_batch)
    other = LossRecorder(B, **batch(0, B))
    other.append_batch(**batch(7, 3))
    r2 = LossRecorder.load(os.path.join(out, 'record-demo.pth'))
    r2.merge(other)
    exp['merged.len'] = np.int64(len(r2))
    exp['merged.recorded_samples'] = np.int64(r2.recorded_samples)
    exp['merged.last_batch_size'] = np.int64(r2.last_batch_size)
    exp['merged.total'] = r2['total'].numpy()
    np.savez_compressed(os.path.join(out, 'expected.npz'), **exp)
    d = torch.load(os.path.join(out, 'record-demo.pth'), weights_only=False)
    print({k: (type(v).__name__, v if not isinstance(v, dict) else {a: tuple(b.shape) for a, b in v.items()}) for k, v in d.items()})


if __name__ == '__main__':
    main()
