"""Diagnostic (not collected): per-tensor gradient distances of the HIP model at the full-size workloads against the
reference's fp32 and fp64 gradients held by the compact goldens.  python tests/diagnostics/full_grad_diag.py [case ...]"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
for p in (REPO, os.path.join(REPO, 'joint-vae_amd')):
    sys.path.insert(0, p)
from oracle.cases import FULL_CASES, get_case          # noqa: E402
from oracle.det_init import det_inputs, load_det_state  # noqa: E402
from cvae import ClassificationVariationalNetwork as Net   # noqa: E402

for name in sys.argv[1:] or list(FULL_CASES):
    g = np.load(os.path.join(REPO, 'tests', 'golden', name + '.npz'))
    case = get_case(name)
    kw = case['net']
    net = Net(**kw)
    load_det_state(net, seed=0)
    net.to('cuda').train()
    x, y, eps = (t.cuda() for t in det_inputs(case['N'], kw['input_shape'], kw['num_labels'], 1, kw['latent_dim']))
    net.optimizer.zero_grad()
    out = net.evaluate(x, y, with_beta=True, kl_var_weighting=case['kl_var_weighting'], epsilon=eps)
    out[2]['total'].mean().backward()
    tot = float(g['total_grad_norm'])
    print(f'== {name}: |g| ref32 {tot:.4f} ref64 {float(g["total_grad_norm64"]):.4f}')
    print(f'{"tensor":34s} {"norm ours/ref32-1":>18s} {"ours/ref64-1":>13s} {"ref32/ref64-1":>13s} {"L2 ours-64":>11s} {"L2 ref32-64":>11s} {"L2 ours-32":>11s}')
    for k in g['grad_names']:
        mine = net.get_parameter(k).grad.detach().double().cpu().numpy()
        r32, r64 = float(g['gnorm.' + k]), float(g['gnorm64.' + k])
        n = np.linalg.norm(mine)
        line = f'{k:34s} {n / r32 - 1:18.2e} {n / r64 - 1:13.2e} {r32 / r64 - 1:13.2e}'
        if 'grad64.' + k in g.files:
            g64 = g['grad64.' + k].astype(np.float64)
            g32 = g['grad.' + k].astype(np.float64)
            line += f' {np.linalg.norm(mine - g64) / r64:11.2e} {np.linalg.norm(g32 - g64) / r64:11.2e} {np.linalg.norm(mine - g32) / r64:11.2e}'
        print(line)
