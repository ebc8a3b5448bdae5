"""Debug helper (GPU box): layer-by-layer forward / backward error of the HIP model vs the fp64 CPU oracle."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from oracle import jvae_oracle as O
from oracle.cases import get_case
from oracle.det_init import det_inputs, load_det_state
from cvae import ClassificationVariationalNetwork as Net
name = sys.argv[1] if len(sys.argv) > 1 else 'c2_n8'
case = get_case(name); kw = case['net']
net = Net(**kw); load_det_state(net, 0); net.to('cuda'); net.train()
x, y, eps = det_inputs(case['N'], kw['input_shape'], kw['num_labels'], net.latent_sampling, kw['latent_dim'])
tape_d = {}
def hook(nm):
    def f(mod, inp, out):
        if out.requires_grad:
            out.retain_grad()
        tape_d[nm] = out
    return f
for pre in ('features', 'imager'):
    for i, m in enumerate(getattr(net, pre)):
        m.register_forward_hook(hook(f'{pre}.{i}'))
net.optimizer.zero_grad()
out = net.evaluate(x.cuda(), y.cuda(), with_beta=True, epsilon=eps.cuda())
out[2]['total'].mean().backward()
sp = O.make_spec(**kw)
P = O.init_state(sp)
P = {k: (v.detach().double().requires_grad_(v.requires_grad) if v.dtype.is_floating_point else v) for k, v in P.items()}
O.TAPE = []
o = O.evaluate(sp, P, x.double(), y, eps.double(), 1.0, 1.0)
o[2]['total'].mean().backward()
def r(a, b):
    return float((a.detach().double().cpu() - b.detach()).abs().max() / b.detach().abs().max())
for nm, t in O.TAPE:
    d = tape_d.get(nm)
    if d is None:
        print(f'{nm:14s} (fused on the GPU side)'); continue
    print(f'{nm:14s} fwd {r(d, t):.1e}   grad {r(d.grad, t.grad):.1e}' if (d.grad is not None and t.grad is not None) else f'{nm:14s} fwd {r(d, t):.1e}')
