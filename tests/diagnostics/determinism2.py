"""GPU box: is the first forward/backward different from later ones (same model, same batch)?"""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from oracle.cases import get_case
from oracle.det_init import det_inputs, load_det_state
from cvae import ClassificationVariationalNetwork as Net
case = get_case('c2_n8'); kw = case['net']
n = Net(**kw); load_det_state(n, 0); n.to('cuda'); n.train()
x, y, eps = (t.cuda() for t in det_inputs(8, kw['input_shape'], 10, 1, 64))
runs = []
for it in range(4):
    n.optimizer.zero_grad()
    out = n.evaluate(x, y, with_beta=True, epsilon=eps)
    out[2]['total'].mean().backward()
    torch.cuda.synchronize()
    runs.append((out[0].detach().clone(), {k: p.grad.clone() for k, p in n.named_parameters() if p.grad is not None},
                 {k: b.clone() for k, b in n.named_buffers()}))
for it in range(1, 4):
    dx = float((runs[it][0] - runs[0][0]).abs().max())
    worst = max((float((runs[it][1][k] - runs[0][1][k]).abs().max() / (runs[0][1][k].abs().max() + 1e-30)), k) for k in runs[0][1])
    print('run', it, 'vs 0: x_reco max abs diff', dx, 'worst grad', worst)
dx = float((runs[3][0] - runs[2][0]).abs().max())
print('run 3 vs 2: x_reco', dx)

# ---- which layer's dL/dz first differs between an even and an odd run?
from module.vae_layers.conv import HipConv2d, HipConvTranspose2d
keep = {}
def hook(name):
    def f(mod, inp, out):
        out.retain_grad(); keep[name] = out
    return f
for name, m in n.named_modules():
    if isinstance(m, (HipConv2d, HipConvTranspose2d)):
        m.register_forward_hook(hook(name))
snaps = []
for it in range(2):
    keep.clear()
    n.optimizer.zero_grad()
    out = n.evaluate(x, y, with_beta=True, epsilon=eps)
    out[2]['total'].mean().backward()
    torch.cuda.synchronize()
    snaps.append({k: (v.detach().clone(), v.grad.clone()) for k, v in keep.items()})
for k in snaps[0]:
    z0, g0 = snaps[0][k]; z1, g1 = snaps[1][k]
    print('%-12s z rel diff %.2e   dL/dz rel diff %.2e' % (k, float((z0 - z1).abs().max() / z0.abs().max()), float((g0 - g1).abs().max() / g0.abs().max())))
