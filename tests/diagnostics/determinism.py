"""GPU box: two identical models, same batch: how far apart are gradients / parameters after 1 and 2 steps?"""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from oracle.cases import get_case
from oracle.det_init import det_inputs, load_det_state
from cvae import ClassificationVariationalNetwork as Net
from jvae_hip import ops
if len(sys.argv) > 1 and sys.argv[1] == 'nooverlap':
    ops.OVERLAP_WGRAD = False
case = get_case('c2_n8'); kw = case['net']
def build():
    n = Net(**kw); load_det_state(n, 0); n.to('cuda'); n.train(); return n
a, b = build(), build()
x, y, eps = (t.cuda() for t in det_inputs(8, kw['input_shape'], 10, 1, 64))
for step in range(2):
    for n in (a, b):
        n.optimizer.zero_grad()
        out = n.evaluate(x, y, with_beta=True, epsilon=eps)
        out[2]['total'].mean().backward()
    torch.cuda.synchronize()
    worst = max(((float((p.grad - q.grad).abs().max() / (q.grad.abs().max() + 1e-30)), k) for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()) if p.grad is not None), key=lambda t: t[0])
    print('step', step, 'worst grad rel diff', worst)
    if step == 0:
        for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
            if p.grad is not None:
                d = float((p.grad - q.grad).abs().max() / (q.grad.abs().max() + 1e-30))
                if d > 1e-5: print('   ', k, d)
    for n in (a, b):
        n.optimizer.clip(n.parameters()); n.optimizer.step()
    torch.cuda.synchronize()
    worst = max(((float((p - q).abs().max() / (q.abs().max() + 1e-30)), k) for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters())), key=lambda t: t[0])
    print('step', step, 'worst param rel diff', worst)
