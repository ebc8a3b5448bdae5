"""GPU box: one bf16 training step of the config-5 geometry; dumps every gradient.  Run twice (JVAE_WGRAD_B8X=0 / 1) and
compare: the two weight-gradient kernels must agree to fp32 summation order."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from oracle.cases import get_case
from oracle.det_init import load_det_state
from cvae import ClassificationVariationalNetwork as Net
kw = get_case('c5_n4')['net']
net = Net(**kw); load_det_state(net, seed=0); net.to('cuda').train(); net.set_compute_dtype('bf16')
torch.manual_seed(1)
x = torch.rand(32, *kw['input_shape'], device='cuda'); y = torch.randint(0, kw['num_labels'], (32,), device='cuda')
torch.manual_seed(7); torch.cuda.manual_seed(7)
net.optimizer.zero_grad()
_, _, losses, _ = net.evaluate(x, y, with_beta=True)
losses['total'].mean().backward()
torch.cuda.synchronize()
out = {n: p.grad.detach().cpu().clone() for n, p in net.named_parameters() if p.grad is not None}
path = sys.argv[1]
if os.path.exists(path):
    ref = torch.load(path)
    worst = 0
    for k, v in out.items():
        d = float((v.double() - ref[k].double()).norm() / ref[k].double().norm().clamp_min(1e-30))
        worst = max(worst, d)
        if d > 1e-5: print('DIFF', k, tuple(v.shape), d)
    print('worst relative L2 difference over %d tensors: %.3e' % (len(out), worst))
else:
    torch.save(out, path); print('saved', len(out))
