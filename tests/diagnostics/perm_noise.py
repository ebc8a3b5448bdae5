"""GPU box: how far do the per-sample losses of config 2 at N = 512 move when the batch is permuted (BatchNorm partial sums are
taken per tile in fp32, so a permutation changes their rounding) - test_full_batch_properties (2) - for three permutations
(round 5: the same spread with the first 4-phase kernel and with the present one, profiles/NOTES.md)."""
import os, sys, torch
REPO = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from oracle.cases import full_config
from oracle.det_init import det_inputs, load_det_state
from cvae import ClassificationVariationalNetwork as Net
case = full_config(2, 512)
kw = case['net']
net = Net(**kw); load_det_state(net, seed=0); net.to('cuda'); net.train()
x, y, eps = (t.cuda() for t in det_inputs(512, kw['input_shape'], 10, 1, 64, seed=7))
_, _, l1, _ = net.evaluate(x, y, with_beta=True, epsilon=eps)
for s in range(3):
    perm = torch.randperm(512, device='cuda', generator=torch.Generator(device='cuda').manual_seed(s))
    _, _, l2, _ = net.evaluate(x[perm], y[perm], with_beta=True, epsilon=eps[:, perm])
    out = []
    for k in ('total', 'cross_x', 'kl', 'wmse'):
        a, b = l2[k].double(), l1[k][perm].double()
        d = (a - b).abs()
        out.append('%s max|d| %.3g (rel to max %.2e, median rel %.1e)' % (k, float(d.max()), float(d.max() / b.abs().max()), float((d / b.abs()).median())))
    print('perm %d:' % s, '; '.join(out))
