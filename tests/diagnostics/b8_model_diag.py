"""GPU box: config-5 geometry model, fp32 vs bf16 compute on the same weights / batch / epsilon: per-sample losses, global
gradient, and a short training run."""
import os, sys, copy, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from oracle.cases import get_case
from oracle.det_init import det_inputs, load_det_state
from cvae import ClassificationVariationalNetwork as Net
N = int(os.environ.get('N', 32))
case = get_case('c5_n4'); kw = case['net']
def build(dtype):
    net = Net(**kw); load_det_state(net, seed=0); net.to('cuda'); net.train(); net.set_compute_dtype(dtype); return net
x, y, eps = det_inputs(N, kw['input_shape'], kw['num_labels'], 1, kw['latent_dim'])
x, y, eps = x.cuda(), y.cuda(), eps.cuda()
out = {}
for dt in ('fp32', 'bf16'):
    net = build(dt)
    net.optimizer.zero_grad()
    _, _, losses, meas = net.evaluate(x, y, batch=0, with_beta=True, epsilon=eps)
    losses['total'].mean().backward()
    g = torch.cat([p.grad.flatten() for p in net.parameters() if p.grad is not None])
    out[dt] = ({k: v.detach().clone() for k, v in losses.items()}, g.clone(), dict(meas))
for k in out['fp32'][0]:
    a, b = out['fp32'][0][k], out['bf16'][0][k]
    if float(a.abs().max()) == 0: continue
    print(f'{k:10s} fp32 mean {float(a.mean()):12.5f} bf16 mean {float(b.mean()):12.5f} max rel {float(((a-b).abs()/a.abs().clamp_min(1e-6)).max()):.3e}')
ga, gb = out['fp32'][1], out['bf16'][1]
print('grad: |fp32| %.4f |bf16| %.4f rel L2 %.4f cos %.5f' % (float(ga.norm()), float(gb.norm()), float((ga-gb).norm()/ga.norm()), float((ga*gb).sum()/(ga.norm()*gb.norm()))))
# short training: same data, fresh eps per step from a seeded generator
torch.manual_seed(1)
data = torch.rand(8, N, *kw['input_shape'], device='cuda'); lab = torch.randint(0, kw['num_labels'], (8, N), device='cuda')
for dt in ('fp32', 'bf16'):
    net = build(dt); torch.manual_seed(7); hist = []
    for step in range(40):
        out_ = net.train_step(data[step % 8], lab[step % 8])
        hist.append(float(out_[0]['total'].mean()) if isinstance(out_, tuple) else float(out_['total'].mean()))
    print(dt, 'loss every 8 steps:', ' '.join('%.1f' % h for h in hist[::8]), 'last %.2f' % hist[-1])
