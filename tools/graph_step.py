"""GPU box: the HIP-graph-captured training step against the eager one (same weights, same batches): parameters after a
few steps, then throughput of both."""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
import bench
wl = int(os.environ.get('WL', 2)); dt = os.environ.get('DT', 'fp32')
side, ncls, B = (64, 20, 256) if wl == 5 else (32, 10, 512)
dev = torch.device('cuda', 0)
def build():
    torch.manual_seed(0)
    n = bench.build_model(dev, wl); n.set_compute_dtype(dt); return n
x = torch.rand(B, 3, side, side, device=dev); y = torch.randint(0, ncls, (B,), device=dev)
a = build()
step = a.graph_train_step(x, y)
t_g = []
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(30): losses, meas = step(x, y)
    torch.cuda.synchronize(); t_g.append((time.time() - t0) / 30 * 1e3)
print('graph  ms/step', ['%.3f' % t for t in t_g], 'loss', float(losses['total'].mean()), 'rmse', meas['rmse'], 'opt step', a.optimizer._groups[0].step, 'dev step', float(a.optimizer._groups[0].hyper[3]))
b = build()
m = None
for i in range(5): _, m = b.train_step(x, y, batch=i, current_measures=m)
t_e = []
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(30): l2, m = b.train_step(x, y, batch=i, current_measures=m)
    torch.cuda.synchronize(); t_e.append((time.time() - t0) / 30 * 1e3)
print('eager  ms/step', ['%.3f' % t for t in t_e], 'loss', float(l2['total'].mean()))
