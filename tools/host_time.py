"""GPU box: host enqueue time per training step (batch 4: the GPU is never the bottleneck) for config 2 fp32 and config 5
fp32 / bf16."""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
import bench
for wl, dt, side, ncls in ((2, 'fp32', 32, 10), (5, 'fp32', 64, 20), (5, 'bf16', 64, 20)):
    net = bench.build_model(torch.device('cuda', 0), wl)
    net.set_compute_dtype(dt)
    x = torch.rand(4, 3, side, side, device='cuda'); y = torch.randint(0, ncls, (4,), device='cuda')
    m = None
    for i in range(10): _, m = net.train_step(x, y, batch=i, current_measures=m)
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(50): _, m = net.train_step(x, y, batch=i, current_measures=m)
    t1 = time.time()
    torch.cuda.synchronize()
    t2 = time.time()
    print(f'config {wl} {dt}: host enqueue {(t1 - t0) / 50 * 1e3:.2f} ms/step (then {1e3 * (t2 - t1):.1f} ms to drain)')
