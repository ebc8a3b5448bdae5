"""GPU box: where does the HOST time of one training step go (tiny batch, so the GPU is never the bottleneck)."""
import cProfile, pstats, os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
import bench
net = bench.build_model(torch.device('cuda', 0))
x = torch.rand(8, 3, 32, 32, device='cuda'); y = torch.randint(0, 10, (8,), device='cuda')
m = None
for i in range(10): _, m = net.train_step(x, y, batch=i, current_measures=m)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(50): _, m = net.train_step(x, y, batch=i, current_measures=m)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(28)
