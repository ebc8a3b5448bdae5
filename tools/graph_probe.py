"""GPU box: replay the HIP-graph-captured training step (config 2, bs=512) a few times - the probe for a rocprofv3 kernel
trace of the graph path (tools/prof_graph.sh); MODE=eager runs the eager step instead."""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
import bench
dev = torch.device('cuda', 0)
torch.manual_seed(0)
net = bench.build_model(dev, 2)
x = torch.rand(512, 3, 32, 32, device=dev); y = torch.randint(0, 10, (512,), device=dev)
n = int(os.environ.get('STEPS', 25))
if os.environ.get('MODE', 'graph') == 'graph':
    step = net.graph_train_step(x, y)
    f = lambda i: step(x, y)
else:
    m = [None]
    def f(i):
        _, m[0] = net.train_step(x, y, batch=i, current_measures=m[0])
for i in range(5): f(i)
torch.cuda.synchronize(); t0 = time.time()
for i in range(n): f(i)
torch.cuda.synchronize()
print('%s: %.3f ms/step' % (os.environ.get('MODE', 'graph'), (time.time() - t0) / n * 1e3))
