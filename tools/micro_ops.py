"""GPU box: which Python lines launch the torch micro-kernels (fills, copies, adds, any/abs ...) of one training step of config 2.
torch.profiler with stacks over one eager train_step(); prints every aten op that launched a kernel, with the innermost repo frame."""
import os, sys, torch
REPO = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from oracle.cases import full_config
from cvae import ClassificationVariationalNetwork as Net
from torch.profiler import profile, ProfilerActivity
case = full_config(2, 512)
net = Net(**case['net']).to('cuda').train()
x = torch.rand(512, 3, 32, 32, device='cuda'); y = torch.randint(0, 10, (512,), device='cuda')
for i in range(4):
    net.train_step(x, y, batch=i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    net.train_step(x, y, batch=5)
    torch.cuda.synchronize()
rows = []
for e in prof.events():
    if e.device_type.name != 'CPU' or not e.name.startswith('aten::') or not e.kernels:
        continue
    frame = next((f for f in e.stack if REPO in f and 'tools/micro_ops' not in f), e.stack[0] if e.stack else '?')
    rows.append((e.time_range.start, e.name, [k.name[:60] for k in e.kernels], frame.replace(REPO + '/', '')))
for _, name, ks, frame in sorted(rows):
    print(f'{name:28s} {ks[0]:62s} {frame}')
print(len(rows), 'aten ops with kernels')
