"""GPU box: which aten ops (= torch-launched kernels, not ours) does one training step contain?  torch.profiler over 3 steps."""
import os, sys, torch
REPO = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
import bench
from torch.profiler import profile, ProfilerActivity
net = bench.build_model(torch.device('cuda', 0), 2)
x = torch.rand(512, 3, 32, 32, device='cuda'); y = torch.randint(0, 10, (512,), device='cuda')
m = None
for i in range(5):
    _, m = net.train_step(x, y, batch=i, current_measures=m)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(3):
        _, m = net.train_step(x, y, batch=5 + i, current_measures=m)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=6)
for e in sorted(ka, key=lambda e: -e.self_device_time_total):
    if e.self_device_time_total <= 0 or not (e.key.startswith('aten::') or 'Memcpy' in e.key or 'Memset' in e.key):
        continue
    st = [s_.split('joint-vae_amd/')[-1] for s_ in (e.stack or []) if 'joint-vae_amd' in s_][:3]
    print(f'{e.count / 3:5.1f}/step  {e.self_device_time_total / 3:7.1f} us/step  {e.key:28s}', ' <- '.join(st))
