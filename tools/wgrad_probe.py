import os, sys, torch
REPO = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import ops
spec = ops.ConvSpec(32, 32, 5, 1, 2, 0, transposed=True)
x = torch.randn(1024, 32, 32, 32, device='cuda'); gy = torch.randn(1024, 32, 32, 32, device='cuda')
gw = torch.zeros(32, 32, 5, 5, device='cuda')
for _ in range(6): ops.conv_wgrad_raw(x, gy, spec, gw.shape, False, gw, None)
torch.cuda.synchronize(); print('ok')
