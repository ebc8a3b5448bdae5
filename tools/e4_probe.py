"""GPU box: the 7x7 head of conv32 (features.12: 64 -> 200 channels on 8x8 maps, 2x2 outputs) in its three directions, a few times each,
for rocprofv3 (tools/prof_kernel.sh TAG gemm_x3_kernel tools/e4_probe.py): the unfold / K-sliced product / fold family of DESIGN.md."""
import os, sys, torch
REPO = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import ops
N, cin, cout, k, H = 512, 64, 200, 7, 8
spec = ops.ConvSpec(cin, cout, k, 1, 0, 0, False)
x = torch.randn(N, cin, H, H, device='cuda')
w = torch.randn(cout, cin, k, k, device='cuda') * 0.05
b = torch.zeros(cout, device='cuda')
y = ops.conv_fwd_raw(x, w, b, spec)
gy = torch.randn_like(y)
for _ in range(6):
    ops.conv_fwd_stats_raw(x, w, b, spec)
    ops.conv_dgrad_raw(gy, w, spec, x.shape)
    ops.conv_wgrad_raw(x, gy, spec, w.shape, False)
torch.cuda.synchronize()
print('ok')
