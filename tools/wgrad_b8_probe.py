"""GPU box: launch the bf16-mode weight gradient of the largest config-5 layer (32 -> 32, 5x5, 512 x 64 x 64 maps, B8 operands) a
few times.  Probe for tools/prof_kernel.sh."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import ops, ops_b8
N, C, H = 512, 32, 64
spec = ops.ConvSpec(C, C, 5, 1, 2, 0, True)
x = ops_b8.pack(torch.randn(N, C, H, H, device='cuda')); gy = ops_b8.pack(torch.randn(N, C, H, H, device='cuda'))
for _ in range(10):
    gw, _ = ops_b8.conv_wgrad_raw(x, gy, spec, (C, C, 5, 5), False)
torch.cuda.synchronize()
print('done', float(gw[0, 0, 0, 0]))
