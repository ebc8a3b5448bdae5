"""GPU box: launch the bf16-mode forward of the largest config-5 layer (ConvT 32 -> 32, 5x5, 512 x 64 x 64 maps, B8 operands, deferred
BatchNorm on the input, BatchNorm sums in the epilogue) a few times.  Probe for tools/prof_kernel.sh."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import ops, ops_b8
N, C, H = 512, 32, 64
spec = ops.ConvSpec(C, C, 5, 1, 2, 0, True)
x = ops_b8.pack(torch.randn(N, C, H, H, device='cuda'))
w = torch.randn(C, C, 5, 5, device='cuda') * 0.03; b = torch.zeros(C, device='cuda')
coef = torch.zeros(2, 32, device='cuda'); coef[0] = torch.rand(32, device='cuda') + 0.5; coef[1] = torch.randn(32, device='cuda') * 0.1
aff = (coef[0], coef[1], True)
for _ in range(10):
    y, st, ns = ops_b8.conv_fwd_raw(x, w, b, spec, want_stats=True, aff=aff)
torch.cuda.synchronize()
print('done', y.shape, ns)
