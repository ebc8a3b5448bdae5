"""GPU box: launch the dominant kernel (imager.15 forward: ConvTranspose2d 32->32 5x5 on 1024x32x32x32) a few times.
Run under rocprofv3 (--kernel-trace --stats, then --pmc FETCH_SIZE, then --pmc WRITE_SIZE in separate passes)."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import ops
spec = ops.ConvSpec(32, 32, 5, 1, 2, 0, transposed=True)
x = torch.randn(1024, 32, 32, 32, device='cuda')
w = torch.randn(32, 32, 5, 5, device='cuda') * 0.03
b = torch.zeros(32, device='cuda')
aff = (torch.rand(32, device='cuda') + 0.5, torch.randn(32, device='cuda') * 0.1, True)      # as in the training step
for _ in range(10):
    y, _, _ = ops.conv_fwd_aff_raw(x, w, b, spec, aff, True)
torch.cuda.synchronize()
print('done', float(y[0, 0, 0, 0]))
