"""GPU box: launch the split-bf16 ("x3") convolution of the largest layer (imager.15 forward, as the step launches it) a
few times.  Run under rocprofv3 --kernel-trace [--pmc ...] for its counters."""
import os, sys, torch
REPO = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import ops
spec = ops.ConvSpec(32, 32, 5, 1, 2, 0, transposed=True)
x = torch.randn(1024, 32, 32, 32, device='cuda')
w = torch.randn(32, 32, 5, 5, device='cuda') * 0.03
b = torch.zeros(32, device='cuda')
aff = (torch.rand(32, device='cuda') + 0.5, torch.randn(32, device='cuda') * 0.1, True)
for _ in range(8):
    y, _, _ = ops.conv_fwd_aff_raw(x, w, b, spec, aff, True)
torch.cuda.synchronize()
print('done', float(y[0, 0, 0, 0]))
