"""GPU box: launch the weight gradient of one config-2 layer a few times as the training step launches it (deferred
BatchNorm on the layer input, accumulation into an existing gradient).  LAYER = D5 (default) | D3 | D1 | E2 | E1 | D4.
Run under rocprofv3 (--kernel-trace --stats, then separate --pmc passes); JVAE_X3=0 selects the fp32-MFMA kernel."""
import os, sys, torch
REPO = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import ops
LAYERS = {'D5': (1024, 32, 32, 1, True, 32), 'D3': (1024, 64, 32, 1, True, 16), 'D1': (1024, 64, 64, 1, True, 8),
          'E2': (512, 32, 64, 1, False, 16), 'E1': (512, 32, 32, 2, False, 32), 'D4': (1024, 32, 32, 2, True, 16)}
N, cin, cout, s, tr, H = LAYERS[os.environ.get('LAYER', 'D5')]
spec = ops.ConvSpec(cin, cout, 5, s, 2, 1 if (tr and s == 2) else 0, transposed=tr)
x = torch.randn(N, cin, H, H, device='cuda')
y = ops.conv_fwd_raw(x, torch.zeros((cin, cout, 5, 5) if tr else (cout, cin, 5, 5), device='cuda'), torch.zeros(cout, device='cuda'), spec)
gy = torch.randn_like(y)
gw = torch.zeros((cin, cout, 5, 5) if tr else (cout, cin, 5, 5), device='cuda')
aff = (torch.rand(cin, device='cuda') + 0.5, torch.randn(cin, device='cuda') * 0.1, True)
for _ in range(8):
    ops.conv_wgrad_raw(x, gy, spec, gw.shape, False, gw, None, aff=aff)
torch.cuda.synchronize()
print('ok', float(gw.abs().max()))
