"""GPU box: one direction of one config-2 layer a few times, as the step launches it (deferred BatchNorm on the input of the forward
where the layer has one), for rocprofv3 / tools/prof_kernel.sh.  LAYER = E0 ... D6 (tools/conv_bench.py), DIR = fwd | dgrad | wgrad."""
import os, sys, torch
REPO = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import ops
LAYERS = {  # N, cin, cout, k, s, p, op, transposed, H
    'E0': (512, 3, 32, 5, 1, 2, 0, False, 32), 'E1': (512, 32, 32, 5, 2, 2, 0, False, 32), 'E2': (512, 32, 64, 5, 1, 2, 0, False, 16),
    'E3': (512, 64, 64, 5, 2, 2, 0, False, 16), 'E4': (512, 64, 200, 7, 1, 0, 0, False, 8), 'D0': (1024, 64, 64, 8, 1, 0, 0, True, 1),
    'D1': (1024, 64, 64, 5, 1, 2, 0, True, 8), 'D2': (1024, 64, 64, 5, 2, 2, 1, True, 8), 'D3': (1024, 64, 32, 5, 1, 2, 0, True, 16),
    'D4': (1024, 32, 32, 5, 2, 2, 1, True, 16), 'D5': (1024, 32, 32, 5, 1, 2, 0, True, 32), 'D6': (1024, 32, 3, 5, 1, 2, 0, False, 32)}
N, cin, cout, k, s, p, op, tr, H = LAYERS[os.environ.get('LAYER', 'D2')]
d = os.environ.get('DIR', 'fwd')
spec = ops.ConvSpec(cin, cout, k, s, p, op, tr)
x = torch.randn(N, cin, H, H, device='cuda')
w = torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), device='cuda') * 0.05
b = torch.zeros(cout, device='cuda')
y = ops.conv_fwd_raw(x, w, b, spec)
gy = torch.randn_like(y)
aff = None
if ops.conv_affine_ok(spec, N, H, H):
    aff = (torch.rand(cin, device='cuda') + 0.5, torch.randn(cin, device='cuda'), True)
for _ in range(8):
    if d == 'fwd':
        ops.conv_fwd_aff_raw(x, w, b, spec, aff, True) if aff is not None else ops.conv_fwd_stats_raw(x, w, b, spec)
    elif d == 'dgrad':
        ops.conv_dgrad_raw(gy, w, spec, x.shape)
    else:
        ops.conv_wgrad_raw(x, gy, spec, w.shape, False, aff=aff)
torch.cuda.synchronize()
print('ok')
