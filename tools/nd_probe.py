"""GPU box: run-to-run bit equality of the FIRST 4-phase kernel's forward with a deferred BatchNorm (JVAE_T2_V1=1) in the library named
by JVAE_HIP_LIB (tools/nd_variants.py builds the variants): 16 launches of D2 / D4 at the step's batch, differing elements counted."""
import os, sys, torch
REPO = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
os.environ['JVAE_T2_V1'] = '1'
from jvae_hip import ops
torch.manual_seed(0)
tag = os.path.basename(os.environ.get('JVAE_HIP_LIB', 'libjvae_hip.so'))
for name, N, cin, cout, H in (('D2', 1024, 64, 64, 8), ('D4', 1024, 32, 32, 16), ('D4 n=37', 37, 32, 32, 16)):
    spec = ops.ConvSpec(cin, cout, 5, 2, 2, 1, True)
    x = torch.randn(N, cin, H, H, device='cuda'); w = torch.randn(cin, cout, 5, 5, device='cuda') * 0.05
    b = torch.randn(cout, device='cuda')
    aff = (torch.rand(cin, device='cuda') + 0.5, torch.randn(cin, device='cuda') * 0.3, True)
    y0, s0, _ = ops.conv_fwd_aff_raw(x, w, b, spec, aff, True)
    y0, s0 = y0.clone(), s0.clone()
    bad, bad_launches = 0, 0
    for r in range(16):
        y, s, _ = ops.conv_fwd_aff_raw(x, w, b, spec, aff, True)
        d = int((y != y0).sum()) + int((s != s0).sum())
        bad += d; bad_launches += d > 0
    print(f'{tag} {name}: {bad} elements differ from the first launch over 16 launches ({bad_launches} launches affected)')
