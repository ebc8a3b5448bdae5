"""GPU box: run-to-run bit equality of conv5_x3_kernel's forward with a deferred BatchNorm in the tree as it was BEFORE the round-4
staging rewrite (commit 6ab4d44^: per-lane channel block, coefficients from an LDS table inside the `live` branch), built twice -
libjvae_old_S.so as committed (scalar table reads), libjvae_old_V.so with the table read as four 16-byte vectors, the variant round 4
measured run-to-run NONdeterministic.  nd_old/ (not in git) holds that tree's package and the two libraries.
usage: JVAE_HIP_LIB=nd_old/libjvae_old_V.so python tools/nd_old_probe.py"""
import os, sys, torch
REPO = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, 'nd_old', 'pkg')]
from jvae_hip import ops
tag = os.path.basename(os.environ.get('JVAE_HIP_LIB', '?'))
torch.manual_seed(0)
for (name, N, cin, cout, H, tr) in (('D5', 1024, 32, 32, 32, True), ('D3', 1024, 64, 32, 16, True), ('D1', 1024, 64, 64, 8, True),
                                    ('E2', 512, 32, 64, 16, False), ('cat', 8, 32, 768, 32, False)):
    spec = ops.ConvSpec(cin, cout, 5, 1, 2, 0, tr)
    x = torch.randn(N, cin, H, H, device='cuda'); w = torch.randn((cin, cout, 5, 5) if tr else (cout, cin, 5, 5), device='cuda') * 0.05
    b = torch.randn(cout, device='cuda')
    aff = (torch.rand(cin, device='cuda') + 0.5, torch.randn(cin, device='cuda') * 0.3, True)
    y0 = ops.conv_fwd_aff_raw(x, w, b, spec, aff, True)[0].clone()
    bad, hit = 0, 0
    for r in range(16):
        d = int((ops.conv_fwd_aff_raw(x, w, b, spec, aff, True)[0] != y0).sum())
        bad += d; hit += d > 0
    print(f'{tag} {name}: {bad} output elements differ from the first launch over 16 launches ({hit} launches affected)')
