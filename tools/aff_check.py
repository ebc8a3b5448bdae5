import os, sys, torch
import torch.nn.functional as F
REPO = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import ops
torch.manual_seed(0)
def check(name, N, cin, cout, H, tr, s=1, op=0):
    spec = ops.ConvSpec(cin, cout, 5, s, 2, op, tr)
    x = torch.randn(N, cin, H, H, device='cuda'); w = torch.randn((cin, cout, 5, 5) if tr else (cout, cin, 5, 5), device='cuda') * 0.05
    b = torch.randn(cout, device='cuda')
    sc = torch.rand(cin, device='cuda') + 0.5; sh = torch.randn(cin, device='cuda') * 0.3
    y = ops.conv_fwd_aff_raw(x, w, b, spec, (sc, sh, True), False)[0]
    xa = torch.relu(x.double().cpu() * sc.double().cpu().view(1, -1, 1, 1) + sh.double().cpu().view(1, -1, 1, 1))
    if tr:
        ref = F.conv_transpose2d(xa, w.double().cpu(), b.double().cpu(), stride=s, padding=2, output_padding=op)
    else:
        ref = F.conv2d(xa, w.double().cpu(), b.double().cpu(), stride=s, padding=2)
    d = (y.double().cpu() - ref).abs()
    print(f'{name}: max err {float(d.max() / ref.abs().max()):.2e}  elems > 1e-4: {int((d > 1e-4 * ref.abs().max()).sum())} / {d.numel()}  mean diff {float((y.double().cpu() - ref).mean()):.2e}')
check('32->768 N=8 H32 conv', 8, 32, 768, 32, False)
check('32->32 N=8 H32 convT', 8, 32, 32, 32, True)
check('64->32 N=8 H16 convT', 8, 64, 32, 16, True)
check('64->64 N=8 H8 convT', 8, 64, 64, 8, True)
check('64->64 N=8 H8 convT s2', 8, 64, 64, 8, True, 2, 1)
check('32->32 N=8 H16 convT s2', 8, 32, 32, 16, True, 2, 1)
check('40->24 N=5 H16 conv (ragged channels)', 5, 40, 24, 16, False)

# where are the wrong elements of the many-output-channel case?
spec = ops.ConvSpec(32, 768, 5, 1, 2, 0, False)
x = torch.randn(8, 32, 32, 32, device='cuda'); w = torch.randn(768, 32, 5, 5, device='cuda') * 0.05; b = torch.randn(768, device='cuda')
sc = torch.rand(32, device='cuda') + 0.5; sh = torch.randn(32, device='cuda') * 0.3
for rep in range(3):
    y = ops.conv_fwd_aff_raw(x, w, b, spec, (sc, sh, True), False)[0]
    xa = torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    ref = F.conv2d(xa.double().cpu(), w.double().cpu(), b.double().cpu(), padding=2)
    bad = ((y.double().cpu() - ref).abs() > 1e-4 * ref.abs().max())
    idx = bad.nonzero()
    print('rep', rep, 'bad', int(bad.sum()), 'images', sorted(set(idx[:, 0].tolist())), 'o-blocks', sorted(set((idx[:, 1] // 32).tolist()))[:30], 'rows', sorted(set(idx[:, 2].tolist())), 'cols', sorted(set(idx[:, 3].tolist()))[:40])
