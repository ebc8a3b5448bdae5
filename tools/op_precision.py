"""Debug helper (GPU box): precision of each op at the real layer shapes against an fp64 CPU reference."""
import os, sys, math
import torch, torch.nn.functional as F
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import ops
def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max())
g = torch.Generator().manual_seed(0)
N = 16
print('--- batchnorm+relu (N,C,P)')
for (n, C, P) in [(16, 32, 1024), (16, 64, 64), (8, 200, 4), (8, 32, 1024)]:
    x = (torch.randn(n, C, P, 1, generator=g) * 1.5 + 0.7)
    gamma = torch.rand(C, generator=g) + 0.5; beta = torch.randn(C, generator=g) * 0.3
    gy = torch.randn(n, C, P, 1, generator=g)
    xr, gr, br = (t.double().requires_grad_(True) for t in (x, gamma, beta))
    yr = torch.relu(F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5)); yr.backward(gy.double())
    x32, g32, b32 = (t.clone().requires_grad_(True) for t in (x, gamma, beta))
    y32 = torch.relu(F.batch_norm(x32, None, None, g32, b32, True, 0.1, 1e-5)); y32.backward(gy)
    xd, gd, bd = (t.cuda().requires_grad_(True) for t in (x, gamma, beta))
    rm, rv = torch.zeros(C).cuda(), torch.ones(C).cuda(); nbt = torch.zeros((), dtype=torch.int64).cuda()
    yd = ops.batchnorm_act(xd, gd, bd, rm, rv, nbt, True, True); yd.backward(gy.cuda())
    print(f'{(n,C,P)}: y {rel(yd,yr):.1e} (torch32 {rel(y32,yr):.1e})  dx {rel(xd.grad,xr.grad):.1e} ({rel(x32.grad,xr.grad):.1e})'
          f'  dgamma {rel(gd.grad,gr.grad):.1e} ({rel(g32.grad,gr.grad):.1e})  dbeta {rel(bd.grad,br.grad):.1e} ({rel(b32.grad,br.grad):.1e})')
print('--- conv (cin,cout,k,s,p,op,tr,H)')
CONVS = [(3, 32, 5, 1, 2, 0, False, 32), (32, 32, 5, 2, 2, 0, False, 32), (64, 200, 7, 1, 0, 0, False, 8),
         (64, 64, 8, 1, 0, 0, True, 1), (64, 64, 5, 2, 2, 1, True, 8), (32, 32, 5, 1, 2, 0, True, 32), (32, 3, 5, 1, 2, 0, False, 32)]
for (cin, cout, k, s, p, op, tr, H) in CONVS:
    x = torch.randn(N, cin, H, H, generator=g)
    w = torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), generator=g) / math.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g)
    def run(dt, dev):
        xx, ww, bb = (t.detach().clone().to(dt).to(dev).requires_grad_(True) for t in (x, w, b))
        if dev == 'cuda':
            y = ops.conv2d(xx, ww, bb, ops.ConvSpec(cin, cout, k, s, p, op, tr))
        elif tr:
            y = F.conv_transpose2d(xx, ww, bb, stride=s, padding=p, output_padding=op)
        else:
            y = F.conv2d(xx, ww, bb, stride=s, padding=p)
        gg = torch.Generator().manual_seed(1)
        gy = torch.randn(y.shape, generator=gg)
        y.backward(gy.to(dt).to(dev))
        return y, xx.grad, ww.grad, bb.grad
    r64, r32, rd = run(torch.float64, 'cpu'), run(torch.float32, 'cpu'), run(torch.float32, 'cuda')
    print((cin, cout, k, s, p, op, tr, H), ' '.join(f'{nm} {rel(a,c):.1e} ({rel(b_,c):.1e})' for nm, a, b_, c in zip(('y', 'dx', 'dw', 'db'), rd, r32, r64)))
