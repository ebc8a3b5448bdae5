import os, sys, torch
REPO = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import ops
torch.manual_seed(0)
for (name, N, cin, cout, H, tr) in (('D5', 1024, 32, 32, 32, True), ('D3', 1024, 64, 32, 16, True), ('D1', 1024, 64, 64, 8, True), ('E2', 512, 32, 64, 16, False), ('cat', 8, 32, 768, 32, False)):
    spec = ops.ConvSpec(cin, cout, 5, 1, 2, 0, tr)
    x = torch.randn(N, cin, H, H, device='cuda'); w = torch.randn((cin, cout, 5, 5) if tr else (cout, cin, 5, 5), device='cuda') * 0.05
    b = torch.randn(cout, device='cuda')
    aff = (torch.rand(cin, device='cuda') + 0.5, torch.randn(cin, device='cuda') * 0.3, True)
    for mode in ('aff', 'plain'):
        f = (lambda: ops.conv_fwd_aff_raw(x, w, b, spec, aff, True)[0]) if mode == 'aff' else (lambda: ops.conv_fwd_raw(x, w, b, spec))
        y0 = f().clone()
        bad = 0
        for r in range(8):
            y = f()
            bad += int((y != y0).sum())
        print(name, mode, 'elements differing from the first launch over 8 launches:', bad)

# weight gradients with a deferred BatchNorm on the layer input (stride 1 and 2), fp32 and bf16 (B8) operands
from jvae_hip import ops_b8
for (name, N, cin, cout, H, s, tr) in (('E1 wgrad', 512, 32, 32, 32, 2, False), ('D2 wgrad', 1024, 64, 64, 8, 2, True), ('D5 wgrad', 1024, 32, 32, 32, 1, True), ('E2 wgrad', 512, 32, 64, 16, 1, False)):
    spec = ops.ConvSpec(cin, cout, 5, s, 2, 1 if (tr and s == 2) else 0, tr)
    x = torch.randn(N, cin, H, H, device='cuda')
    wshape = (cin, cout, 5, 5) if tr else (cout, cin, 5, 5)
    y = ops.conv_fwd_raw(x, torch.zeros(wshape, device='cuda'), torch.zeros(cout, device='cuda'), spec)
    gy = torch.randn_like(y)
    aff = (torch.rand(cin, device='cuda') + 0.5, torch.randn(cin, device='cuda') * 0.3, True)
    def f():
        gw = torch.zeros(wshape, device='cuda')
        ops.conv_wgrad_raw(x, gy, spec, wshape, False, gw, None, aff=aff)
        return gw
    g0 = f().clone(); bad = 0
    for r in range(8): bad += int((f() != g0).sum())
    print(name, 'fp32 aff: elements differing over 8 launches:', bad)
    xb, gyb = ops_b8.pack(x), ops_b8.pack(gy)
    C8 = (cin + 7) // 8 * 8
    coef = torch.zeros(2, C8, device='cuda'); coef[0, :cin], coef[1, :cin] = aff[0], aff[1]
    affb = (coef[0], coef[1], True)
    def fb():
        return ops_b8.conv_wgrad_raw(xb, gyb, spec, wshape, False, aff=affb)[0]
    def fy():
        return ops_b8.conv_fwd_raw(xb, torch.randn(wshape, device='cuda', generator=torch.Generator(device='cuda').manual_seed(1)) * 0.05, None, spec, aff=affb)[0]
    for nm, fn in (('bf16 wgrad aff', fb), ('bf16 fwd aff', fy)):
        try:
            a0 = fn().clone(); bad = 0
            for r in range(8): bad += int((fn() != a0).sum())
            print(name, nm, 'elements differing over 8 launches:', bad)
        except Exception as e:
            print(name, nm, 'skipped:', str(e)[:60])
