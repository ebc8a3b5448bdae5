import os, sys, torch
REPO = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import ops, lib as L
lib = L.load()
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
torch.manual_seed(0)
for (name, N, cin, cout, H, tr) in (('D5', 1024, 32, 32, 32, True), ('D3', 1024, 64, 32, 16, True), ('D1', 1024, 64, 64, 8, True), ('E2', 512, 32, 64, 16, False), ('odd', 37, 48, 40, 16, False)):
    spec = ops.ConvSpec(cin, cout, 5, 1, 2, 0, tr)
    x = torch.randn(N, cin, H, H, device='cuda'); w = torch.randn((cin, cout, 5, 5) if tr else (cout, cin, 5, 5), device='cuda') * 0.05
    b = torch.randn(cout, device='cuda')
    res = {}
    for sh in (0, 1):
        lib.jvae_conv2d_set_split_shape16(sh)
        y, st, ns = ops.conv_fwd_stats_raw(x, w, b, spec)
        gy = torch.randn_like(y)
        gx = ops.conv_dgrad_raw(gy.clone().normal_(generator=torch.Generator(device='cuda').manual_seed(1)), w, spec, x.shape)
        res[sh] = (y, st, gx)
    y0, st0, gx0 = res[0]; y1, st1, gx1 = res[1]
    print(name, 'fwd maxdiff', float((y0 - y1).abs().max()), 'scale', float(y0.abs().max()), 'stats diff', float((st0 - st1).abs().max() / st0.abs().max()), 'dgrad diff', float((gx0 - gx1).abs().max()), float(gx0.abs().max()))
    fl = 2.0 * x.numel() * cout * 25
    line = name
    for rnd in range(2):
        for sh in (0, 1):
            lib.jvae_conv2d_set_split_shape16(sh)
            tf = timeit(lambda: ops.conv_fwd_stats_raw(x, w, b, spec)); td = timeit(lambda: ops.conv_dgrad_raw(gy, w, spec, x.shape))
            line += f' | sh{sh} fwd {tf:6.1f} us {fl/tf/1e6:6.1f} TF dgrad {td:6.1f} us {fl/td/1e6:6.1f} TF'
    print(line)
