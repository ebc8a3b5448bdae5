"""GPU box: time every conv layer of config 2 (bs=512; decoder on 1024 latents) in its three directions."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import ops
LAYERS = [  # name, N, cin, cout, k, s, p, op, transposed, H
    ('E0', 512, 3, 32, 5, 1, 2, 0, False, 32), ('E1', 512, 32, 32, 5, 2, 2, 0, False, 32),
    ('E2', 512, 32, 64, 5, 1, 2, 0, False, 16), ('E3', 512, 64, 64, 5, 2, 2, 0, False, 16),
    ('E4', 512, 64, 200, 7, 1, 0, 0, False, 8), ('D0', 1024, 64, 64, 8, 1, 0, 0, True, 1),
    ('D1', 1024, 64, 64, 5, 1, 2, 0, True, 8), ('D2', 1024, 64, 64, 5, 2, 2, 1, True, 8),
    ('D3', 1024, 64, 32, 5, 1, 2, 0, True, 16), ('D4', 1024, 32, 32, 5, 2, 2, 1, True, 16),
    ('D5', 1024, 32, 32, 5, 1, 2, 0, True, 32), ('D6', 1024, 32, 3, 5, 1, 2, 0, False, 32)]
def timeit(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
tot = 0.
for name, N, cin, cout, k, s, p, op, tr, H in LAYERS:
    spec = ops.ConvSpec(cin, cout, k, s, p, op, tr)
    x = torch.randn(N, cin, H, H, device='cuda')
    w = torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), device='cuda') * 0.05
    b = torch.zeros(cout, device='cuda')
    y = ops.conv_fwd_raw(x, w, b, spec)
    gy = torch.randn_like(y)
    flops = 2.0 * y.numel() / cout * cout * cin * k * k if not tr else 2.0 * x.numel() * cout * k * k
    aff = None
    if os.environ.get('AFF') == '1' and ops.conv_affine_ok(spec, N, H, H):      # deferred BatchNorm on the input
        aff = (torch.rand(cin, device='cuda') + 0.5, torch.randn(cin, device='cuda'), True)
    if aff is not None:
        t_f = timeit(lambda: ops.conv_fwd_aff_raw(x, w, b, spec, aff, True))
    else:
        t_f = timeit(lambda: ops.conv_fwd_stats_raw(x, w, b, spec))
    t_d = timeit(lambda: ops.conv_dgrad_raw(gy, w, spec, x.shape))
    t_w = timeit(lambda: ops.conv_wgrad_raw(x, gy, spec, w.shape, False, aff=aff))
    tot += t_f + t_d + t_w
    print(f'{name} {flops/1e9:6.2f} GF  fwd {t_f:7.1f} us {flops/t_f/1e6:6.1f} TF | dgrad {t_d:7.1f} us {flops/t_d/1e6:6.1f} TF | wgrad {t_w:7.1f} us {flops/t_w/1e6:6.1f} TF')
print('sum of all conv kernels (E0 dgrad included although unused): %.2f ms' % (tot / 1e3))
