"""GPU box: time the conv layers of config 5 (3x64x64, conv32+/deconv32+, bs=256; decoder on 512 latents) in the bf16
B8 path, direction by direction; directions without a native bf16 kernel are reported as '-'."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import ops, ops_b8
B = int(os.environ.get('B', 256))
LAYERS = [  # name, N, cin, cout, k, s, p, op, transposed, H
    ('E0', B, 3, 32, 5, 1, 2, 0, False, 64), ('E1', B, 32, 32, 5, 2, 2, 0, False, 64),
    ('E2', B, 32, 64, 5, 1, 2, 0, False, 32), ('E3', B, 64, 64, 5, 2, 2, 0, False, 32),
    ('E4', B, 64, 128, 5, 1, 2, 0, False, 16), ('E5', B, 128, 128, 5, 2, 2, 0, False, 16),
    ('D1', 2 * B, 128, 128, 5, 1, 2, 0, True, 8), ('D2', 2 * B, 128, 128, 5, 2, 2, 1, True, 8),
    ('D3', 2 * B, 128, 64, 5, 1, 2, 0, True, 16), ('D4', 2 * B, 64, 64, 5, 2, 2, 1, True, 16),
    ('D5', 2 * B, 64, 32, 5, 1, 2, 0, True, 32), ('D6', 2 * B, 32, 32, 5, 2, 2, 1, True, 32),
    ('D7', 2 * B, 32, 32, 5, 1, 2, 0, True, 64), ('D8', 2 * B, 32, 3, 5, 1, 2, 0, False, 64)]
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
    oh, ow = spec.out_hw(H, H)
    x = ops_b8.pack(torch.randn(N, cin, H, H, device='cuda'))
    gy = ops_b8.pack(torch.randn(N, cout, oh, ow, device='cuda'))
    w = torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), device='cuda') * 0.05
    b = torch.zeros(cout, device='cuda')
    flops = 2.0 * N * oh * ow * cout * cin * k * k if not tr else 2.0 * N * H * H * cin * cout * k * k
    mask = ops_b8.native_mask(spec, N, H, H)
    out = f'{name} {flops/1e9:6.2f} GF '
    for bit, nm, f in ((1, 'fwd', lambda: ops_b8.conv_fwd_raw(x, w, b, spec, want_stats=True)),
                       (2, 'dgrad', lambda: ops_b8.conv_dgrad_raw(gy, w, spec, N, H, H)),
                       (4, 'wgrad', (lambda: ops_b8.conv_wgrad_raw(x, gy, spec, w.shape, False)) if hasattr(ops_b8, 'conv_wgrad_raw') else None)):
        if mask & bit and f is not None:
            t = timeit(f); tot += t
            out += f'| {nm} {t:7.1f} us {flops/t/1e6:6.1f} TF '
        else:
            out += f'| {nm}       -            '
    print(out)
print('sum of native bf16 conv kernels: %.2f ms' % (tot / 1e3))
