"""GPU box: time the stride-1 split-bf16 convolutions of config 2 (forward with BatchNorm sums, and dgrad); JVAE_X3=0 for the
fp32-MFMA kernel, JVAE_HIP_LIB=<other build> for an A/B on one box."""
import os, sys, torch
REPO = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import ops
def timeit(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (name, N, cin, cout, H) in (('D5', 1024, 32, 32, 32), ('D3', 1024, 64, 32, 16), ('D1', 1024, 64, 64, 8)):
    spec = ops.ConvSpec(cin, cout, 5, 1, 2, 0, True)
    x = torch.randn(N, cin, H, H, device='cuda'); w = torch.randn(cin, cout, 5, 5, device='cuda') * 0.05
    b = torch.zeros(cout, device='cuda'); y = ops.conv_fwd_raw(x, w, b, spec); gy = torch.randn_like(y)
    fl = 2.0 * x.numel() * cout * 25
    tf = timeit(lambda: ops.conv_fwd_stats_raw(x, w, b, spec)); td = timeit(lambda: ops.conv_dgrad_raw(gy, w, spec, x.shape))
    tp = timeit(lambda: ops.conv_fwd_raw(x, w, b, spec)); tn = timeit(lambda: ops.conv_fwd_raw(x, w, None, spec)); ts = timeit(lambda: ops.conv_fwd_stats_raw(x, w, None, spec))
    print(f'JVAE_X3={os.environ.get("JVAE_X3", "1")} {name} fwd+stats {tf:6.1f} us {fl/tf/1e6:6.1f} TF | dgrad {td:6.1f} us {fl/td/1e6:6.1f} TF | fwd (bias, no stats) {tp:6.1f} us | fwd (no bias, no stats) {tn:6.1f} us | fwd (no bias, stats) {ts:6.1f} us')
