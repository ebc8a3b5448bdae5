"""GPU box: the 3-channel head convolution (32 -> 3, 5x5, 1024 x 32 x 32): time + result under JVAE_SMALLCO_V2 (set by the caller)."""
import os, sys, torch
REPO = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import ops
def timeit(f, reps=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
torch.manual_seed(0)
for (N, cin, cout, H) in ((1024, 32, 3, 32), (512, 32, 3, 64), (37, 32, 3, 32)):
    spec = ops.ConvSpec(cin, cout, 5, 1, 2, 0, False)
    x = torch.randn(N, cin, H, H, device='cuda'); w = torch.randn(cout, cin, 5, 5, device='cuda') * 0.05; b = torch.randn(cout, device='cuda')
    aff = (torch.rand(cin, device='cuda') + 0.5, torch.randn(cin, device='cuda') * 0.3, True)
    y = ops.conv_fwd_raw(x, w, b, spec)
    ya = ops.conv_fwd_aff_raw(x, w, b, spec, aff, False)[0]
    ref = torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu(), padding=2)
    xa = torch.relu(x.double().cpu() * aff[0].double().cpu().view(1, -1, 1, 1) + aff[1].double().cpu().view(1, -1, 1, 1))
    refa = torch.nn.functional.conv2d(xa, w.double().cpu(), b.double().cpu(), padding=2)
    e = float((y.double().cpu() - ref).abs().max() / ref.abs().max()); ea = float((ya.double().cpu() - refa).abs().max() / refa.abs().max())
    t = timeit(lambda: ops.conv_fwd_raw(x, w, b, spec)); ta = timeit(lambda: ops.conv_fwd_aff_raw(x, w, b, spec, aff, False))
    fl = 2.0 * N * H * H * cin * cout * 25
    print(f'V2={os.environ.get("JVAE_SMALLCO_V2", "1")} N={N} H={H}: plain {t:6.1f} us {fl/t/1e6:5.1f} TF err {e:.1e} | deferred BN {ta:6.1f} us err {ea:.1e}')
