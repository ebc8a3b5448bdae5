"""GPU box: the 4-phase stride-2 transposed kernel (conv_t2_x3.hip) on the four launches of config 2 - D2 / D4 forward with the
deferred BatchNorm of their input and the BatchNorm sums of their output, E1 / E3 dgrad - checked against an fp64 reference at a small
ragged batch (outputs AND the folded BatchNorm sums), then timed at the step's batch."""
import math, os, sys, torch
import torch.nn.functional as F
REPO = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import ops
CASES = [('D2 fwd', 1024, 64, 64, 8, True), ('D4 fwd', 1024, 32, 32, 16, True), ('E1 dgrad', 512, 32, 32, 16, False),
         ('E3 dgrad', 512, 64, 64, 8, False)]
def rel(a, b): return float((a.double().cpu() - b).abs().max() / b.abs().max())
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
tag = 't2s'
for name, N, cin, cout, H, fwd in CASES:
    g = torch.Generator().manual_seed(cin + H)
    n = 5 if H == 16 else 19                                   # ragged: partially filled tiles
    x = torch.randn(n, cin, H, H, generator=g) * torch.exp(torch.randn(n, cin, 1, 1, generator=g))
    w = torch.randn(cin, cout, 5, 5, generator=g) / math.sqrt(cin * 25)
    b = torch.randn(cout, generator=g)
    sc, sh = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.5
    if fwd:
        spec = ops.ConvSpec(cin, cout, 5, 2, 2, 1, True)
        a = torch.relu(x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
        ref = F.conv_transpose2d(a, w.double(), b.double(), stride=2, padding=2, output_padding=1)
        y, st, ns = ops.conv_fwd_aff_raw(x.cuda(), w.cuda(), b.cuda(), spec, (sc.cuda(), sh.cuda(), True), True)
        e_y = rel(y, ref)
        st = st.view(-1)[:cout * ns * 2].view(cout, ns, 2).double().sum(1).cpu()
        piv = ref - b.double().view(1, -1, 1, 1)
        e_s = float(max(((st[:, 0] - piv.sum((0, 2, 3))).abs() / piv.abs().sum((0, 2, 3))).max(),
                        ((st[:, 1] - (piv * piv).sum((0, 2, 3))).abs() / (piv * piv).sum((0, 2, 3))).max()))
        print(f'{tag} {name}: check n={n} out {e_y:.2e} stats {e_s:.2e} nsplit {ns}', 'OK' if e_y < 3e-6 and e_s < 1e-5 else 'FAIL')
        xx = torch.randn(N, cin, H, H, device='cuda'); ww = torch.randn(cin, cout, 5, 5, device='cuda') * 0.05
        bb = torch.zeros(cout, device='cuda'); aff = (torch.rand(cin, device='cuda') + 0.5, torch.randn(cin, device='cuda'), True)
        t = timeit(lambda: ops.conv_fwd_aff_raw(xx, ww, bb, spec, aff, True))
    else:
        spec = ops.ConvSpec(cout, cin, 5, 2, 2, 0, False)    # the stride-2 conv whose dgrad this is: dy has cin channels here
        ref = F.conv_transpose2d(x.double(), w.double(), stride=2, padding=2, output_padding=1)
        y = ops.conv_dgrad_raw(x.cuda(), w.cuda(), spec, (n, cout, 2 * H, 2 * H))
        e_y = rel(y, ref)
        print(f'{tag} {name}: check n={n} out {e_y:.2e}', 'OK' if e_y < 3e-6 else 'FAIL')
        xx = torch.randn(N, cin, H, H, device='cuda'); ww = torch.randn(cin, cout, 5, 5, device='cuda') * 0.05
        t = timeit(lambda: ops.conv_dgrad_raw(xx, ww, spec, (N, cout, 2 * H, 2 * H)))
    fl = 2.0 * N * H * H * cin * cout * 25
    print(f'{tag} {name}: {t:7.1f} us  {fl / t / 1e6:6.1f} TF/s')
