"""GPU box: the polyphase stride-2 forward-type kernel (conv_s2_x3.hip) on the four launches of config 2 - E1 / E3 forward with the
deferred BatchNorm of their input and the BatchNorm sums of their output, D4 / D2 dgrad (the dgrad of a stride-2 transposed layer is a
stride-2 forward-type convolution of the output gradient) - checked against an fp64 reference at a small ragged batch (outputs AND
the folded BatchNorm sums), then timed at the step's batch.  JVAE_EXP_S2_OFF=1: the fp32 matrix-core kernel (A/B)."""
import math, os, sys, torch
import torch.nn.functional as F
REPO = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import ops
# name, N, channels in, channels out, OUTPUT side, forward?
CASES = [('E1 fwd', 512, 32, 32, 16, True), ('E3 fwd', 512, 64, 64, 8, True), ('D4 dgrad', 1024, 32, 32, 16, False),
         ('D2 dgrad', 1024, 64, 64, 8, False)]
def rel(a, b): return float((a.double().cpu() - b).abs().max() / b.abs().max())
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
tag = 'fp32' if os.environ.get('JVAE_EXP_S2_OFF') == '1' else 's2s'
if os.environ.get('ONLY'):
    CASES = [c for c in CASES if os.environ['ONLY'] in c[0]]
for name, N, cin, cout, H, fwd in CASES:
    g = torch.Generator().manual_seed(cin + H)
    n = 5 if H == 16 else 19                                   # ragged: partially filled tiles
    if fwd:
        x = torch.randn(n, cin, 2 * H, 2 * H, generator=g) * torch.exp(torch.randn(n, cin, 1, 1, generator=g))
        w = torch.randn(cout, cin, 5, 5, generator=g) / math.sqrt(cin * 25)
        b = torch.randn(cout, generator=g)
        sc, sh = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.5
        spec = ops.ConvSpec(cin, cout, 5, 2, 2, 0, False)
        a = torch.relu(x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
        ref = F.conv2d(a, w.double(), b.double(), stride=2, padding=2)
        y, st, ns = ops.conv_fwd_aff_raw(x.cuda(), w.cuda(), b.cuda(), spec, (sc.cuda(), sh.cuda(), True), True)
        e_y = rel(y, ref)
        st = st.view(-1)[:cout * ns * 2].view(cout, ns, 2).double().sum(1).cpu()
        piv = ref - b.double().view(1, -1, 1, 1)
        e_s = float(max(((st[:, 0] - piv.sum((0, 2, 3))).abs() / piv.abs().sum((0, 2, 3))).max(),
                        ((st[:, 1] - (piv * piv).sum((0, 2, 3))).abs() / (piv * piv).sum((0, 2, 3))).max()))
        y2 = ops.conv_fwd_raw(x.cuda(), w.cuda(), b.cuda(), spec)
        e_p = rel(y2, F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=2))
        print(f'{tag} {name}: check n={n} out {e_y:.2e} stats {e_s:.2e} nsplit {ns} plain {e_p:.2e}',
              'OK' if e_y < 3e-6 and e_s < 1e-5 and e_p < 3e-6 else 'FAIL')
        xx = torch.randn(N, cin, 2 * H, 2 * H, device='cuda'); ww = torch.randn(cout, cin, 5, 5, device='cuda') * 0.05
        bb = torch.zeros(cout, device='cuda'); aff = (torch.rand(cin, device='cuda') + 0.5, torch.randn(cin, device='cuda'), True)
        t = timeit(lambda: ops.conv_fwd_aff_raw(xx, ww, bb, spec, aff, True))
    else:
        # the transposed layer cin -> cout (small H -> 2H); its dgrad maps gy (cout channels, 2H) to gx (cin channels, H)
        spec = ops.ConvSpec(cin, cout, 5, 2, 2, 1, True)
        gy = torch.randn(n, cout, 2 * H, 2 * H, generator=g)
        w = torch.randn(cin, cout, 5, 5, generator=g) / math.sqrt(cout * 25)
        ref = F.conv2d(gy.double(), w.double(), stride=2, padding=2)            # ConvT weight (cin, cout, k, k) read as Conv2d (out=cin, in=cout)
        y = ops.conv_dgrad_raw(gy.cuda(), w.cuda(), spec, (n, cin, H, H))
        e_y = rel(y, ref)
        print(f'{tag} {name}: check n={n} out {e_y:.2e}', 'OK' if e_y < 3e-6 else 'FAIL')
        gg = torch.randn(N, cout, 2 * H, 2 * H, device='cuda'); ww = torch.randn(cin, cout, 5, 5, device='cuda') * 0.05
        t = timeit(lambda: ops.conv_dgrad_raw(gg, ww, spec, (N, cin, H, H)))
    fl = 2.0 * N * H * H * cin * cout * 25
    print(f'{tag} {name}: {t:7.1f} us  {fl / t / 1e6:6.1f} TF/s')
