"""GPU parity of every HIP op (through the C ABI) against a plain PyTorch-CPU fp32 reference of the same op.
Tolerances are stated per test; fp32 MFMA accumulates in k order, so 1e-4 relative to the output scale is
the bar (north_star: 1e-4 relative fp32)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def test_library_loaded_on_gpu():
    from jvae_hip import lib
    assert lib.load().jvae_version()
    assert torch.cuda.is_available()


@pytest.mark.parametrize('M,N,K', [(512, 128, 800), (1024, 4096, 64), (37, 75, 33), (200, 3136, 2048), (64, 64, 1)])
@pytest.mark.parametrize('layout', ['nn', 'nt', 'tn', 'tt'])
def test_gemm_layouts(M, N, K, layout):
    from jvae_hip import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(K, N, generator=g)
    bias = torch.randn(N, generator=g)
    ref = A @ B + bias
    Ad = (A if layout[0] == 'n' else A.t().contiguous()).to(DEV)
    Bd = (B if layout[1] == 'n' else B.t().contiguous()).to(DEV)
    sA = (K, 1, 0) if layout[0] == 'n' else (1, M, 0)
    sB = (N, 1, 0) if layout[1] == 'n' else (1, K, 0)
    C = torch.empty(M, N, device=DEV)
    ops.gemm(M, N, K, Ad, sA, Bd, sB, C, (N, 1, 0), bias=bias.to(DEV), bias_mode=1)
    assert rel(C, ref) < 2e-5
    # split-K with atomics on a pre-zeroed C
    C2 = torch.zeros(M, N, device=DEV)
    ops.gemm(M, N, K, Ad, sA, Bd, sB, C2, (N, 1, 0), bias=bias.to(DEV), bias_mode=1, splitk=4)
    assert rel(C2, ref) < 2e-5


CONVS = [  # (cin, cout, k, s, p, op, transposed, H)   every layer geometry of conv32 / deconv32 / conv32+ / deconv32+
    (3, 32, 5, 1, 2, 0, False, 32), (32, 32, 5, 2, 2, 0, False, 32), (32, 64, 5, 1, 2, 0, False, 16),
    (64, 64, 5, 2, 2, 0, False, 16), (64, 200, 7, 1, 0, 0, False, 8), (128, 200, 3, 1, 0, 0, False, 8),
    (64, 64, 8, 1, 0, 0, True, 1), (64, 64, 5, 1, 2, 0, True, 8), (64, 64, 5, 2, 2, 1, True, 8),
    (64, 32, 5, 1, 2, 0, True, 16), (32, 32, 5, 2, 2, 1, True, 16), (32, 32, 5, 1, 2, 0, True, 32),
    (32, 3, 5, 1, 2, 0, False, 32), (8, 128, 4, 1, 0, 0, True, 5), (5, 7, 3, 1, 1, 0, False, 9),
]


@pytest.mark.parametrize('cin,cout,k,s,p,op,tr,H', CONVS)
@pytest.mark.parametrize('N', [3, 8])
def test_conv_all_directions(cin, cout, k, s, p, op, tr, H, N):
    from jvae_hip import ops
    g = torch.Generator().manual_seed(cin * 131 + cout * 17 + k + H)
    x = torch.randn(N, cin, H, H, generator=g)
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    w = torch.randn(wshape, generator=g) / math.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    if tr:
        yr = F.conv_transpose2d(xr, wr, br, stride=s, padding=p, output_padding=op)
    else:
        yr = F.conv2d(xr, wr, br, stride=s, padding=p)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    spec = ops.ConvSpec(cin, cout, k, s, p, op, tr)
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    yd = ops.conv2d(xd, wd, bd, spec)
    assert yd.shape == yr.shape
    assert rel(yd, yr) < 2e-5
    yd.backward(gy.to(DEV))
    assert rel(xd.grad, xr.grad) < 2e-5
    assert rel(wd.grad, wr.grad) < 5e-5
    assert rel(bd.grad, br.grad) < 2e-5


@pytest.mark.parametrize('N', [64, 37, 512])
@pytest.mark.parametrize('bias', [True, False])
def test_small_grid_7x7_head_at_training_batch_sizes(N, bias):
    """features.12 of conv32 (Conv2d 64 -> 200, 7x7 on 8x8 maps -> 2x2) at training batch sizes incl. a ragged one (the
    all-geometries test above runs N = 3 / 8, where the K-sliced split-bf16 products are not selected): forward, dgrad and
    weight gradient against torch, accumulation onto an existing gradient, run-to-run bit equality."""
    from jvae_hip import ops
    g = torch.Generator().manual_seed(N + int(bias))
    x = torch.randn(N, 64, 8, 8, generator=g)
    w = torch.randn(200, 64, 7, 7, generator=g) / math.sqrt(64 * 49)
    b = torch.randn(200, generator=g) if bias else None
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    spec = ops.ConvSpec(64, 200, 7, 1, 0, 0, False)
    outs = []
    for _ in range(2):
        xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
        bd = b.to(DEV).requires_grad_(True) if bias else None
        yd = ops.conv2d(xd, wd, bd, spec)
        yd.backward(gy.to(DEV), retain_graph=True)
        first = wd.grad.clone()
        yd.backward(gy.to(DEV))
        outs.append((yd.detach().clone(), xd.grad.clone(), first, wd.grad.clone()))
    y0, gx0, gw1, gw2 = outs[0]
    assert rel(y0, yr) < 2e-5
    assert rel(gx0, 2 * xr.grad) < 2e-5
    assert rel(gw1, wr.grad) < 5e-5 and rel(gw2, 2 * wr.grad) < 5e-5
    if bias:
        assert rel(bd.grad, 2 * br.grad) < 2e-5
    for a, c in zip(outs[0], outs[1]):
        assert torch.equal(a, c)


@pytest.mark.parametrize('N', [300, 1024])
def test_point_input_transposed_conv_weight_gradient(N):
    """imager.0 of deconv32 (ConvTranspose2d 64->64 8x8 on a 1x1 input) at training batch sizes: the weight gradient is the
    plain product x^T . dy cut into K pieces (over the batch) that are folded in a fixed order - accumulated onto an existing
    gradient, bit-reproducible from run to run."""
    from jvae_hip import ops
    g = torch.Generator().manual_seed(N)
    x = torch.randn(N, 64, 1, 1, generator=g)
    w = torch.randn(64, 64, 8, 8, generator=g) / 8
    gy = torch.randn(N, 64, 8, 8, generator=g)
    ref = torch.einsum('nc,nokl->cokl', x[:, :, 0, 0].double(), gy.double()).float()
    spec = ops.ConvSpec(64, 64, 8, 1, 0, 0, True)
    outs = []
    for _ in range(2):
        xd, wd = x.to(DEV), w.to(DEV).requires_grad_(True)
        y = ops.conv2d(xd, wd, None, spec)
        y.backward(gy.to(DEV), retain_graph=True)
        first = wd.grad.clone()
        y.backward(gy.to(DEV))                      # second backward: accumulates onto the first
        outs.append((first, wd.grad.clone()))
    assert rel(outs[0][0], ref) < 2e-5 and rel(outs[0][1], 2 * ref) < 2e-5
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize('cin,cout,tr,H,N', [(32, 32, True, 32, 6), (32, 64, False, 16, 9), (64, 32, True, 16, 5),
                                             (64, 64, True, 8, 7), (48, 40, False, 16, 3), (128, 64, False, 64, 1),
                                             (17, 33, False, 8, 1), (16, 3, True, 32, 2), (250, 32, False, 8, 3),
                                             # 2048 workgroups: the 16x16x32 kernel takes TWO tiles per workgroup (round 4 / 5)
                                             (32, 64, False, 32, 256)])
@pytest.mark.parametrize('shape16', [1, 0], ids=['16x16x32', '32x32x16'])
def test_split_bf16_conv_is_fp32_accurate(cin, cout, tr, H, N, shape16):
    """conv_x3.hip computes fp32 convolutions on the bf16 matrix cores (3-way exact operand split, 6 products): measured
    against an fp64 reference its error must be at the level of the fp32-MFMA kernel's own rounding (not bf16's 4e-3),
    on data with a wide dynamic range, in forward (with BatchNorm partial sums) and dgrad; the two modes agree to 2e-6.
    Both MFMA shapes of the kernel (jvae_conv2d_set_split_shape16: 16x16x32 is the default up to 32-wide maps, 32x32x16 serves
    64-wide ones and - through the switch - everything)."""
    from jvae_hip import lib, ops
    L = lib.load()
    if shape16 == 0 and N > 16:
        pytest.skip('the large case is the two-tiles-per-workgroup form of the 16x16x32 kernel')
    g = torch.Generator().manual_seed(cin * 7 + cout + H)
    x = torch.randn(N, cin, H, H, generator=g) * torch.exp(2 * torch.randn(N, cin, 1, 1, generator=g))
    w = torch.randn((cin, cout, 5, 5) if tr else (cout, cin, 5, 5), generator=g) / math.sqrt(cin * 25)
    b = torch.randn(cout, generator=g)
    if tr:
        ref = F.conv_transpose2d(x.double(), w.double(), b.double(), padding=2)
    else:
        ref = F.conv2d(x.double(), w.double(), b.double(), padding=2)
    gy = torch.randn(ref.shape, generator=g)
    if tr:
        gref = F.conv2d(gy.double(), w.double(), padding=2)
    else:
        gref = F.conv_transpose2d(gy.double(), w.double(), padding=2)
    spec = ops.ConvSpec(cin, cout, 5, 1, 2, 0, tr)
    xd, wd, bd, gyd = x.to(DEV), w.to(DEV), b.to(DEV), gy.to(DEV)
    out = {}
    old = L.jvae_conv2d_set_split_bf16(1)
    old_shape = L.jvae_conv2d_set_split_shape16(shape16)
    try:
        for mode in (1, 0):
            L.jvae_conv2d_set_split_bf16(mode)
            y = ops.conv_fwd_raw(xd, wd, bd, spec)
            ys = ops.conv_fwd_stats_raw(xd, wd, bd, spec)
            dx = ops.conv_dgrad_raw(gyd, wd, spec, xd.shape)
            y2, st, ns = ys
            assert torch.equal(y, y2)
            if st is not None:        # BatchNorm partial sums of (y - bias) from the kernel epilogue (DPP reductions)
                part = st[:cout * ns * 2].view(cout, ns, 2).double().sum(1).cpu()
                yc = (y.double().cpu() - b.double().view(1, -1, 1, 1))
                assert float((part[:, 0] - yc.sum((0, 2, 3))).abs().max() / yc.abs().sum((0, 2, 3)).max()) < 1e-5
                assert float((part[:, 1] - (yc * yc).sum((0, 2, 3))).abs().max() / (yc * yc).sum((0, 2, 3)).max()) < 1e-5
            out[mode] = (rel(y, ref), rel(dx, gref), y, dx)
    finally:
        L.jvae_conv2d_set_split_bf16(old)
        L.jvae_conv2d_set_split_shape16(old_shape)
    assert out[1][0] < 3e-6 and out[1][1] < 3e-6, out[1][:2]          # split bf16: measured 2e-7 .. 1.3e-6 (K = 1600 .. 3200)
    assert out[0][0] < 5e-6 and out[0][1] < 5e-6, out[0][:2]          # fp32 MFMA (k-ordered fmaf chain): up to 1.5e-6
    assert out[1][0] < 2 * out[0][0] + 1e-7 and out[1][1] < 2 * out[0][1] + 1e-7, (out[1][:2], out[0][:2])
    assert rel(out[1][2], out[0][2]) < 5e-6 and rel(out[1][3], out[0][3]) < 5e-6


@pytest.mark.parametrize('cin,cout,s,tr,H,N', [(32, 32, 1, True, 32, 6), (32, 64, 1, False, 16, 9), (64, 32, 1, True, 16, 5),
                                               (64, 64, 1, True, 8, 7), (32, 32, 2, False, 32, 5), (64, 64, 2, True, 8, 6),
                                               (3, 32, 1, False, 32, 4), (32, 3, 1, False, 32, 4), (48, 40, 2, False, 16, 3),
                                               (17, 33, 1, False, 8, 2)])
def test_split_bf16_wgrad_is_fp32_accurate(cin, cout, s, tr, H, N):
    """conv_wgrad_x3.hip computes the fp32 weight gradient on the bf16 matrix cores (both operands split exactly 3-way,
    6 products, pixel-major LDS planes read with ds_read_b64_tr_b16).  Against an fp64 reference on data with a wide
    dynamic range its error must sit at the fp32-MFMA kernel's own rounding level (not bf16's 4e-3); the two kernels agree
    to 5e-6; with a deferred BatchNorm(+ReLU) on the layer input as well; run-to-run bit-identical."""
    from jvae_hip import lib, ops
    L = lib.load()
    g = torch.Generator().manual_seed(cin * 7 + cout + H + s)
    x = torch.randn(N, cin, H, H, generator=g) * torch.exp(2 * torch.randn(N, cin, 1, 1, generator=g))
    wshape = (cin, cout, 5, 5) if tr else (cout, cin, 5, 5)
    op = 1 if (tr and s == 2) else 0
    sc = torch.rand(cin, generator=g) + 0.5
    sh = torch.randn(cin, generator=g) * 0.5
    spec = ops.ConvSpec(cin, cout, 5, s, 2, op, tr)
    res = {}
    for aff in (None, (sc, sh, True)):
        a = x.double() if aff is None else torch.relu(x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
        w64 = torch.zeros(wshape, dtype=torch.float64, requires_grad=True)
        y64 = (F.conv_transpose2d(a, w64, None, stride=s, padding=2, output_padding=op) if tr
               else F.conv2d(a, w64, None, stride=s, padding=2))
        gy = torch.randn(y64.shape, generator=torch.Generator().manual_seed(5)) * \
            torch.exp(torch.randn(N, y64.shape[1], 1, 1, generator=torch.Generator().manual_seed(6)))
        y64.backward(gy.double())
        affd = None if aff is None else (sc.to(DEV), sh.to(DEV), True)
        if affd is not None and not ops.conv_affine_ok(spec, N, H, H):
            continue
        old = L.jvae_conv2d_set_split_bf16(1)
        try:
            out = {}
            for mode in (1, 0):
                L.jvae_conv2d_set_split_bf16(mode)
                gw, _ = ops.conv_wgrad_raw(x.to(DEV), gy.to(DEV), spec, wshape, False, aff=affd)
                gw2, _ = ops.conv_wgrad_raw(x.to(DEV), gy.to(DEV), spec, wshape, False, aff=affd)
                assert torch.equal(gw, gw2)
                out[mode] = (rel(gw, w64.grad), gw)
        finally:
            L.jvae_conv2d_set_split_bf16(old)
        assert out[1][0] < 3e-6, (aff is not None, out[1][0], out[0][0])      # split bf16
        assert out[0][0] < 5e-6, (aff is not None, out[0][0])                  # fp32 MFMA (k-ordered fmaf chain)
        assert out[1][0] < 2 * out[0][0] + 2e-7, (out[1][0], out[0][0])
        assert rel(out[1][1], out[0][1]) < 5e-6
        res[aff is not None] = (out[1][0], out[0][0])
    print(f'wgrad {cin}->{cout} s{s} tr={tr} H={H}: error vs fp64 split-bf16 / fp32-MFMA: {res}')


# (98,32,1024), (672,32,1024), (42,3,4096), (49,32,1024), (37,64,256): plans whose trailing image parts are EMPTY
# ((parts-1)*ceil(N/parts) >= N) - the ragged last batches the reference never drops (cvae.py:2245-2249)
@pytest.mark.parametrize('N,C,P,relu', [(8, 32, 1024, True), (5, 3, 1024, False), (16, 200, 4, True), (2, 64, 63, True),
                                        (98, 32, 1024, True), (672, 32, 1024, True), (42, 3, 4096, False),
                                        (49, 32, 1024, True), (37, 64, 256, True), (74, 3, 1023, False),
                                        (8, 32, 1024, 2), (37, 64, 255, 2)])          # 2: leaky ReLU (misc.py:24-27)
def test_batchnorm_train(N, C, P, relu):
    from jvae_hip import ops
    g = torch.Generator().manual_seed(N + C + P)
    x = torch.randn(N, C, P, 1, generator=g) * 2 + 3
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    xr, gr, br = (t.clone().requires_grad_(True) for t in (x, gamma, beta))
    rmr, rvr = rm.clone(), rv.clone()
    pre = F.batch_norm(xr, rmr, rvr, gr, br, True, 0.1, 1e-5)
    act = {0: lambda t: t, 1: torch.relu, 2: F.leaky_relu}[int(relu)]
    yr = act(pre)
    gy = torch.randn(yr.shape, generator=g)
    xd, gd, bd = (t.to(DEV).requires_grad_(True) for t in (x, gamma, beta))
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    yd = ops.batchnorm_act(xd, gd, bd, rmd, rvd, nbt, True, relu)
    assert rel(yd, yr) < 1e-5
    assert rel(rmd, rmr) < 1e-6 and rel(rvd, rvr) < 1e-5 and int(nbt) == 1
    # Two fp32 evaluations of the pre-activation differ by ~1e-6: among millions of elements a few sit that close to the ReLU
    # threshold and take the other branch.  The kernel's backward re-derives ITS forward's mask bit-exactly (bn_coef), so the
    # reference backward is taken with the kernel's mask - which may differ from torch's only where |pre| is rounding noise.
    mask = torch.ones_like(pre, dtype=torch.bool)
    if relu:
        mask = yd.detach().cpu() > 0
        flipped = mask != (pre.detach() > 0)
        assert int(flipped.sum()) <= 8 and (not bool(flipped.any()) or float(pre.detach()[flipped].abs().max()) < 2e-5)
    pre.backward(gy * (mask if relu != 2 else torch.where(mask, 1., 0.01)))
    yd.backward(gy.to(DEV))
    assert rel(xd.grad, xr.grad) < 1e-4
    assert rel(gd.grad, gr.grad) < 1e-4 and rel(bd.grad, br.grad) < 1e-4
    # eval mode uses the running statistics
    ye = ops.batchnorm_act(xd.detach(), gd.detach(), bd.detach(), rmd, rvd, nbt, False, relu)
    yer = F.batch_norm(x, rmr, rvr, gamma, beta, False, 0.1, 1e-5)
    assert rel(ye, act(yer)) < 1e-5


@pytest.mark.parametrize('N,C,P', [(49, 32, 1024), (98, 64, 256), (5, 3, 1023)])
def test_channel_sum_ragged_partitions(N, C, P):
    from jvae_hip import ops
    g = torch.Generator().manual_seed(N)
    t = torch.randn(N, C, P, generator=g)
    out = ops.channel_sum(t.to(DEV), N, C, P)
    assert rel(out, t.double().sum((0, 2)).float()) < 1e-5


@pytest.mark.parametrize('act', [0, 1, 2, 3])
def test_linear(act):
    from jvae_hip import ops
    g = torch.Generator().manual_seed(act)
    x = torch.randn(2, 48, 800, generator=g)
    w = torch.randn(64, 800, generator=g) / 28
    b = torch.randn(64, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.linear(xr, wr, br)
    yr = [lambda t: t, torch.relu, torch.sigmoid, F.leaky_relu][act](yr)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    yd = ops.linear(xd, wd, bd, act)
    assert rel(yd, yr) < 2e-5
    yd.backward(gy.to(DEV))
    assert rel(xd.grad, xr.grad) < 2e-5 and rel(wd.grad, wr.grad) < 2e-5 and rel(bd.grad, br.grad) < 2e-5


AFF_CONVS = [(32, 32, 5, 2, 2, 0, False, 32), (32, 64, 5, 1, 2, 0, False, 16), (64, 64, 5, 1, 2, 0, True, 8),
             (64, 64, 5, 2, 2, 1, True, 8), (32, 32, 5, 2, 2, 1, True, 16), (32, 32, 5, 1, 2, 0, True, 32),
             (32, 3, 5, 1, 2, 0, False, 32), (64, 32, 5, 1, 2, 0, True, 16),
             # channel counts that are no multiples of the 16-channel K step / 8-channel staging block / 32-channel output block
             # (round 4: the staging clamps the channel address and the coefficient index of the missing channels)
             (24, 32, 5, 1, 2, 0, False, 16), (48, 40, 5, 1, 2, 0, False, 16), (17, 33, 5, 1, 2, 0, False, 8),
             (40, 32, 5, 2, 2, 0, False, 16)]


@pytest.mark.parametrize('cin,cout,k,s,p,op,tr,H', AFF_CONVS)
@pytest.mark.parametrize('relu', [True, False, 2])
def test_conv_with_deferred_batchnorm_input(cin, cout, k, s, p, op, tr, H, relu):
    """conv(x*scale + shift [relu]) with the per-channel transform applied inside the kernels (forward and weight
    gradient) against PyTorch on the explicitly transformed input; dgrad is w.r.t. the transformed input."""
    from jvae_hip import ops
    N = 5
    g = torch.Generator().manual_seed(cin * 7 + cout + H + int(relu))
    x = torch.randn(N, cin, H, H, generator=g)
    sc = torch.rand(cin, generator=g) + 0.5
    sh = torch.randn(cin, generator=g) * 0.5
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    w = (torch.randn(wshape, generator=g) / math.sqrt(cin * k * k)).requires_grad_(True)
    b = torch.randn(cout, generator=g)
    a = x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    a = {0: lambda t: t, 1: torch.relu, 2: F.leaky_relu}[int(relu)](a)            # 2: leaky ReLU (slope 0.01)
    yr = F.conv_transpose2d(a, w, b, stride=s, padding=p, output_padding=op) if tr else F.conv2d(a, w, b, stride=s, padding=p)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    spec = ops.ConvSpec(cin, cout, k, s, p, op, tr)
    assert ops.conv_affine_ok(spec, N, H, H)
    aff = (sc.to(DEV), sh.to(DEV), relu)
    y, st, ns = ops.conv_fwd_aff_raw(x.to(DEV), w.detach().to(DEV), b.to(DEV), spec, aff, True)
    assert rel(y, yr) < 3e-5
    gw, _ = ops.conv_wgrad_raw(x.to(DEV), gy.to(DEV), spec, wshape, False, aff=aff)
    assert rel(gw, w.grad) < 3e-5
    assert not ops.conv_affine_ok(ops.ConvSpec(64, 200, 7, 1, 0), N, 8, 8)       # generic path: must materialise


@pytest.mark.parametrize('cin,cout,k,s,p,op,tr,H', [(128, 200, 3, 1, 0, 0, False, 8), (8, 128, 4, 1, 0, 0, True, 5)])
def test_small_grid_wgrad_joint_product(cin, cout, k, s, p, op, tr, H):
    """6x6 / 5x5 folded grids (heads of conv32+ / deconv32+) at a batch where the weight gradient runs as one product
    over (position, image) with deterministic K slices: against PyTorch, and run-to-run identical."""
    from jvae_hip import ops
    N = 64
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(N, cin, H, H, generator=g)
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    w = (torch.randn(wshape, generator=g) / math.sqrt(cin * k * k)).requires_grad_(True)
    b = torch.randn(cout, generator=g).requires_grad_(True)
    yr = F.conv_transpose2d(x, w, b, stride=s, padding=p, output_padding=op) if tr else F.conv2d(x, w, b, stride=s, padding=p)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    spec = ops.ConvSpec(cin, cout, k, s, p, op, tr)
    gw1, gb1 = ops.conv_wgrad_raw(x.to(DEV), gy.to(DEV), spec, wshape, True)
    gw2, _ = ops.conv_wgrad_raw(x.to(DEV), gy.to(DEV), spec, wshape, True)
    assert rel(gw1, w.grad) < 3e-5 and rel(gb1, b.grad) < 3e-5
    assert torch.equal(gw1, gw2)
    assert rel(ops.conv_fwd_raw(x.to(DEV), w.detach().to(DEV), b.detach().to(DEV), spec), yr) < 3e-5


@pytest.mark.parametrize('act', [0, 1])
def test_linear_split_k(act):
    """Dense head of the 64x64 model (256 x 7200 -> 200): too few output tiles, so K is sliced over the batch dimension
    of one GEMM launch and folded in a fixed order; must agree with the plain product and be run-to-run identical."""
    from jvae_hip import ops
    assert ops._linear_split(256, 200, 7200) > 1 and ops._linear_split(512, 64, 64) == 1
    g = torch.Generator().manual_seed(5 + act)
    x = torch.randn(256, 7200, generator=g)
    w = torch.randn(200, 7200, generator=g) / 85
    b = torch.randn(200, generator=g)
    yr = F.linear(x, w, b)
    yr = torch.relu(yr) if act else yr
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    y1 = ops.linear(xd, wd, bd, act)
    y2 = ops.linear(xd, wd, bd, act)
    assert rel(y1, yr) < 2e-5
    assert torch.equal(y1, y2)
    # backward through the sliced products (weight gradient: K = rows)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr2 = F.linear(xr, wr, b)
    yr2 = torch.relu(yr2) if act else yr2
    gy = torch.randn(yr2.shape, generator=g)
    yr2.backward(gy)
    xd2, wd2 = xd.clone().requires_grad_(True), wd.clone().requires_grad_(True)
    ops.linear(xd2, wd2, bd, act).backward(gy.to(DEV))
    assert rel(xd2.grad, xr.grad) < 2e-5 and rel(wd2.grad, wr.grad) < 2e-5


@pytest.mark.parametrize('prior,var_dim', [('gaussian', 'scalar'), ('gaussian', 'diag'), ('gaussian', 'full'),
                                           ('tilted', 'scalar'), ('uniform', 'scalar')])
@pytest.mark.parametrize('K,L', [(64, 1), (200, 2), (16, 1)])
def test_latent_against_oracle(prior, var_dim, K, L):
    """clip + reparameterise + KL, forward and backward, vs the CPU oracle's prior_kl (oracle/jvae_oracle.py)."""
    from jvae_hip import ops
    from oracle import jvae_oracle as O
    from oracle.det_init import det_tensor
    N, C = 12, 7
    g = torch.Generator().manual_seed(K + L)
    mu = torch.randn(N, K, generator=g)
    lv_raw = torch.randn(N, K, generator=g) * 2
    lv_raw[0, 0], lv_raw[1, 1] = 25., -30.           # exercise the +-20 clip and its zero sub-gradient
    eps = torch.randn(L + 1, N, K, generator=g)
    eps[0] = 0
    y = torch.randint(0, C, (N,), generator=g)
    means = det_tensor('encoder.prior.mean', (C, K))
    T = det_tensor('encoder.prior._var_parameter', {'scalar': (C,), 'diag': (C, K), 'full': (C, K, K)}[var_dim])
    w = 0.3
    tau = {'tilted': 5., 'uniform': 1.5}.get(prior, 0.)
    sp = dict(K=K, prior=dict(distribution=prior, tau=tau, var_dim=var_dim))
    mr, lr_, er, meansr, Tr = (t.clone().requires_grad_(True) for t in (mu, lv_raw, eps, means, T))
    P = {'encoder.prior.mean': meansr, 'encoder.prior._var_parameter': Tr}
    lvc = torch.clip(lr_, -20, 20)
    zr = mr + torch.exp(0.5 * lvc) * er
    kd = O.prior_kl(sp, P, mr, lvc, y, w)
    gz = torch.randn(zr.shape, generator=g)
    gk, gd, gv = (torch.randn(N, generator=g) for _ in range(3))
    obj = (zr * gz).sum() + (kd['kl'] * gk).sum() + (kd['distance'] * gd).sum() + (kd['var_kl'] * gv).sum()
    obj.backward()
    alpha = 0.
    if prior == 'uniform':
        phi = 0.5 * (1 + math.erf(tau / math.sqrt(2)))
        alpha = math.log(2 * tau) - math.log(2 * phi - 1)
    md, ld, meansd, Td = (t.to(DEV).requires_grad_(True) for t in (mu, lv_raw, means, T))
    lv, z, kl, zd, vkl, dzd = ops.latent(md, ld, eps.to(DEV), y.to(DEV), meansd, Td, prior=prior, var_dim=var_dim,
                                         tau=tau, alpha=alpha, w=w)
    assert rel(lv, lvc) < 1e-6 and rel(z, zr) < 1e-5
    assert rel(kl, kd['kl']) < 2e-5 and rel(zd, kd['distance']) < 2e-5
    if prior != 'tilted':
        assert rel(vkl, kd['var_kl']) < 2e-5
    dm = means.mean(0)
    dz_ref = (mu - dm).pow(2).sum(1) + (means.pow(2).sum(1).mean(0) - dm.pow(2).sum())
    assert rel(dzd, dz_ref) < 2e-5
    objd = (z * gz.to(DEV)).sum() + (kl * gk.to(DEV)).sum() + (zd * gd.to(DEV)).sum() + (vkl * gv.to(DEV)).sum()
    objd.backward()
    assert rel(md.grad, mr.grad) < 5e-5
    assert rel(ld.grad, lr_.grad) < 5e-5
    assert float(ld.grad[0, 0]) == 0. and float(ld.grad[1, 1]) == 0.
    assert rel(meansd.grad, meansr.grad) < 5e-5
    if var_dim != 'scalar':
        assert rel(Td.grad, Tr.grad) < 5e-5


@pytest.mark.parametrize('is_log', [True, False])
def test_recon(is_log):
    from jvae_hip import ops
    g = torch.Generator().manual_seed(5)
    L, N, D = 2, 6, 3072
    xr_ = torch.randn(L + 1, N, 3, 32, 32, generator=g)
    x = torch.rand(N, 3, 32, 32, generator=g)
    s = torch.tensor([0.3 if is_log else 0.7])
    a, sr = xr_.clone().requires_grad_(True), s.clone().requires_grad_(True)
    sig = sr.exp() if is_log else sr
    ref = ((a[1:] / sig - x / sig) ** 2).reshape(L, N, -1).mean(-1)
    gw = torch.randn(L, N, generator=g)
    (ref * gw).sum().backward()
    ad, sd = xr_.to(DEV).requires_grad_(True), s.to(DEV).requires_grad_(True)
    out = ops.recon_wmse(ad, x.to(DEV), sd, is_log)
    assert rel(out, ref) < 1e-5
    (out * gw.to(DEV)).sum().backward()
    assert rel(ad.grad, a.grad) < 1e-5 and float(ad.grad[0].abs().max()) == 0.
    assert rel(sd.grad, sr.grad) < 1e-5


@pytest.mark.parametrize('shape', [((3,), (5,), (3, 8, 8)), ((2,), (4,), (1, 28, 28)), ((1,), (3,), (7,))])
def test_public_mse_loss_every_row_without_a_padded_copy(shape):
    """module.losses.mse_loss (reference losses.py:8-27): mean squares of EVERY row of x_output against x_target, values and
    gradient, against F.mse_loss; the kernels get the address one row in front of x_output (ops.mse_rows) instead of a copy."""
    from module.losses import mse_loss
    lead, batch, img = shape
    g = torch.Generator().manual_seed(11)
    xo = torch.randn(*lead, *batch, *img, generator=g)
    xt = torch.rand(*batch, *img, generator=g)
    a = xo.clone().requires_grad_(True)
    ref = F.mse_loss(a, xt.expand_as(a), reduction='none').flatten(len(lead) + len(batch)).mean(-1)
    gw = torch.randn(ref.shape, generator=g)
    (ref * gw).sum().backward()
    ad = xo.to(DEV).requires_grad_(True)
    out = mse_loss(ad, xt.to(DEV), ndim=len(img), batch_mean=False)
    assert tuple(out.shape) == tuple(ref.shape) and rel(out, ref) < 1e-5
    (out * gw.to(DEV)).sum().backward()
    assert rel(ad.grad, a.grad) < 1e-5
    assert abs(float(mse_loss(ad.detach(), xt.to(DEV), ndim=len(img))) - float(ref.mean())) < 1e-5 * float(ref.mean())


def test_cross_entropy():
    from jvae_hip import ops
    g = torch.Generator().manual_seed(9)
    L, N, C = 3, 10, 100
    lg = torch.randn(L, N, C, generator=g) * 3
    y = torch.randint(0, C, (N,), generator=g)
    a = lg.clone().requires_grad_(True)
    ref = F.cross_entropy(a.reshape(-1, C), y.repeat(L), reduction='none').reshape(L, N)
    gw = torch.randn(L, N, generator=g)
    (ref * gw).sum().backward()
    d = lg.to(DEV).requires_grad_(True)
    out = ops.cross_entropy_rows(d, y.to(DEV))
    assert rel(out, ref) < 1e-5
    (out * gw.to(DEV)).sum().backward()
    assert rel(d.grad, a.grad) < 1e-5


def test_clip_and_adam_against_torch():
    from jvae_hip import ops
    g = torch.Generator().manual_seed(3)
    n = 100003
    p0 = torch.randn(n, generator=g)
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pr], lr=1e-3, weight_decay=3e-5)
    pd = torch.zeros(n + 1, device=DEV)[:n]      # deliberately a 16-byte aligned base
    pd.copy_(p0)
    m, v = torch.zeros_like(pd), torch.zeros_like(pd)
    acc = torch.zeros(1, device=DEV)
    for step in range(1, 4):
        gr = torch.randn(n, generator=g) * (50 if step == 1 else 0.01)     # step 1 clips, later ones do not
        pr.grad = gr.clone()
        tn = torch.nn.utils.clip_grad_norm_([pr], 100.)
        opt.step()
        gd = gr.to(DEV)
        ops.sqnorm_accum(gd, acc, reset=True)
        assert abs(float(acc.sqrt()) - float(tn)) < 1e-4 * float(tn)
        flag = torch.zeros(1, dtype=torch.int32, device=DEV)
        ops.adam_step(pd, gd, m, v, 1e-3, 0.9, 0.999, 1e-8, 3e-5, step, max_norm=100., sqnorm=acc, flag=flag)
        assert rel(pd, pr) < 1e-6 and int(flag) == 0
    gd = torch.full((n,), float('nan'), device=DEV)
    ops.adam_step(pd, gd, m, v, 1e-3, 0.9, 0.999, 1e-8, 3e-5, 4, flag=flag)
    assert int(flag) == 1


@pytest.mark.parametrize('N,H,C,pad', [(64, 32, 3, 4), (7, 28, 1, 3), (5, 64, 3, 8)])
def test_input_pipeline_bit_exact(N, H, C, pad):
    """flip + edge-pad crop + /255 on the device equals the numpy restatement of the reference's
    torchvision transform chain bit for bit (byte / index work)."""
    import numpy as np
    from jvae_hip import ops
    from oracle.augment_oracle import augment
    rng = np.random.default_rng(N * 100 + H)
    imgs = rng.integers(0, 256, size=(N, H, H, C), dtype=np.uint8)
    flip = rng.integers(0, 2, size=N).astype(bool)
    dy = rng.integers(0, 2 * pad + 1, size=N).astype(np.int32)
    dx = rng.integers(0, 2 * pad + 1, size=N).astype(np.int32)
    dy[0], dx[0], dy[-1], dx[-1] = 0, 0, 2 * pad, 2 * pad            # the extreme crops (pure edge replication)
    ref = augment(imgs, flip, dy, dx, pad)
    out = ops.augment_batch(torch.from_numpy(imgs).to(DEV), torch.from_numpy(flip).to(DEV), torch.from_numpy(dy).to(DEV),
                            torch.from_numpy(dx).to(DEV), pad=pad)
    assert out.shape == (N, C, H, H) and out.dtype == torch.float32
    assert np.array_equal(out.cpu().numpy(), ref)
    ident = ops.augment_batch(torch.from_numpy(imgs).to(DEV))
    assert np.array_equal(ident.cpu().numpy(), imgs.transpose(0, 3, 1, 2).astype(np.float32) / np.float32(255))
    f, a, b = ops.draw_augmentation(N, pad, torch.device(DEV))
    assert f.shape == (N,) and int(a.max()) <= 2 * pad and int(b.min()) >= 0


@pytest.mark.parametrize('mode,K,S,P,H,W', [('max', 2, 2, 0, 32, 32), ('max', 3, 2, 1, 17, 13), ('avg', 2, 2, 0, 16, 16),
                                            ('avg', 3, 1, 1, 9, 11), ('avg', 1, 1, 0, 1, 1), ('max', 2, 2, 0, 5, 7)])
def test_pool2d(mode, K, S, P, H, W):
    """Tokens M / A of the layer DSL (reference conv.py:201-206): nn.MaxPool2d / nn.AvgPool2d forward and backward,
    bit-exact (max: values and routing; avg: same divisor and summation order up to 1 ulp)."""
    from jvae_hip import ops
    g = torch.Generator().manual_seed(K * 100 + S * 10 + P + H)
    x = torch.randn(3, 5, H, W, generator=g)
    x[0, 0, 0, :2] = 1.5                                  # a tie: the first maximum wins
    xr = x.clone().requires_grad_(True)
    f = F.max_pool2d if mode == 'max' else F.avg_pool2d
    yr = f(xr, K, S, P)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = x.to(DEV).requires_grad_(True)
    yd = ops.pool2d(xd, K, S, P, ops.POOL_MAX if mode == 'max' else ops.POOL_AVG)
    assert yd.shape == yr.shape
    yd.backward(gy.to(DEV))
    if mode == 'max':
        assert torch.equal(yd.detach().cpu(), yr.detach()) and torch.equal(xd.grad.cpu(), xr.grad)
    else:
        assert rel(yd, yr) < 1e-6 and rel(xd.grad, xr.grad) < 1e-6


@pytest.mark.parametrize('scale,H,W', [(2, 4, 4), (2, 16, 16), (3, 5, 7)])
def test_upsample_nearest(scale, H, W):
    """Token U (reference conv.py:208-212): nn.UpsamplingNearest2d forward (a copy) and backward (block sums)."""
    from jvae_hip import ops
    g = torch.Generator().manual_seed(scale + H)
    x = torch.randn(2, 6, H, W, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr, scale_factor=scale, mode='nearest')
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = x.to(DEV).requires_grad_(True)
    yd = ops.upsample_nearest(xd, scale)
    yd.backward(gy.to(DEV))
    assert torch.equal(yd.detach().cpu(), yr.detach())
    assert rel(xd.grad, xr.grad) < 1e-6


@pytest.mark.parametrize('momentum,nesterov,wd', [(0., False, 0.), (0.9, False, 3e-5), (0.9, True, 1e-4)])
def test_sgd_matches_torch(momentum, nesterov, wd):
    """Optimizer(optim_type='sgd') (reference optimizers.py:39-40 -> torch.optim.SGD) with clip_grad_norm_, 4 steps."""
    from module.optimizers import Optimizer
    g = torch.Generator().manual_seed(5)
    shapes = [(7, 5), (3,), (2, 3, 5, 5)]
    init = [torch.randn(s, generator=g) for s in shapes]
    ref = [torch.nn.Parameter(t.clone()) for t in init]
    dev = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    kw = dict(momentum=momentum, nesterov=nesterov) if momentum else {}
    topt = torch.optim.SGD(ref, lr=0.05, weight_decay=wd, **kw)
    opt = Optimizer(dev, optim_type='sgd', lr=0.05, weight_decay=wd, grad_clipping=1.5, **kw)
    for step in range(4):
        grads = [torch.randn(s, generator=g) * (3. if step == 1 else 0.2) for s in shapes]
        topt.zero_grad()
        opt.zero_grad()
        for p, q, gr in zip(ref, dev, grads):
            p.grad = gr.clone()
            if q.grad is None:
                q.grad = gr.clone().to(DEV)
            else:
                q.grad.copy_(gr.to(DEV))
        torch.nn.utils.clip_grad_norm_(ref, 1.5)
        topt.step()
        opt.clip()
        opt.step()
        for p, q in zip(ref, dev):
            assert rel(q, p) < 2e-6, step
    sd = opt.state_dict()
    assert set(sd['param_groups'][0]) >= {'lr', 'momentum', 'dampening', 'weight_decay', 'nesterov', 'params'}
    if momentum:
        for i, p in enumerate(ref):
            assert rel(sd['state'][i]['momentum_buffer'], topt.state[p]['momentum_buffer']) < 2e-6


def test_categorical_loss():
    """module/losses.py:30-49 of the reference: 256-way pixel cross entropy summed over the image, with gradient."""
    from module.losses import categorical_loss
    g = torch.Generator().manual_seed(11)
    L_, N, C, H, W = 2, 3, 3, 4, 5
    x = torch.rand(N, C, H, W, generator=g)
    out = torch.randn(L_, N, 256, C, H, W, generator=g)
    ref_in = out.clone().requires_grad_(True)
    tgt = (x.expand(L_, N, C, H, W) * 255).long().view(-1, C, H, W)
    ref = F.cross_entropy(ref_in.view(-1, 256, C, H, W), tgt, reduction='none').view(L_, N, -1).sum(-1)
    gy = torch.randn(L_, N, generator=g)
    ref.backward(gy)
    od = out.to(DEV).requires_grad_(True)
    got = categorical_loss(od, x.to(DEV), ndim=3, batch_mean=False)
    assert got.shape == ref.shape and rel(got, ref) < 2e-6
    got.backward(gy.to(DEV))
    assert rel(od.grad, ref_in.grad) < 2e-5
    assert abs(float(categorical_loss(od.detach(), x.to(DEV))) - float(ref.mean())) < 1e-3


def test_dropout_kernel_and_module():
    """nn.Dropout of the dense trunks (reference layers.py:287-288): masked scaling, same mask in backward, identity in
    eval mode; the random stream is the kernel's own, so the check is on the distribution."""
    from jvae_hip import ops
    from module.vae_layers.layers import HipDropout
    x = torch.randn(512, 300, device=DEV).abs() + 0.1
    for p in (0.1, 0.5):
        xr = x.clone().requires_grad_(True)
        y = ops.dropout(xr, p, 12345)
        kept = y != 0
        frac = float(kept.float().mean())
        assert abs(frac - (1 - p)) < 0.01, (p, frac)
        assert rel(y[kept], x[kept] / (1 - p)) < 1e-6
        assert torch.equal(y, ops.dropout(x, p, 12345)) and not torch.equal(y, ops.dropout(x, p, 12346))
        y.sum().backward()
        assert torch.equal(xr.grad != 0, kept) and rel(xr.grad[kept], torch.full_like(xr.grad[kept], 1 / (1 - p))) < 1e-6
        # no structure along rows / columns
        assert float(kept.float().mean(0).std()) < 0.05 and float(kept.float().mean(1).std()) < 0.06
    m = HipDropout(0.3).to(DEV)
    m.eval()
    assert m(x) is x
    m.train()
    torch.manual_seed(7)
    a, b = m(x), m(x)
    m2 = HipDropout(0.3).to(DEV)
    torch.manual_seed(7)                    # same CPU-generator state -> same device seed sequence: repeatable runs
    assert torch.equal(a, m2(x)) and torch.equal(b, m2(x)) and not torch.equal(a, b)
    assert abs(float((a != 0).float().mean()) - 0.7) < 0.01
    # the seed is device-resident and advanced on the stream: a captured graph draws a fresh mask at every replay
    xs = x.clone()
    g = torch.cuda.CUDAGraph()
    m(xs)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        ys = m(xs)
    g.replay(); r1 = ys.clone()
    g.replay(); r2 = ys.clone()
    assert not torch.equal(r1, r2) and abs(float((r2 != 0).float().mean()) - 0.7) < 0.01
    # end to end: an MLP model with dropout in both dense trunks trains (finite, decreasing loss) and evaluates
    from cvae import ClassificationVariationalNetwork as Net
    torch.manual_seed(0)
    net = Net(input_shape=(1, 28, 28), num_labels=10, type='cvae', features=None, upsampler=None, encoder=[64, 32],
              decoder=[32, 64], classifier=[], batch_norm=False, latent_dim=8, dropout=0.2, sigma={'value': 0.5}, gamma=0.,
              output_activation='sigmoid',
              prior=dict(distribution='gaussian', init_mean=0., learned_means=True, var_dim='scalar', freeze_means=0),
              optimizer=dict(optim_type='adam', lr=2e-3)).to(DEV)
    net.train()
    xb, yb = torch.rand(64, 1, 28, 28, device=DEV), torch.randint(0, 10, (64,), device=DEV)
    tot = [float(net.train_step(xb, yb, batch=i)[0]['total'].mean()) for i in range(12)]
    assert all(math.isfinite(t) for t in tot) and tot[-1] < tot[0]
    net.eval()
    a_, b_ = net.evaluate(xb, yb)[2]['total'], net.evaluate(xb, yb)[2]['total']
    assert a_.shape == (64,)


@pytest.mark.parametrize('cin,cout,H,N,tr', [(32, 32, 16, 5, True), (64, 64, 8, 3, True), (24, 32, 8, 1, True),
                                             (64, 32, 32, 2, True), (32, 32, 32, 3, False), (64, 64, 16, 2, False),
                                             # round 5: several tiles per workgroup (4 at 2048 tiles; 2 at 514, the last one half
                                             # empty), 32 -> 64 channels (two K-step-free output blocks), 48 channels (a partial K step)
                                             (32, 32, 16, 1024, True), (64, 64, 8, 1027, True), (32, 64, 16, 4, True),
                                             (48, 32, 8, 5, False)])
def test_split_bf16_stride2_transposed_conv(cin, cout, H, N, tr):
    """conv_t2_x3.hip: ConvTranspose2d(5, stride 2, padding 2, output_padding 1) forward (tr) and the dgrad of
    Conv2d(5, stride 2, padding 2) on the bf16 matrix cores with 3-way operand splitting, against an fp64 reference
    and against the fp32-MFMA 4-phase kernel (odd batch sizes: partially filled tiles)."""
    from jvae_hip import lib, ops
    L = lib.load()
    g = torch.Generator().manual_seed(cin + cout + H + N)
    x = torch.randn(N, cin, H, H, generator=g) * torch.exp(torch.randn(N, cin, 1, 1, generator=g))
    if tr:       # forward of the transposed layer: small (cin) -> big (cout)
        w = torch.randn(cin, cout, 5, 5, generator=g) / math.sqrt(cin * 25)
        b = torch.randn(cout, generator=g)
        ref = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2, padding=2, output_padding=1)
        spec = ops.ConvSpec(cin, cout, 5, 2, 2, 1, True)
        run = lambda: ops.conv_fwd_stats_raw(x.to(DEV), w.to(DEV), b.to(DEV), spec)[0]
    else:        # dgrad of the stride-2 convolution: dy (cin = its output channels) -> dx (cout = its input channels)
        w = torch.randn(cin, cout, 5, 5, generator=g) / math.sqrt(cin * 25)
        ref = F.conv_transpose2d(x.double(), w.double(), stride=2, padding=2, output_padding=1)
        spec = ops.ConvSpec(cout, cin, 5, 2, 2, 0, False)
        run = lambda: ops.conv_dgrad_raw(x.to(DEV), w.to(DEV), spec, (N, cout, 2 * H, 2 * H))
    old = L.jvae_conv2d_set_split_bf16(1)
    try:
        y1 = run()
        L.jvae_conv2d_set_split_bf16(0)
        y0 = run()
    finally:
        L.jvae_conv2d_set_split_bf16(old)
    assert y1.shape == ref.shape
    assert rel(y1, ref) < 3e-6 and rel(y0, ref) < 5e-6 and rel(y1, y0) < 5e-6


@pytest.mark.parametrize('N,cin,cout,H,s,tr', [(256, 32, 32, 32, 1, True), (256, 64, 32, 16, 1, True), (128, 64, 64, 8, 1, True),
                                               (128, 32, 64, 16, 1, False), (8, 32, 768, 32, 1, False),
                                               (128, 32, 32, 32, 2, False), (256, 64, 64, 8, 2, True),
                                               # several tiles per workgroup: conv5_x3_kernel takes two from 2048 workgroups on
                                               # (ADVICE r4), the round-5 4-phase kernel up to four while 512 workgroups remain
                                               (256, 32, 64, 32, 1, False), (1024, 32, 32, 16, 2, True), (512, 64, 64, 8, 2, True)])
def test_conv_kernels_are_run_to_run_deterministic(N, cin, cout, H, s, tr):
    """Every launch of the same convolution on the same data gives the same BITS - forward with and without the deferred
    BatchNorm (with its BatchNorm sums), dgrad, weight gradient with the deferred BatchNorm, in fp32 and (5x5 bf16 kernels) on
    B8 operands.  Grids with several workgroups per CU, launched six times each.  (Round 4: a staging variant that read the
    deferred BatchNorm's coefficients as 16-byte LDS vectors was 6 % faster and changed a few thousand to a few million outputs
    from launch to launch while every single-launch parity test passed; tools/x3_determinism.py.)"""
    from jvae_hip import ops, ops_b8
    g = torch.Generator().manual_seed(N + cin + cout + H)
    spec = ops.ConvSpec(cin, cout, 5, s, 2, 1 if (tr and s == 2) else 0, tr)
    wshape = (cin, cout, 5, 5) if tr else (cout, cin, 5, 5)
    x = torch.randn(N, cin, H, H, generator=g).to(DEV)
    w = (torch.randn(wshape, generator=g) * 0.05).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    aff = ((torch.rand(cin, generator=g) + 0.5).to(DEV), (torch.randn(cin, generator=g) * 0.3).to(DEV), True)
    y = ops.conv_fwd_raw(x, w, b, spec)
    gy = torch.randn(y.shape, generator=g).to(DEV)

    def wgrad():
        gw = torch.zeros(wshape, device=DEV)
        ops.conv_wgrad_raw(x, gy, spec, wshape, False, gw, None, aff=aff if ops.conv_affine_ok(spec, N, H, H) else None)
        return gw
    cases = {'forward': lambda: ops.conv_fwd_raw(x, w, b, spec),
             'forward with BatchNorm sums': lambda: ops.conv_fwd_stats_raw(x, w, b, spec)[0],
             'dgrad': lambda: ops.conv_dgrad_raw(gy, w, spec, x.shape),
             'weight gradient': wgrad}
    if ops.conv_affine_ok(spec, N, H, H):
        cases['forward, deferred BatchNorm'] = lambda: ops.conv_fwd_aff_raw(x, w, b, spec, aff, True)[0]
        # (round 5: the new 4-phase kernel lost the bias of 16 outputs in a first launch WITHOUT the BatchNorm sums only)
        cases['forward, deferred BatchNorm, no sums'] = lambda: ops.conv_fwd_aff_raw(x, w, b, spec, aff, False)[0]
    if ops_b8.native_mask(spec, N, H, H) == 7 and ops_b8.conv_affine_ok(spec, N, H, H):
        xb, gyb = ops_b8.pack(x), ops_b8.pack(gy)
        C8 = (cin + 7) // 8 * 8
        coef = torch.zeros(2, C8, device=DEV)
        coef[0, :cin], coef[1, :cin] = aff[0], aff[1]
        affb = (coef[0], coef[1], True)
        cases['bf16 forward, deferred BatchNorm'] = lambda: ops_b8.conv_fwd_raw(xb, w, None, spec, aff=affb)[0]
        cases['bf16 weight gradient, deferred BatchNorm'] = lambda: ops_b8.conv_wgrad_raw(xb, gyb, spec, wshape, False, aff=affb)[0]
    for name, f in cases.items():
        first = f().clone()
        for _ in range(5):
            assert torch.equal(f(), first), name
