"""GPU parity of the bf16 ("B8" layout) kernels.  Reference = PyTorch-CPU fp32 of the same op evaluated on the
bf16-ROUNDED operands: products of bf16 numbers are exact in fp32 and the kernels accumulate in fp32, so only the
summation order (<= ~1e-5 relative) and the final rounding of a bf16 output (2^-9 relative per element) differ.
Tolerances: fp32 outputs 2e-5 of the output scale; bf16 outputs 2^-8 (0.4 %) of the output scale."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'
BF_TOL = 2.0 ** -8


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def rbf(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize('N,C,H', [(3, 3, 8), (2, 32, 16), (5, 200, 6), (1, 17, 5)])
def test_pack_unpack_roundtrip(N, C, H):
    from jvae_hip import ops_b8
    g = torch.Generator().manual_seed(N + C + H)
    x = torch.randn(N, C, H, H, generator=g)
    xb = ops_b8.pack(x.to(DEV))
    assert xb.shape == (N, (C + 7) // 8, H, H, 8) and xb.dtype == torch.bfloat16
    ref = torch.zeros(N, xb.shape[1] * 8, H, H)
    ref[:, :C] = rbf(x)
    lay = xb.float().cpu().permute(0, 1, 4, 2, 3).reshape(N, -1, H, H)     # (N, CB, 8, H, W) -> channels
    assert torch.equal(lay, ref)
    back = ops_b8.unpack(xb, C)
    assert torch.equal(back.cpu(), rbf(x))
    acc = torch.ones(N, C, H, H, device=DEV)
    ops_b8.unpack(xb, C, out=acc, accumulate=True)
    assert torch.equal(acc.cpu(), rbf(x) + 1)


B8_CONVS = [  # (cin, cout, k, s, p, op, transposed, H): the 5x5 layers of conv32(+) / deconv32(+)
    (3, 32, 5, 1, 2, 0, False, 32), (3, 32, 5, 1, 2, 0, False, 64), (32, 32, 5, 2, 2, 0, False, 32),
    (32, 32, 5, 2, 2, 0, False, 64), (32, 64, 5, 1, 2, 0, False, 16), (64, 64, 5, 2, 2, 0, False, 16),
    (64, 128, 5, 1, 2, 0, False, 16), (128, 128, 5, 2, 2, 0, False, 16), (64, 64, 5, 1, 2, 0, True, 8),
    (128, 128, 5, 1, 2, 0, True, 8), (64, 32, 5, 1, 2, 0, True, 16), (32, 32, 5, 1, 2, 0, True, 32),
    (32, 32, 5, 1, 2, 0, True, 64), (32, 3, 5, 1, 2, 0, False, 32), (32, 3, 5, 1, 2, 0, False, 64),
    (64, 64, 5, 2, 2, 1, True, 8), (32, 32, 5, 2, 2, 1, True, 16), (32, 32, 5, 2, 2, 1, True, 32),
    (24, 40, 5, 1, 2, 0, False, 16),
]


@pytest.mark.parametrize('cin,cout,k,s,p,op,tr,H', B8_CONVS)
@pytest.mark.parametrize('N', [3, 8])
def test_b8_conv_native_directions(cin, cout, k, s, p, op, tr, H, N):
    from jvae_hip import ops, ops_b8
    g = torch.Generator().manual_seed(cin * 131 + cout * 17 + k + H)
    x = rbf(torch.randn(N, cin, H, H, generator=g))
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    w = torch.randn(wshape, generator=g) / math.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g)
    wr = rbf(w)                                    # the kernel rounds the fp32 master weights to bf16
    spec = ops.ConvSpec(cin, cout, k, s, p, op, tr)
    mask = ops_b8.native_mask(spec, N, H, H)
    conv = (lambda t, ww, bb: F.conv_transpose2d(t, ww, bb, stride=s, padding=p, output_padding=op)) if tr else \
           (lambda t, ww, bb: F.conv2d(t, ww, bb, stride=s, padding=p))
    xr = x.clone().requires_grad_(True)
    yr = conv(xr, wr, b)
    xb = ops_b8.pack(x.to(DEV))
    wd, bd = w.to(DEV), b.to(DEV)
    if mask & ops_b8.FWD:
        f32_out = not (tr and s == 2)              # the 4-phase transposed kernel only writes B8 (always feeds a BatchNorm)
        if f32_out:
            y32, st, ns = ops_b8.conv_fwd_raw(xb, wd, bd, spec, out_f32=True, want_stats=True)
            assert rel(y32, yr) < 2e-5
        yb, st2, ns2 = ops_b8.conv_fwd_raw(xb, wd, bd, spec, want_stats=True)
        if not f32_out:
            st, ns = st2, ns2
        assert ns > 0
        part = st[:cout * ns * 2].view(cout, ns, 2).double().sum(1).cpu()      # layout (Cout, nsplit, 2)
        d = (yr.detach() - b.view(1, -1, 1, 1)).double()
        assert torch.allclose(part[:, 0], d.sum((0, 2, 3)), rtol=1e-4, atol=1e-3 * float(d.abs().sum((0, 2, 3)).max()))
        assert torch.allclose(part[:, 1], (d * d).sum((0, 2, 3)), rtol=1e-4)
        assert yb.dtype == torch.bfloat16 and yb.shape == (N, (cout + 7) // 8, yr.shape[2], yr.shape[3], 8)
        assert rel(ops_b8.unpack(yb, cout), yr) < BF_TOL
        pad = yb.float().cpu().permute(0, 1, 4, 2, 3).reshape(N, -1, yr.shape[2], yr.shape[3])[:, cout:]
        assert float(pad.abs().max()) == 0. if pad.numel() else True      # padding channels stay exactly zero
    gy = rbf(torch.randn(yr.shape, generator=g))
    wr_ = wr.clone().requires_grad_(True)
    br_ = b.clone().requires_grad_(True)
    conv(x, wr_, br_).backward(gy)
    if mask & ops_b8.WGRAD:
        gyb = ops_b8.pack(gy.to(DEV))
        gw, gb = ops_b8.conv_wgrad_raw(xb, gyb, spec, wshape, True)
        assert rel(gw, wr_.grad) < 3e-5
        assert rel(gb, br_.grad) < 3e-5
        slot_w, slot_b = torch.ones(wshape, device=DEV), torch.ones(cout, device=DEV)
        ops_b8.conv_wgrad_raw(xb, gyb, spec, wshape, True, slot_w, slot_b)       # accumulate in place
        assert rel(slot_w - 1, wr_.grad) < 3e-5 and rel(slot_b - 1, br_.grad) < 3e-5
    if mask & ops_b8.DGRAD:
        yr.backward(gy)
        gx = ops_b8.conv_dgrad_raw(ops_b8.pack(gy.to(DEV)), wd, spec, N, H, H)
        assert rel(ops_b8.unpack(gx, cin), xr.grad) < BF_TOL
    expected = 7
    assert mask & expected == expected, (mask, expected)
