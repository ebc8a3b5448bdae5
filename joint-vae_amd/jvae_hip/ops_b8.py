"""bf16 activation path ("B8" layout) - wrappers around the *_b8 entry points of libjvae_hip.so.

A B8 tensor is a torch.bfloat16 tensor of shape (N, ceil(C/8), H, W, 8): eight consecutive channels of a pixel are
contiguous (csrc/conv_b8.hip).  Master weights, gradients of parameters, BatchNorm statistics and all loss math stay
fp32.  A layer direction without a native bf16 kernel runs through the fp32 kernels between two layout conversions.
"""
from ctypes import byref, c_int

import torch

from . import lib as L
from . import ops as O

FWD, DGRAD, WGRAD = 1, 2, 4


def cblocks(C):
    return (C + 7) // 8


def pack(x):
    """fp32 (N, C, H, W) -> B8 (N, CB, H, W, 8) bf16."""
    x = O._c(O._f32(x, 'b8.pack'))
    N, C, H, W = x.shape
    y = torch.empty((N, cblocks(C), H, W, 8), device=x.device, dtype=torch.bfloat16)
    L.check(L.load().jvae_b8_pack_f32(L.ptr(x), L.ptr(y), N, C, H * W, L.stream_ptr()), 'jvae_b8_pack_f32')
    return y


def unpack(y, C, out=None, accumulate=False):
    """B8 -> fp32 (N, C, H, W)."""
    N, CB, H, W, e = y.shape
    assert e == 8 and CB == cblocks(C) and y.dtype == torch.bfloat16 and y.is_contiguous()
    if out is None:
        out = torch.empty((N, C, H, W), device=y.device, dtype=torch.float32)
        accumulate = False
    L.check(L.load().jvae_b8_unpack_f32(L.ptr(y), L.ptr(out), N, C, H * W, int(accumulate), L.stream_ptr()),
            'jvae_b8_unpack_f32')
    return out


def native_mask(spec, N, H, W):
    return L.load().jvae_conv2d_native_b8(*spec.geom(N, H, W))


def _ws(geom, device):
    nbytes = L.load().jvae_conv2d_workspace_bytes_b8(*geom)
    ws = L.workspace(max(nbytes, 16), device)
    return ws, ws.numel()


def conv_fwd_raw(x, w, b, spec, out_f32=False, want_stats=False):
    """x: B8.  -> (y, stats, nsplit); y is B8, or fp32 NCHW with out_f32.  Raises JvaeHipError(ENOTSUP) when the geometry
    has no native bf16 kernel (ask native_mask first)."""
    lib = L.load()
    N, _, H, W, _ = x.shape
    geom = spec.geom(N, H, W)
    oh, ow = spec.out_hw(H, W)
    if out_f32:
        y = torch.empty((N, spec.cout, oh, ow), device=x.device, dtype=torch.float32)
    else:
        y = torch.empty((N, cblocks(spec.cout), oh, ow, 8), device=x.device, dtype=torch.bfloat16)
    stats, ns = None, c_int(0)
    if want_stats:
        cap = lib.jvae_conv2d_stats_splits_b8(*geom)
        if cap > 0:
            stats = torch.empty((spec.cout * cap * 2,), device=x.device, dtype=torch.float32)
    ws, nb = _ws(geom, x.device)
    rc = lib.jvae_conv2d_fwd_b8(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), int(out_f32), L.ptr(stats), byref(ns), *geom,
                                L.ptr(ws), nb, L.stream_ptr())
    L.check(rc, 'jvae_conv2d_fwd_b8')
    return y, (stats if ns.value > 0 else None), ns.value


def conv_dgrad_raw(gy, w, spec, N, H, W):
    """gy: B8 of the layer output -> B8 gradient of the layer input (N, cblocks(cin), H, W, 8)."""
    gx = torch.empty((N, cblocks(spec.cin), H, W, 8), device=gy.device, dtype=torch.bfloat16)
    geom = spec.geom(N, H, W)
    ws, nb = _ws(geom, gy.device)
    rc = L.load().jvae_conv2d_dgrad_b8(L.ptr(gy), L.ptr(w), L.ptr(gx), *geom, L.ptr(ws), nb, L.stream_ptr())
    L.check(rc, 'jvae_conv2d_dgrad_b8')
    return gx


def conv_wgrad_raw(x, gy, spec, wshape, want_bias, w_slot=None, b_slot=None):
    """x, gy: B8.  -> (gw, gb) fp32; with slots the result is ADDED into them (see ops.conv_wgrad_raw)."""
    N, _, H, W, _ = x.shape
    inplace = w_slot is not None and (b_slot is not None or not want_bias)
    gw = w_slot if inplace else torch.empty(wshape, device=x.device, dtype=torch.float32)
    gb = None
    if want_bias:
        gb = b_slot if inplace else torch.empty(spec.cout, device=x.device, dtype=torch.float32)
    geom = spec.geom(N, H, W)
    ws, nb = _ws(geom, x.device)
    rc = L.load().jvae_conv2d_wgrad_b8(L.ptr(x), L.ptr(gy), L.ptr(gw), L.ptr(gb), int(inplace), *geom, L.ptr(ws), nb,
                                       L.stream_ptr())
    L.check(rc, 'jvae_conv2d_wgrad_b8')
    return (None, None) if inplace else (gw, gb)
