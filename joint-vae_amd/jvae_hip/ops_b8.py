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
    return O._geom_query('jvae_conv2d_native_b8', spec.geom(N, H, W))


def _ws(geom, device):
    nbytes = O._geom_query('jvae_conv2d_workspace_bytes_b8', geom)
    ws = L.workspace(max(nbytes, 16), device)
    return ws, ws.numel()


def conv_affine_ok(spec, N, H, W):
    """Forward and weight gradient of this layer run on bf16 kernels that can apply a deferred BatchNorm to the input."""
    return bool(O._geom_query('jvae_conv2d_affine_ok_b8', spec.geom(N, H, W)))


def conv_fwd_raw(x, w, b, spec, out_f32=False, want_stats=False, aff=None):
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
        cap = O._geom_query('jvae_conv2d_stats_splits_b8', geom)
        if cap > 0:
            stats = torch.empty((spec.cout * cap * 2,), device=x.device, dtype=torch.float32)
    ws, nb = _ws(geom, x.device)
    if aff is not None:
        rc = lib.jvae_conv2d_fwd_aff_b8(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), int(out_f32), L.ptr(stats), byref(ns),
                                        L.ptr(aff[0]), L.ptr(aff[1]), int(aff[2]), *geom, L.ptr(ws), nb, L.stream_ptr())
    else:
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


def conv_wgrad_raw(x, gy, spec, wshape, want_bias, w_slot=None, b_slot=None, aff=None):
    """x, gy: B8.  -> (gw, gb) fp32; with slots the result is ADDED into them (see ops.conv_wgrad_raw).
    aff = (scale, shift, relu): x is a pre-BatchNorm tensor normalised while it is staged."""
    N, _, H, W, _ = x.shape
    inplace = w_slot is not None and (b_slot is not None or not want_bias)
    gw = w_slot if inplace else torch.empty(wshape, device=x.device, dtype=torch.float32)
    gb = None
    if want_bias:
        gb = b_slot if inplace else torch.empty(spec.cout, device=x.device, dtype=torch.float32)
    geom = spec.geom(N, H, W)
    ws, nb = _ws(geom, x.device)
    if aff is not None:
        rc = L.load().jvae_conv2d_wgrad_aff_b8(L.ptr(x), L.ptr(gy), L.ptr(gw), L.ptr(gb), int(inplace),
                                               L.ptr(aff[0]), L.ptr(aff[1]), int(aff[2]), *geom, L.ptr(ws), nb,
                                               L.stream_ptr())
    else:
        rc = L.load().jvae_conv2d_wgrad_b8(L.ptr(x), L.ptr(gy), L.ptr(gw), L.ptr(gb), int(inplace), *geom, L.ptr(ws), nb,
                                           L.stream_ptr())
    L.check(rc, 'jvae_conv2d_wgrad_b8')
    return (None, None) if inplace else (gw, gb)


# ----------------------------------------------------------------------------------------- autograd layer
def is_b8(t):
    return t.dtype == torch.bfloat16 and t.dim() == 5 and t.shape[-1] == 8


class _Pack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.C = x.shape[1]
        return pack(x)

    @staticmethod
    def backward(ctx, gy):
        return unpack(O._c(gy), ctx.C)


class _Unpack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xb, C):
        return unpack(xb, C)

    @staticmethod
    def backward(ctx, gy):
        return pack(gy), None


def to_b8(x):
    return _Pack.apply(x)


def from_b8(xb, C):
    return _Unpack.apply(xb, C)


class _ConvB8(torch.autograd.Function):
    """nn.Conv2d / nn.ConvTranspose2d on a B8 input.  Output: B8, or fp32 NCHW with out_f32 (the last layer of a stack).
    Directions without a native bf16 kernel take the fp32 kernels between two layout conversions."""

    @staticmethod
    def forward(ctx, x, w, b, spec, dead_bias, stats_out, out_f32, aff=None):
        N, _, H, W, _ = x.shape
        mask = native_mask(spec, N, H, W)
        ctx.w_ref, ctx.b_ref = w, b
        w = O._c(w)
        want_stats = stats_out is not None
        ctx.aff = aff                       # only given where conv_affine_ok(): forward and wgrad are native
        if mask & FWD and not (out_f32 and spec.transposed and spec.s == 2):
            y, st, ns = conv_fwd_raw(x, w, b, spec, out_f32=out_f32, want_stats=want_stats, aff=aff)
        else:
            x32 = unpack(x, spec.cin)
            if want_stats:
                y32, st, ns = O.conv_fwd_stats_raw(x32, w, b, spec)
            else:
                y32, st, ns = O.conv_fwd_raw(x32, w, b, spec), None, 0
            y = y32 if out_f32 else pack(y32)
        if want_stats:
            stats_out['stats'], stats_out['nsplit'], stats_out['pivot'] = st, ns, b
        ctx.save_for_backward(x, w)
        ctx.spec, ctx.mask, ctx.out_f32 = spec, mask, out_f32
        ctx.has_bias = b is not None
        ctx.dead_bias = dead_bias
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        spec, mask = ctx.spec, ctx.mask
        N, _, H, W, _ = x.shape
        gy = O._c(gy)
        cache = {'b8': None if ctx.out_f32 else gy, 'f32': gy if ctx.out_f32 else None}

        def g_b8():
            if cache['b8'] is None:
                cache['b8'] = pack(cache['f32'])
            return cache['b8']

        def g_f32():
            if cache['f32'] is None:
                cache['f32'] = unpack(cache['b8'], spec.cout)
            return cache['f32']

        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if mask & DGRAD:
                gx = conv_dgrad_raw(g_b8(), w, spec, N, H, W)
            else:
                gx = pack(O.conv_dgrad_raw(g_f32(), w, spec, (N, spec.cin, H, W)))
        want_b = ctx.has_bias and ctx.needs_input_grad[2] and not ctx.dead_bias
        if ctx.needs_input_grad[1] or want_b:
            w_slot = O._grad_slot(ctx.w_ref)
            b_slot = O._grad_slot(ctx.b_ref) if want_b else None
            native = bool(mask & WGRAD)
            if native:
                gyw = g_b8()
                run = lambda ws_, bs_: conv_wgrad_raw(x, gyw, spec, w.shape, want_b, ws_, bs_, ctx.aff)
            else:
                gyw = g_f32()
                x32 = unpack(x, spec.cin)
                run = lambda ws_, bs_: O.conv_wgrad_raw(x32, gyw, spec, w.shape, want_b, ws_, bs_)
            if O.OVERLAP_WGRAD and w_slot is not None and (b_slot is not None or not want_b):
                main = torch.cuda.current_stream(x.device)
                side = L.side_stream(x.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    run(w_slot, b_slot)
                x.record_stream(side)
                gyw.record_stream(side)
                if ctx.aff is not None:
                    ctx.aff[0].record_stream(side)
                if not native:
                    x32.record_stream(side)
                O._join_after_backward()
            else:
                gw, gb = run(w_slot, b_slot)
        if ctx.dead_bias and ctx.has_bias and ctx.needs_input_grad[2] and O._grad_slot(ctx.b_ref) is None:
            gb = torch.zeros_like(ctx.b_ref)
        return gx, gw, gb, None, None, None, None, None


def conv2d(x, w, b, spec, dead_bias=False, stats_out=None, out_f32=False, aff=None):
    return _ConvB8.apply(x, w, b, spec, dead_bias, stats_out, out_f32, aff)


class _BatchNormActB8(torch.autograd.Function):
    """nn.BatchNorm2d (train or eval) + optional ReLU on a B8 tensor; statistics and parameters fp32."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, nbt, training, relu, momentum, eps, ext, C):
        x = O._c(x)
        N, CB, H, W, _ = x.shape
        HW = H * W
        lib = L.load()
        y = torch.empty_like(x)
        mean = torch.empty(C, device=x.device, dtype=torch.float32)
        invstd = torch.empty(C, device=x.device, dtype=torch.float32)
        ws = L.workspace(lib.jvae_bn_workspace_bytes_b8(C), x.device)
        use_ext = ext is not None and ext.get('stats') is not None and training
        rc = lib.jvae_bn_fwd_b8(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv), L.ptr(nbt), L.ptr(y),
                                L.ptr(mean), L.ptr(invstd), N, C, HW, momentum, eps, int(training), int(relu),
                                L.ptr(ext['stats']) if use_ext else None, int(ext['nsplit']) if use_ext else 0,
                                L.ptr(ext.get('pivot')) if use_ext else None, L.ptr(ws), ws.numel(), L.stream_ptr())
        L.check(rc, 'jvae_bn_fwd_b8')
        if training:
            ctx.save_for_backward(x, gamma, beta, mean, invstd)
            ctx.relu = relu
            ctx.dims = (N, C, HW)
            ctx.g_ref, ctx.b_ref = gamma, beta
        else:
            ctx.dims = None
        return y

    @staticmethod
    def backward(ctx, gy):
        if ctx.dims is None:
            raise L.JvaeHipError('backward through eval-mode BatchNorm is not part of the training step')
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        N, C, HW = ctx.dims
        gy = O._c(gy)
        lib = L.load()
        gx = torch.empty_like(x)
        sg, sb = O._grad_slot(ctx.g_ref), O._grad_slot(ctx.b_ref)
        inplace = sg is not None and sb is not None
        gg = sg if inplace else torch.empty(C, device=x.device, dtype=torch.float32)
        gb = sb if inplace else torch.empty(C, device=x.device, dtype=torch.float32)
        ws = L.workspace(lib.jvae_bn_workspace_bytes_b8(C), x.device)
        rc = lib.jvae_bn_bwd_b8(L.ptr(gy), L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(mean), L.ptr(invstd), L.ptr(gx),
                                L.ptr(gg), L.ptr(gb), int(inplace), N, C, HW, int(ctx.relu), L.ptr(ws), ws.numel(),
                                L.stream_ptr())
        L.check(rc, 'jvae_bn_bwd_b8')
        if inplace:
            gg = gb = None
        return gx, gg, gb, None, None, None, None, None, None, None, None, None


class _SyncBatchNormActB8(torch.autograd.Function):
    """Train-mode BatchNorm2d (+ReLU) on a B8 tensor whose statistics span all data-parallel ranks (ops._SyncBatchNormAct
    for the bf16 layout): two (C,2) fp32 all-reduces per layer, one in forward, one in backward."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, nbt, relu, momentum, eps, world, group, C):
        import torch.distributed as dist
        x = O._c(x)
        N, CB, H, W, _ = x.shape
        HW = H * W
        lib = L.load()
        ws = L.workspace(lib.jvae_bn_workspace_bytes_b8(C), x.device)
        pivot = rm.detach().clone()                       # identical on every rank; rm itself is updated by the kernel
        sums = torch.empty((C, 2), device=x.device, dtype=torch.float32)
        L.check(lib.jvae_bn_sums_b8(L.ptr(x), L.ptr(pivot), L.ptr(sums), N, C, HW, L.ptr(ws), ws.numel(), L.stream_ptr()),
                'jvae_bn_sums_b8')
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        y = torch.empty_like(x)
        mean = torch.empty(C, device=x.device, dtype=torch.float32)
        invstd = torch.empty(C, device=x.device, dtype=torch.float32)
        rc = lib.jvae_bn_fwd_sync_b8(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv), L.ptr(nbt), L.ptr(y),
                                     L.ptr(mean), L.ptr(invstd), N, C, HW, momentum, eps, int(relu), L.ptr(sums),
                                     L.ptr(pivot), int(world), L.ptr(ws), ws.numel(), L.stream_ptr())
        L.check(rc, 'jvae_bn_fwd_sync_b8')
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        ctx.cfg = (N, C, HW, relu, int(world), group)
        ctx.g_ref, ctx.b_ref = gamma, beta
        return y

    @staticmethod
    def backward(ctx, gy):
        import torch.distributed as dist
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        N, C, HW, relu, world, group = ctx.cfg
        gy = O._c(gy)
        lib = L.load()
        ws = L.workspace(lib.jvae_bn_workspace_bytes_b8(C), x.device)
        local = torch.empty((C, 2), device=x.device, dtype=torch.float32)
        rc = lib.jvae_bn_bwd_sums_b8(L.ptr(gy), L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(mean), L.ptr(invstd),
                                     L.ptr(local), N, C, HW, int(relu), L.ptr(ws), ws.numel(), L.stream_ptr())
        L.check(rc, 'jvae_bn_bwd_sums_b8')
        glob = local.clone()
        dist.all_reduce(glob, op=dist.ReduceOp.SUM, group=group)
        gx = torch.empty_like(x)
        sg, sb = O._grad_slot(ctx.g_ref), O._grad_slot(ctx.b_ref)
        inplace = sg is not None and sb is not None
        gg = sg if inplace else torch.empty(C, device=x.device, dtype=torch.float32)
        gb = sb if inplace else torch.empty(C, device=x.device, dtype=torch.float32)
        rc = lib.jvae_bn_bwd_sync_b8(L.ptr(gy), L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(mean), L.ptr(invstd),
                                     L.ptr(local), L.ptr(glob), world, L.ptr(gx), L.ptr(gg), L.ptr(gb), int(inplace),
                                     N, C, HW, int(relu), L.ptr(ws), ws.numel(), L.stream_ptr())
        L.check(rc, 'jvae_bn_bwd_sync_b8')
        if inplace:
            gg = gb = None
        return gx, gg, gb, None, None, None, None, None, None, None, None, None


def sync_batchnorm_act(x, C, gamma, beta, running_mean, running_var, num_batches_tracked, relu, momentum, eps, world, group=None):
    """Synchronised train-mode BatchNorm(+ReLU) on a B8 tensor (statistics over all data-parallel ranks)."""
    return _SyncBatchNormActB8.apply(x, gamma, beta, running_mean, running_var, num_batches_tracked, relu, momentum, eps,
                                     world, group, C)


class _BatchNormDeferB8(torch.autograd.Function):
    """BatchNorm(+ReLU) on a B8 tensor deferred into the next bf16 convolution (see ops._BatchNormDefer): forward produces
    the statistics and the (scale, shift) rows only and returns the input itself; backward is the ordinary one."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, nbt, training, relu, momentum, eps, ext, C):
        x = O._c(x)
        N, CB, H, W, _ = x.shape
        HW = H * W
        lib = L.load()
        mean = torch.empty(C, device=x.device, dtype=torch.float32)
        invstd = torch.empty(C, device=x.device, dtype=torch.float32)
        coef = torch.empty((2, CB * 8), device=x.device, dtype=torch.float32)
        ws = L.workspace(lib.jvae_bn_workspace_bytes_b8(C), x.device)
        use_ext = ext is not None and ext.get('stats') is not None and training
        rc = lib.jvae_bn_finalize_b8(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv), L.ptr(nbt),
                                     L.ptr(mean), L.ptr(invstd), L.ptr(coef), N, C, HW, momentum, eps, int(training),
                                     L.ptr(ext['stats']) if use_ext else None, int(ext['nsplit']) if use_ext else 0,
                                     L.ptr(ext.get('pivot')) if use_ext else None, L.ptr(ws), ws.numel(), L.stream_ptr())
        L.check(rc, 'jvae_bn_finalize_b8')
        if training:
            ctx.save_for_backward(x, gamma, beta, mean, invstd)
            ctx.relu = relu
            ctx.dims = (N, C, HW)
            ctx.g_ref, ctx.b_ref = gamma, beta
        else:
            ctx.dims = None
        ctx.mark_non_differentiable(coef)
        ctx.set_materialize_grads(False)        # no zero-fill launch for the (never used) gradient of `coef`
        return x.view_as(x), coef

    @staticmethod
    def backward(ctx, gy, _gcoef):
        return _BatchNormActB8.backward(ctx, gy)


def batchnorm_defer(x, C, gamma, beta, running_mean, running_var, num_batches_tracked, training, relu,
                    momentum=0.1, eps=1e-5, ext=None):
    """-> (x_alias, (scale, shift, relu)) for conv2d(..., aff=...) of a layer with conv_affine_ok()."""
    xa, coef = _BatchNormDeferB8.apply(x, gamma, beta, running_mean, running_var, num_batches_tracked, training, relu,
                                       momentum, eps, ext, C)
    return xa, (coef[0], coef[1], relu)


def batchnorm_act(x, C, gamma, beta, running_mean, running_var, num_batches_tracked, training, relu,
                  momentum=0.1, eps=1e-5, ext=None):
    return _BatchNormActB8.apply(x, gamma, beta, running_mean, running_var, num_batches_tracked, training, relu,
                                 momentum, eps, ext, C)


class _ReluB8(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = O._c(x)
        y = torch.empty_like(x)
        L.check(L.load().jvae_relu_fwd_b8(L.ptr(x), L.ptr(y), x.numel() // 8, L.stream_ptr()), 'jvae_relu_fwd_b8')
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, = ctx.saved_tensors
        gy = O._c(gy)
        gx = torch.empty_like(gy)
        L.check(L.load().jvae_relu_bwd_b8(L.ptr(gy), L.ptr(y), L.ptr(gx), y.numel() // 8, L.stream_ptr()), 'jvae_relu_bwd_b8')
        return gx


def relu(x):
    return _ReluB8.apply(x)
