"""Autograd-aware wrappers around the HIP kernels of libjvae_hip.so.

PyTorch is plumbing here (device memory, streams, the autograd tape); every FLOP of the training step is
executed by the hand-written gfx950 kernels.  Every function raises if its tensors are not on the GPU.
"""
import math
from ctypes import byref, c_int

import torch

from . import lib as L

RELU, SIGMOID, IDENT, LEAKY = 1, 2, 0, 3          # jvae_act_* kinds (LEAKY: nn.LeakyReLU(), negative slope 0.01)
# the `relu` / `in_relu` argument of the BatchNorm and convolution entry points: 0 none, 1 ReLU, 2 leaky ReLU (jvae_hip.h)
BN_ACT = {IDENT: 0, RELU: 1, LEAKY: 2}
OVERLAP_WGRAD = True          # weight-gradient kernels on a second HIP stream (see _Conv.backward)
ACT_KIND = {'relu': RELU, 'sigmoid': SIGMOID, 'linear': IDENT, 'leaky': LEAKY, None: IDENT}


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _f32(t, what):
    if t.dtype != torch.float32:
        raise L.JvaeHipError(f'{what}: fp32 expected, got {t.dtype}')
    return t


def _grad_slot(param):
    """The tensor a parameter gradient can be accumulated into in place (its existing, dense .grad), or None.

    When the optimiser has flattened the model every .grad is a view of one zeroed flat buffer: the backward
    kernels then add their result straight into it (exactly what autograd's AccumulateGrad would do with one
    more elementwise kernel per parameter) and hand `None` back to autograd."""
    if param is None or not isinstance(param, torch.nn.Parameter):
        return None
    g = param.grad
    if g is None or not g.is_contiguous() or g.dtype != torch.float32 or g.shape != param.shape or not g.is_cuda:
        return None
    return g


def _join_after_backward():
    """When the autograd engine finishes this backward pass, the main stream waits for the side stream, so that
    whoever reads .grad next (optimiser, all-reduce, a test) sees the finished weight gradients.  Queued once per
    side-stream launch (an event wait each: cheap), so no state survives a backward pass that raised."""
    torch.autograd.Variable._execution_engine.queue_callback(L.join_side_stream)


def _close_span_after_backward():
    """The span of constant weights that a train-mode evaluate() / forward() opened (packed weights served from the step's
    cache, lib.pack_cache_begin) ends with the backward pass that needed it: the engine's end-of-backward callback disarms
    the cache, whether or not an optimiser step follows.  Queued by every convolution backward (a host-side flag flip each)."""
    torch.autograd.Variable._execution_engine.queue_callback(L.pack_cache_end)


# ------------------------------------------------------------------------------------------- gemm
def gemm(M, N, K, A, sA, B, sB, C, sC, bias=None, bias_mode=0, flags=0, splitk=1, batch=1):
    """Raw strided product on the current stream.  sA = (sAm, sAk, sAb) etc. (elements)."""
    lib = L.load()
    rc = lib.jvae_gemm_f32(M, N, K, batch, L.ptr(A), *sA, L.ptr(B), *sB, L.ptr(C), *sC,
                           L.ptr(bias), bias_mode, flags, splitk, L.stream_ptr())
    L.check(rc, 'jvae_gemm_f32')


def channel_sum(t, N, C, P, out=None, accumulate=False):
    lib = L.load()
    if out is None:
        out = torch.empty(C, device=t.device, dtype=torch.float32)
    ws = L.workspace(lib.jvae_channel_sum_workspace_bytes(C), t.device)
    L.check(lib.jvae_channel_sum_f32(L.ptr(t), L.ptr(out), N, C, P, int(accumulate), L.ptr(ws), ws.numel(), L.stream_ptr()),
            'channel_sum')
    return out


def _linear_split(R, O, I):
    """K slices for y = x W^T when the (R, O) output alone cannot fill the chip (64x64 tiles, 256 CUs)."""
    tiles = ((R + 63) // 64) * ((O + 63) // 64)
    if tiles >= 128 or I < 512:
        return 1
    for S in (32, 16, 8, 4, 2):
        if tiles * S <= 1024 and I % S == 0 and I // S >= 128:
            return S
    return 1


class _Linear(torch.autograd.Function):
    """y = x W^T + b (+ReLU / sigmoid).  nn.Linear + activation, layers.py:283-296, cvae.py:291-301."""

    @staticmethod
    def forward(ctx, x, w, b, act):
        x2 = _c(_f32(x, 'linear').reshape(-1, x.shape[-1]))
        w_in = w
        w = _c(w)
        R, I = x2.shape
        O = w.shape[0]
        y = torch.empty((R, O), device=x.device, dtype=torch.float32)
        S = _linear_split(R, O, I)
        if S > 1:
            # few output tiles, long K: K slices as the batch of one launch, then a fixed-order fold (+bias, ReLU)
            part = torch.empty((S, R, O), device=x.device, dtype=torch.float32)
            Ks = I // S
            gemm(R, O, Ks, x2, (I, 1, Ks), w, (1, I, Ks), part, (O, 1, R * O), batch=S)
            L.check(L.load().jvae_splitk_fold_f32(L.ptr(part), L.ptr(b), L.ptr(y), S, R * O, O, int(act == RELU), 0,
                                                  L.stream_ptr()), 'jvae_splitk_fold_f32')
        else:
            gemm(R, O, I, x2, (I, 1, 0), w, (1, I, 0), y, (O, 1, 0), bias=b, bias_mode=1 if b is not None else 0,
                 flags=2 if act == RELU else 0)
        if act in (SIGMOID, LEAKY):           # (ReLU rides in the product's epilogue; these two are a pass of their own, in place)
            L.check(L.load().jvae_act_fwd_f32(L.ptr(y), L.ptr(y), y.numel(), act, L.stream_ptr()), 'act_fwd')
        ctx.save_for_backward(x2, w, y if act != IDENT else None)
        ctx.act = act
        ctx.has_bias = b is not None
        ctx.xshape = x.shape
        ctx.w_ref, ctx.b_ref = w_in, b
        return y.reshape(*x.shape[:-1], O)

    @staticmethod
    def backward(ctx, gy):
        x2, w, y = ctx.saved_tensors
        lib = L.load()
        R, I = x2.shape
        O = w.shape[0]
        gy = _c(gy.reshape(R, O))
        if ctx.act != IDENT:
            g = torch.empty_like(gy)
            L.check(lib.jvae_act_bwd_f32(L.ptr(gy), L.ptr(y), L.ptr(g), gy.numel(), ctx.act, L.stream_ptr()), 'act_bwd')
            gy = g
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty((R, I), device=gy.device, dtype=torch.float32)
            gemm(R, I, O, gy, (O, 1, 0), w, (I, 1, 0), gx, (I, 1, 0))
            gx = gx.reshape(ctx.xshape)
        if ctx.needs_input_grad[1]:
            slot = _grad_slot(ctx.w_ref)
            gw = slot if slot is not None else torch.zeros((O, I), device=gy.device, dtype=torch.float32)
            S = _linear_split(O, I, R)
            if S > 1:           # few (O, I) tiles: batch rows sliced over the launch's batch dimension, fixed-order fold
                part = torch.empty((S, O, I), device=gy.device, dtype=torch.float32)
                Rs = R // S
                gemm(O, I, Rs, gy, (1, O, Rs * O), x2, (I, 1, Rs * I), part, (I, 1, O * I), batch=S)
                L.check(lib.jvae_splitk_fold_f32(L.ptr(part), None, L.ptr(gw), S, O * I, I, 0, 1, L.stream_ptr()),
                        'jvae_splitk_fold_f32')
            else:
                gemm(O, I, R, gy, (1, O, 0), x2, (I, 1, 0), gw, (I, 1, 0), flags=1)      # no atomics: deterministic
            if slot is not None:
                gw = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            slot = _grad_slot(ctx.b_ref)
            if slot is not None:
                channel_sum(gy, R, O, 1, out=slot, accumulate=True)
            else:
                gb = channel_sum(gy, R, O, 1)
        return gx, gw, gb, None


def linear(x, w, b=None, act=IDENT):
    return _Linear.apply(x, w, b, act)


# ------------------------------------------------------------------------------------------- conv
_GEOM_CACHE = {}          # pure functions of the layer geometry asked through the C ABI (host-side memo: every op asks)


def _geom_query(name, geom):
    key = (name, geom)
    r = _GEOM_CACHE.get(key)
    if r is None:
        _GEOM_CACHE[key] = r = getattr(L.load(), name)(*geom)
    return r


class ConvSpec:
    __slots__ = ('cin', 'cout', 'k', 's', 'p', 'op', 'transposed')

    def __init__(self, cin, cout, k, s, p, op=0, transposed=False):
        self.cin, self.cout, self.k, self.s, self.p, self.op, self.transposed = cin, cout, k, s, p, op, bool(transposed)

    def out_hw(self, H, W):
        key = ('hw', self.k, self.s, self.p, self.op, self.transposed, H, W)
        r = _GEOM_CACHE.get(key)
        if r is None:
            oh, ow = c_int(), c_int()
            rc = L.load().jvae_conv2d_out_shape(H, W, self.k, self.k, self.s, self.p, self.op, int(self.transposed),
                                                byref(oh), byref(ow))
            L.check(rc, 'jvae_conv2d_out_shape')
            _GEOM_CACHE[key] = r = (oh.value, ow.value)
        return r

    def geom(self, N, H, W):
        return (N, self.cin, H, W, self.cout, self.k, self.k, self.s, self.p, self.op, int(self.transposed))


def _conv_ws(geom, device):
    nbytes = _geom_query('jvae_conv2d_workspace_bytes', geom)
    ws = L.workspace(nbytes, device)
    return ws, ws.numel()


def conv_fwd_raw(x, w, b, spec):
    N, _, H, W = x.shape
    oh, ow = spec.out_hw(H, W)
    y = torch.empty((N, spec.cout, oh, ow), device=x.device, dtype=torch.float32)
    geom = spec.geom(N, H, W)
    ws, nb = _conv_ws(geom, x.device)
    rc = L.load().jvae_conv2d_fwd_f32(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), *geom, L.ptr(ws), nb, L.stream_ptr())
    L.check(rc, 'jvae_conv2d_fwd_f32')
    return y


def conv_fwd_stats_raw(x, w, b, spec):
    """-> (y, stats, nsplit): forward + the BatchNorm partial sums of (y - bias) written by the conv epilogue
    (stats is None / nsplit 0 when the kernel selected for this geometry does not produce them)."""
    lib = L.load()
    N, _, H, W = x.shape
    geom = spec.geom(N, H, W)
    cap = _geom_query('jvae_conv2d_stats_splits', geom)
    if cap <= 0:
        return conv_fwd_raw(x, w, b, spec), None, 0
    oh, ow = spec.out_hw(H, W)
    y = torch.empty((N, spec.cout, oh, ow), device=x.device, dtype=torch.float32)
    stats = torch.empty((spec.cout * cap * 2,), device=x.device, dtype=torch.float32)
    ws, nb = _conv_ws(geom, x.device)
    ns = c_int(0)
    rc = lib.jvae_conv2d_fwd_stats_f32(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(stats), byref(ns), *geom,
                                       L.ptr(ws), nb, L.stream_ptr())
    L.check(rc, 'jvae_conv2d_fwd_stats_f32')
    return y, (stats if ns.value > 0 else None), ns.value


def conv_affine_ok(spec, N, H, W):
    """Can this layer apply a deferred BatchNorm(+ReLU) to its input (forward and weight gradient kernels)?"""
    return bool(_geom_query('jvae_conv2d_affine_ok', spec.geom(N, H, W)))


class LaunchProbe:
    """HIP-event pairs around ONE convolution launch inside the running step (bench.py's `roofline`: the dominant kernel timed
    where it runs - same clocks, caches and neighbours as in the step - on the stream it is launched on).  `match(spec, N, H)`
    selects the launch; events are recorded on the current stream, read after the timed region."""

    def __init__(self, match):
        self.match, self.events, self.armed = match, [], False

    def times_ms(self):
        return [a.elapsed_time(b) for a, b in self.events]


FWD_AFF_PROBE = None          # set by bench.py only


def conv_fwd_aff_raw(x, w, b, spec, aff, want_stats):
    """Forward with a = [relu](x*scale[c] + shift[c]) applied to the input while it is staged; aff = (scale, shift, relu).
    -> (y, stats, nsplit) as conv_fwd_stats_raw."""
    lib = L.load()
    N, _, H, W = x.shape
    geom = spec.geom(N, H, W)
    oh, ow = spec.out_hw(H, W)
    y = torch.empty((N, spec.cout, oh, ow), device=x.device, dtype=torch.float32)
    stats, ns = None, c_int(0)
    if want_stats:
        cap = _geom_query('jvae_conv2d_stats_splits', geom)
        if cap > 0:
            stats = torch.empty((spec.cout * cap * 2,), device=x.device, dtype=torch.float32)
    ws, nb = _conv_ws(geom, x.device)
    probe = FWD_AFF_PROBE
    timed = probe is not None and probe.armed and probe.match(spec, N, H)
    if timed:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    rc = lib.jvae_conv2d_fwd_aff_f32(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(stats), byref(ns),
                                     L.ptr(aff[0]), L.ptr(aff[1]), int(aff[2]), *geom, L.ptr(ws), nb, L.stream_ptr())
    if timed:
        ev[1].record()
        probe.events.append(ev)
    L.check(rc, 'jvae_conv2d_fwd_aff_f32')
    return y, (stats if ns.value > 0 else None), ns.value


def conv_dgrad_raw(gy, w, spec, xshape):
    N, _, H, W = xshape
    gx = torch.empty(xshape, device=gy.device, dtype=torch.float32)
    geom = spec.geom(N, H, W)
    ws, nb = _conv_ws(geom, gy.device)
    rc = L.load().jvae_conv2d_dgrad_f32(L.ptr(gy), L.ptr(w), L.ptr(gx), *geom, L.ptr(ws), nb, L.stream_ptr())
    L.check(rc, 'jvae_conv2d_dgrad_f32')
    return gx


def conv_wgrad_raw(x, gy, spec, wshape, want_bias, w_slot=None, b_slot=None, aff=None):
    """-> (gw, gb).  With w_slot / b_slot (existing .grad tensors) the result is ADDED there and None is returned
    for that gradient.  The C entry point takes one accumulate flag: both slots or neither.
    aff = (scale, shift, relu): x is a pre-BatchNorm tensor whose normalisation is applied while it is staged."""
    N, _, H, W = x.shape
    inplace = w_slot is not None and (b_slot is not None or not want_bias)
    gw = w_slot if inplace else torch.empty(wshape, device=x.device, dtype=torch.float32)
    gb = None
    if want_bias:
        gb = b_slot if inplace else torch.empty(spec.cout, device=x.device, dtype=torch.float32)
    geom = spec.geom(N, H, W)
    ws, nb = _conv_ws(geom, x.device)
    if aff is not None:
        rc = L.load().jvae_conv2d_wgrad_aff_f32(L.ptr(x), L.ptr(gy), L.ptr(gw), L.ptr(gb), int(inplace),
                                                L.ptr(aff[0]), L.ptr(aff[1]), int(aff[2]), *geom, L.ptr(ws), nb,
                                                L.stream_ptr())
    else:
        rc = L.load().jvae_conv2d_wgrad_f32(L.ptr(x), L.ptr(gy), L.ptr(gw), L.ptr(gb), int(inplace), *geom, L.ptr(ws), nb,
                                            L.stream_ptr())
    L.check(rc, 'jvae_conv2d_wgrad_f32')
    return (None, None) if inplace else (gw, gb)


class _Conv(torch.autograd.Function):
    """nn.Conv2d / nn.ConvTranspose2d (conv.py:186-196)."""

    @staticmethod
    def forward(ctx, x, w, b, spec, dead_bias, stats_out, aff=None):
        x = _c(_f32(x, 'conv'))
        if not L.span_depth:               # a convolution called outside forward() / evaluate() (net.imager(z), a raw module call):
            L.pack_cache_end()             # nobody vouches for the weights now - an armed cache (train-mode evaluate() without its
        ctx.w_ref, ctx.b_ref = w, b        # backward yet) may predate a change through .data; pack per call
        w = _c(w)
        ctx.aff = aff
        if aff is not None:                # x is a pre-BatchNorm tensor: normalise (+ReLU) while staging it
            y, st, ns = conv_fwd_aff_raw(x, w, b, spec, aff, stats_out is not None)
            if stats_out is not None:
                stats_out['stats'], stats_out['nsplit'], stats_out['pivot'] = st, ns, b
        elif stats_out is not None:        # the caller is a train-mode BatchNorm: let the conv epilogue do its sums
            y, st, ns = conv_fwd_stats_raw(x, w, b, spec)
            stats_out['stats'], stats_out['nsplit'], stats_out['pivot'] = st, ns, b
        else:
            y = conv_fwd_raw(x, w, b, spec)
        ctx.save_for_backward(x, w)
        ctx.spec = spec
        ctx.has_bias = b is not None
        ctx.dead_bias = dead_bias
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = _c(gy)
        gx = gw = gb = None
        _close_span_after_backward()
        want_b = ctx.has_bias and ctx.needs_input_grad[2] and not ctx.dead_bias
        w_slot = b_slot = None
        on_side = False
        if ctx.needs_input_grad[1] or want_b:
            w_slot = _grad_slot(ctx.w_ref)
            b_slot = _grad_slot(ctx.b_ref) if want_b else None
            # the first layer of the model (its input needs no gradient) is the END of the backward chain: its weight
            # gradient stays on the main stream, where it runs beside the side stream's last kernels instead of behind them
            on_side = (OVERLAP_WGRAD and ctx.needs_input_grad[0] and w_slot is not None
                       and (b_slot is not None or not want_b))
        if on_side:
            # the weight gradient needs gy and x, not this layer's dgrad: the side stream forks off BEFORE the dgrad is
            # queued.  In a captured graph the dgrad is then the first successor of gy's producer and stays on the
            # launch queue (the runtime moves later successors to other queues: a 10 us hand-over per hop of the chain)
            main = torch.cuda.current_stream(x.device)
            side = L.side_stream(x.device)
            side.wait_stream(main)
        if ctx.needs_input_grad[0]:
            gx = conv_dgrad_raw(gy, w, ctx.spec, x.shape)
        if ctx.needs_input_grad[1] or want_b:
            if on_side:
                # in-place into the flat gradient buffer: nothing downstream of this node consumes the result before
                # the optimiser, so the kernel goes to the side stream and overlaps the rest of backward
                with torch.cuda.stream(side):
                    conv_wgrad_raw(x, gy, ctx.spec, w.shape, want_b, w_slot, b_slot, ctx.aff)
                x.record_stream(side)
                gy.record_stream(side)
                if ctx.aff is not None:
                    ctx.aff[0].record_stream(side)
                    ctx.aff[1].record_stream(side)
                _join_after_backward()
            else:
                gw, gb = conv_wgrad_raw(x, gy, ctx.spec, w.shape, want_b, w_slot, b_slot, ctx.aff)
        if ctx.dead_bias and ctx.has_bias and ctx.needs_input_grad[2] and _grad_slot(ctx.b_ref) is None:
            gb = torch.zeros_like(ctx.b_ref)          # exact value; makes the bias a regular optimiser citizen
        return gx, gw, gb, None, None, None, None


def conv2d(x, w, b, spec, dead_bias=False, stats_out=None, aff=None):
    """dead_bias: the bias feeds a train-mode BatchNorm, which removes the channel mean: its true gradient is
    exactly zero (what autograd would produce is rounding noise), so the channel reduction is skipped.
    aff: (scale, shift, relu) of a deferred BatchNorm on x (batchnorm_defer)."""
    return _Conv.apply(x, w, b, spec, dead_bias, stats_out, aff)


# ------------------------------------------------------------------------------------------- batch norm
class _BatchNormAct(torch.autograd.Function):
    """nn.BatchNorm2d (train or eval) + optional ReLU (conv.py:214-220)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, nbt, training, relu, momentum, eps, ext):
        x = _c(_f32(x, 'batchnorm'))
        N, C = x.shape[0], x.shape[1]
        P = x.numel() // max(N * C, 1)
        lib = L.load()
        y = torch.empty_like(x)
        mean = torch.empty(C, device=x.device, dtype=torch.float32)
        invstd = torch.empty(C, device=x.device, dtype=torch.float32)
        nb = lib.jvae_bn_workspace_bytes(C)
        ws = L.workspace(nb, x.device)
        if ext is not None and ext.get('stats') is not None and training:
            rc = lib.jvae_bn_fwd_ext_f32(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv), L.ptr(nbt), L.ptr(y),
                                         L.ptr(mean), L.ptr(invstd), N, C, P, momentum, eps, int(training), int(relu),
                                         L.ptr(ext['stats']), int(ext['nsplit']), L.ptr(ext.get('pivot')),
                                         L.ptr(ws), ws.numel(), L.stream_ptr())
        else:
            rc = lib.jvae_bn_fwd_f32(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv), L.ptr(nbt), L.ptr(y),
                                     L.ptr(mean), L.ptr(invstd), N, C, P, momentum, eps, int(training), int(relu),
                                     L.ptr(ws), ws.numel(), L.stream_ptr())
        L.check(rc, 'jvae_bn_fwd_f32')
        if training:
            ctx.save_for_backward(x, gamma, beta, mean, invstd)
            ctx.relu = relu
            ctx.dims = (N, C, P)
            ctx.g_ref, ctx.b_ref = gamma, beta
        else:
            ctx.dims = None
        return y

    @staticmethod
    def backward(ctx, gy):
        if ctx.dims is None:
            raise L.JvaeHipError('backward through eval-mode BatchNorm is not part of the training step')
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        N, C, P = ctx.dims
        gy = _c(gy)
        lib = L.load()
        gx = torch.empty_like(x)
        sg, sb = _grad_slot(ctx.g_ref), _grad_slot(ctx.b_ref)
        inplace = sg is not None and sb is not None
        gg = sg if inplace else torch.empty(C, device=x.device, dtype=torch.float32)
        gb = sb if inplace else torch.empty(C, device=x.device, dtype=torch.float32)
        nb = lib.jvae_bn_workspace_bytes(C)
        ws = L.workspace(nb, x.device)
        rc = lib.jvae_bn_bwd_f32(L.ptr(gy), L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(mean), L.ptr(invstd), L.ptr(gx),
                                 L.ptr(gg), L.ptr(gb), int(inplace), N, C, P, int(ctx.relu), L.ptr(ws), ws.numel(),
                                 L.stream_ptr())
        L.check(rc, 'jvae_bn_bwd_f32')
        if inplace:
            gg = gb = None
        return gx, gg, gb, None, None, None, None, None, None, None, None


def batchnorm_act(x, gamma, beta, running_mean, running_var, num_batches_tracked, training, relu,
                  momentum=0.1, eps=1e-5, ext=None):
    """ext: {'stats', 'nsplit', 'pivot'} filled by conv2d(..., stats_out=ext) for the tensor x."""
    return _BatchNormAct.apply(x, gamma, beta, running_mean, running_var, num_batches_tracked, training, relu,
                               momentum, eps, ext)


class _BatchNormDefer(torch.autograd.Function):
    """nn.BatchNorm2d (train or eval) + optional ReLU whose arithmetic is deferred into the next convolution: forward
    only produces the statistics and the per-channel (scale, shift); the returned activation IS the input tensor (the
    consumer applies fmaf(x, scale, shift) [+ReLU] while staging it).  Backward is the ordinary BatchNorm backward on the
    gradient the consumer's dgrad returns (which is w.r.t. the normalised activation)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, nbt, training, relu, momentum, eps, ext):
        x = _c(_f32(x, 'batchnorm'))
        N, C = x.shape[0], x.shape[1]
        P = x.numel() // max(N * C, 1)
        lib = L.load()
        mean = torch.empty(C, device=x.device, dtype=torch.float32)
        invstd = torch.empty(C, device=x.device, dtype=torch.float32)
        coef = torch.empty((2, C), device=x.device, dtype=torch.float32)
        ws = L.workspace(lib.jvae_bn_workspace_bytes(C), x.device)
        use_ext = ext is not None and ext.get('stats') is not None and training
        rc = lib.jvae_bn_finalize_f32(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv), L.ptr(nbt),
                                      L.ptr(mean), L.ptr(invstd), L.ptr(coef[0]), L.ptr(coef[1]), N, C, P, momentum, eps,
                                      int(training), L.ptr(ext['stats']) if use_ext else None,
                                      int(ext['nsplit']) if use_ext else 0, L.ptr(ext.get('pivot')) if use_ext else None,
                                      L.ptr(ws), ws.numel(), L.stream_ptr())
        L.check(rc, 'jvae_bn_finalize_f32')
        if training:
            ctx.save_for_backward(x, gamma, beta, mean, invstd)
            ctx.relu = relu
            ctx.dims = (N, C, P)
            ctx.g_ref, ctx.b_ref = gamma, beta
        else:
            ctx.dims = None
        ctx.mark_non_differentiable(coef)
        ctx.set_materialize_grads(False)        # no zero-fill launch for the (never used) gradient of `coef`
        return x.view_as(x), coef

    @staticmethod
    def backward(ctx, gy, _gcoef):
        g = _BatchNormAct.backward(ctx, gy)
        return g


def batchnorm_defer(x, gamma, beta, running_mean, running_var, num_batches_tracked, training, relu,
                    momentum=0.1, eps=1e-5, ext=None):
    """-> (x_alias, (scale, shift, relu)): hand both to conv2d(..., aff=...) of a layer with conv_affine_ok()."""
    xa, coef = _BatchNormDefer.apply(x, gamma, beta, running_mean, running_var, num_batches_tracked, training, relu,
                                     momentum, eps, ext)
    return xa, (coef[0], coef[1], relu)


class _SyncBatchNormAct(torch.autograd.Function):
    """Train-mode BatchNorm2d (+ReLU) whose statistics span all data-parallel ranks (what the single-process reference
    computes on the whole global batch): two (C,2) all-reduces per layer, one in forward, one in backward."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, nbt, relu, momentum, eps, world, group):
        import torch.distributed as dist
        x = _c(_f32(x, 'sync batchnorm'))
        N, C = x.shape[0], x.shape[1]
        P = x.numel() // max(N * C, 1)
        lib = L.load()
        ws = L.workspace(lib.jvae_bn_workspace_bytes(C), x.device)
        pivot = rm.detach().clone()                       # identical on every rank; rm itself is updated by the kernel
        sums = torch.empty((C, 2), device=x.device, dtype=torch.float32)
        L.check(lib.jvae_bn_sums_f32(L.ptr(x), L.ptr(pivot), L.ptr(sums), N, C, P, L.ptr(ws), ws.numel(), L.stream_ptr()),
                'jvae_bn_sums_f32')
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        y = torch.empty_like(x)
        mean = torch.empty(C, device=x.device, dtype=torch.float32)
        invstd = torch.empty(C, device=x.device, dtype=torch.float32)
        rc = lib.jvae_bn_fwd_sync_f32(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv), L.ptr(nbt), L.ptr(y),
                                      L.ptr(mean), L.ptr(invstd), N, C, P, momentum, eps, int(relu), L.ptr(sums),
                                      L.ptr(pivot), int(world), L.ptr(ws), ws.numel(), L.stream_ptr())
        L.check(rc, 'jvae_bn_fwd_sync_f32')
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        ctx.cfg = (N, C, P, relu, int(world), group)
        ctx.g_ref, ctx.b_ref = gamma, beta
        return y

    @staticmethod
    def backward(ctx, gy):
        import torch.distributed as dist
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        N, C, P, relu, world, group = ctx.cfg
        gy = _c(gy)
        lib = L.load()
        ws = L.workspace(lib.jvae_bn_workspace_bytes(C), x.device)
        local = torch.empty((C, 2), device=x.device, dtype=torch.float32)
        rc = lib.jvae_bn_bwd_sums_f32(L.ptr(gy), L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(mean), L.ptr(invstd),
                                      L.ptr(local), N, C, P, int(relu), L.ptr(ws), ws.numel(), L.stream_ptr())
        L.check(rc, 'jvae_bn_bwd_sums_f32')
        glob = local.clone()
        dist.all_reduce(glob, op=dist.ReduceOp.SUM, group=group)
        gx = torch.empty_like(x)
        sg, sb = _grad_slot(ctx.g_ref), _grad_slot(ctx.b_ref)
        inplace = sg is not None and sb is not None
        gg = sg if inplace else torch.empty(C, device=x.device, dtype=torch.float32)
        gb = sb if inplace else torch.empty(C, device=x.device, dtype=torch.float32)
        rc = lib.jvae_bn_bwd_sync_f32(L.ptr(gy), L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(mean), L.ptr(invstd),
                                      L.ptr(local), L.ptr(glob), world, L.ptr(gx), L.ptr(gg), L.ptr(gb), int(inplace),
                                      N, C, P, int(relu), L.stream_ptr())
        L.check(rc, 'jvae_bn_bwd_sync_f32')
        if inplace:
            gg = gb = None
        return gx, gg, gb, None, None, None, None, None, None, None, None


def sync_batchnorm_act(x, gamma, beta, running_mean, running_var, num_batches_tracked, relu, momentum, eps, world, group=None):
    return _SyncBatchNormAct.apply(x, gamma, beta, running_mean, running_var, num_batches_tracked, relu, momentum, eps,
                                   world, group)


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kind):
        x = _c(_f32(x, 'act'))
        y = torch.empty_like(x)
        L.check(L.load().jvae_act_fwd_f32(L.ptr(x), L.ptr(y), x.numel(), kind, L.stream_ptr()), 'act_fwd')
        ctx.save_for_backward(y)
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, gy):
        y, = ctx.saved_tensors
        gy = _c(gy)
        gx = torch.empty_like(gy)
        L.check(L.load().jvae_act_bwd_f32(L.ptr(gy), L.ptr(y), L.ptr(gx), gy.numel(), ctx.kind, L.stream_ptr()), 'act_bwd')
        return gx, None


def act(x, kind):
    if kind == IDENT:
        if not x.is_cuda:
            raise L.JvaeHipError('jvae_hip ops need tensors resident on the GPU (no CPU fallback)')
        return x
    return _Act.apply(x, kind)


# ------------------------------------------------------------------------------------------- dropout
class _Dropout(torch.autograd.Function):
    """nn.Dropout(p) in train mode (layers.py:287-288): counter-based mask, regenerated in backward from the seed.
    `seed`: a Python int, or a 1-element int64 DEVICE tensor (graph-capturable: no host value enters the launch)."""

    @staticmethod
    def forward(ctx, x, p, seed):
        x = _c(_f32(x, 'dropout'))
        y = torch.empty_like(x)
        if torch.is_tensor(seed):
            L.check(L.load().jvae_dropout_dev_f32(L.ptr(x), L.ptr(y), x.numel(), float(p), L.ptr(seed), 0, L.stream_ptr()),
                    'dropout')
        else:
            seed = int(seed)
            L.check(L.load().jvae_dropout_f32(L.ptr(x), L.ptr(y), x.numel(), float(p), seed, L.stream_ptr()), 'dropout')
        ctx.cfg = (float(p), seed)
        return y

    @staticmethod
    def backward(ctx, gy):
        p, seed = ctx.cfg
        gy = _c(gy)
        gx = torch.empty_like(gy)
        if torch.is_tensor(seed):
            L.check(L.load().jvae_dropout_dev_f32(L.ptr(gy), L.ptr(gx), gy.numel(), p, L.ptr(seed), 0, L.stream_ptr()),
                    'dropout_bwd')
        else:
            L.check(L.load().jvae_dropout_f32(L.ptr(gy), L.ptr(gx), gy.numel(), p, seed, L.stream_ptr()), 'dropout_bwd')
        return gx, None, None


def dropout(x, p, seed):
    return _Dropout.apply(x, p, seed)


# ------------------------------------------------------------------------------- pooling / up-sampling
POOL_MAX, POOL_AVG = 0, 1


class _Pool2d(torch.autograd.Function):
    """nn.MaxPool2d / nn.AvgPool2d of the layer DSL (tokens M / A; reference module/vae_layers/conv.py:201-206)."""

    @staticmethod
    def forward(ctx, x, K, S, P, mode):
        lib = L.load()
        x = _c(_f32(x, 'pool2d'))
        N, C, H, W = x.shape
        oh, ow = c_int(0), c_int(0)
        L.check(lib.jvae_pool2d_out_shape(H, W, K, S, P, byref(oh), byref(ow)), 'pool2d_out_shape')
        y = torch.empty((N, C, oh.value, ow.value), device=x.device, dtype=torch.float32)
        idx = torch.empty((N, C, oh.value, ow.value), device=x.device, dtype=torch.int32) if mode == POOL_MAX else None
        L.check(lib.jvae_pool2d_fwd_f32(L.ptr(x), L.ptr(y), L.ptr(idx), N * C, H, W, K, S, P, mode, L.stream_ptr()), 'pool2d_fwd')
        ctx.save_for_backward(idx)
        ctx.cfg = (N, C, H, W, K, S, P, mode)
        return y

    @staticmethod
    def backward(ctx, gy):
        idx, = ctx.saved_tensors
        N, C, H, W, K, S, P, mode = ctx.cfg
        gy = _c(gy)
        gx = torch.empty((N, C, H, W), device=gy.device, dtype=torch.float32)
        L.check(L.load().jvae_pool2d_bwd_f32(L.ptr(gy), L.ptr(idx), L.ptr(gx), N * C, H, W, K, S, P, mode, L.stream_ptr()),
                'pool2d_bwd')
        return gx, None, None, None, None


def pool2d(x, kernel_size, stride=None, padding=0, mode=POOL_MAX):
    return _Pool2d.apply(x, int(kernel_size), int(stride or kernel_size), int(padding), mode)


class _UpsampleNearest(torch.autograd.Function):
    """nn.UpsamplingNearest2d(scale_factor=int) (token U; reference module/vae_layers/conv.py:208-212)."""

    @staticmethod
    def forward(ctx, x, scale):
        x = _c(_f32(x, 'upsample_nearest'))
        N, C, H, W = x.shape
        y = torch.empty((N, C, H * scale, W * scale), device=x.device, dtype=torch.float32)
        L.check(L.load().jvae_upsample_nearest_fwd_f32(L.ptr(x), L.ptr(y), N * C, H, W, scale, L.stream_ptr()), 'upsample_fwd')
        ctx.cfg = (N, C, H, W, scale)
        return y

    @staticmethod
    def backward(ctx, gy):
        N, C, H, W, scale = ctx.cfg
        gy = _c(gy)
        gx = torch.empty((N, C, H, W), device=gy.device, dtype=torch.float32)
        L.check(L.load().jvae_upsample_nearest_bwd_f32(L.ptr(gy), L.ptr(gx), N * C, H, W, scale, L.stream_ptr()), 'upsample_bwd')
        return gx, None


def upsample_nearest(x, scale):
    if int(scale) != scale or scale < 1:
        raise L.JvaeHipError('nearest up-sampling is built for integer scale factors')
    return _UpsampleNearest.apply(x, int(scale))


# ------------------------------------------------------------------------------------------- latent
PRIOR_KIND = {'gaussian': 0, 'tilted': 1, 'uniform': 2}
VAR_KIND = {'scalar': 0, 'diag': 1, 'full': 2}


class _Latent(torch.autograd.Function):
    """clip + reparameterise + KL terms; see csrc/latent.hip for the reference anchors."""

    @staticmethod
    def forward(ctx, mu, lv_raw, eps, y, means, T, cfg):
        lib = L.load()
        mu = _c(_f32(mu, 'latent'))
        lv_raw = _c(lv_raw)
        eps = _c(eps)
        means = _c(means)
        T = _c(T)
        y = _c(y)
        if y.dtype != torch.int64:
            raise L.JvaeHipError('class labels must be int64')
        N, K = mu.shape
        Ls = eps.shape[0] - 1
        C = means.shape[0]
        dev = mu.device
        dict_ = torch.empty(K + 1, device=dev, dtype=torch.float32)
        L.check(lib.jvae_dict_stats_f32(L.ptr(means), L.ptr(dict_), C, K, L.stream_ptr()), 'dict_stats')
        lv = torch.empty_like(mu)
        z = torch.empty((Ls + 1, N, K), device=dev, dtype=torch.float32)
        kl, zd, vkl, dzd = (torch.empty(N, device=dev, dtype=torch.float32) for _ in range(4))
        forced = cfg.get('forced_lv')
        args = (N, K, Ls, C, cfg['prior'], cfg['var_dim'], cfg['tau'], cfg['alpha'], cfg['w'], int(cfg['sampled']),
                int(forced is not None))
        rc = lib.jvae_latent_fwd_f32(L.ptr(mu), L.ptr(lv_raw), L.ptr(eps), L.ptr(y), L.ptr(means), L.ptr(T), L.ptr(dict_),
                                     L.ptr(lv), L.ptr(z), L.ptr(kl), L.ptr(zd), L.ptr(vkl), L.ptr(dzd),
                                     *args, float(forced or 0.), L.stream_ptr())
        L.check(rc, 'jvae_latent_fwd_f32')
        ctx.save_for_backward(mu, lv_raw, lv, eps, y, means, T)
        ctx.args = args
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(dzd)
        return lv, z, kl, zd, vkl, dzd

    @staticmethod
    def backward(ctx, g_lv, g_z, g_kl, g_zd, g_vkl, _g_dzd):
        mu, lv_raw, lv, eps, y, means, T = ctx.saved_tensors
        lib = L.load()
        N, K, Ls, C, prior, var_dim = ctx.args[:6]
        dev = mu.device
        gmu = torch.empty_like(mu)
        glv = torch.empty_like(mu)
        need_means = ctx.needs_input_grad[4]
        need_T = ctx.needs_input_grad[5] and var_dim != 0
        gmeans = torch.zeros_like(means) if need_means else None
        gT = torch.zeros_like(T) if need_T else None
        nb = 16 * N + 8 * N * mu.shape[-1]
        ws = L.workspace(nb, dev)

        def opt(t):
            return None if t is None else _c(t)
        rc = lib.jvae_latent_bwd_f32(L.ptr(mu), L.ptr(lv_raw), L.ptr(lv), L.ptr(eps), L.ptr(y), L.ptr(means), L.ptr(T),
                                     L.ptr(opt(g_z)), L.ptr(opt(g_kl)), L.ptr(opt(g_zd)), L.ptr(opt(g_vkl)),
                                     None, L.ptr(opt(g_lv)),
                                     L.ptr(gmu), L.ptr(glv), L.ptr(gmeans), L.ptr(gT),
                                     *ctx.args, L.ptr(ws), ws.numel(), L.stream_ptr())
        L.check(rc, 'jvae_latent_bwd_f32')
        return gmu, glv, None, None, gmeans, gT, None


def latent(mu, lv_raw, eps, y, means, T, *, prior='gaussian', var_dim='scalar', tau=0., alpha=0., w=1.,
           sampled=True, forced_lv=None):
    """-> (log_var (N,K), z (L+1,N,K), kl, zdist, var_kl, dzdist (N,))"""
    cfg = dict(prior=PRIOR_KIND[prior], var_dim=VAR_KIND[var_dim], tau=float(tau), alpha=float(alpha), w=float(w),
               sampled=bool(sampled), forced_lv=forced_lv)
    return _Latent.apply(mu, lv_raw, eps, y, means, T, cfg)


# ------------------------------------------------------------------------------------------- losses
SIGMA_VALUE, SIGMA_LOG, SIGMA_CODED, SIGMA_RMSE = 0, 1, 2, 3      # kinds of sigma the loss kernels know (csrc/loss.hip)


def _check_sigma(sigma, mode, N):
    want = N if mode == SIGMA_CODED else 1
    if mode not in (SIGMA_VALUE, SIGMA_LOG, SIGMA_CODED, SIGMA_RMSE) or (mode != SIGMA_RMSE and sigma.numel() != want):
        raise L.JvaeHipError('sigma must be one value, one log value, N coded log values or the rmse kind; per-dimension '
                             'sigma is not built (the reference itself fails on it: cvae.py:649 / 789 shapes)')


class _Recon(torch.autograd.Function):
    """wmse (L,N) of x_reco[1:] against x (losses.py:8-27; cvae.py:649-652); sigma kinds: csrc/loss.hip."""

    @staticmethod
    def forward(ctx, x_reco, x, sigma, mode, snapshot):
        x_reco = _c(_f32(x_reco, 'recon'))
        x = _c(x)
        sigma = _c(sigma)
        mode = int(mode)
        Lp1, N = x_reco.shape[0], x_reco.shape[1]
        _check_sigma(sigma, mode, N)
        D = x.numel() // max(N, 1)
        wmse = torch.empty((Lp1 - 1, N), device=x.device, dtype=torch.float32)
        rc = L.load().jvae_recon_fwd_f32(L.ptr(x_reco), L.ptr(x), L.ptr(sigma), mode, L.ptr(wmse),
                                         Lp1 - 1, N, D, L.stream_ptr())
        L.check(rc, 'jvae_recon_fwd_f32')
        # snapshot: the caller's decay rule will change `sigma` in place before backward (see recon_bwd_kernel)
        ctx.sigma_fwd = sigma.detach().clone() if (snapshot and mode == SIGMA_VALUE) else None
        ctx.save_for_backward(x_reco, x, sigma, wmse)
        ctx.mode = mode
        ctx.dims = (Lp1 - 1, N, D)
        return wmse

    @staticmethod
    def backward(ctx, g):
        x_reco, x, sigma, wmse = ctx.saved_tensors
        Ls, N, D = ctx.dims
        g = _c(g)
        gxr = torch.empty_like(x_reco)
        gs = torch.empty_like(sigma) if (ctx.needs_input_grad[2] and ctx.mode != SIGMA_RMSE) else None
        ws = L.workspace(4 * Ls * N + 16, x.device)
        rc = L.load().jvae_recon_bwd_f32(L.ptr(x_reco), L.ptr(x), L.ptr(sigma), ctx.mode, L.ptr(ctx.sigma_fwd), L.ptr(g),
                                         L.ptr(wmse), L.ptr(gxr), L.ptr(gs), 0, Ls, N, D, L.ptr(ws), ws.numel(),
                                         L.stream_ptr())
        L.check(rc, 'jvae_recon_bwd_f32')
        return gxr, None, gs, None, None


def recon_wmse(x_reco, x, sigma, sigma_is_log, snapshot=False):
    """sigma_is_log: bool, or one of SIGMA_VALUE / SIGMA_LOG / SIGMA_CODED / SIGMA_RMSE."""
    return _Recon.apply(x_reco, x, sigma, int(sigma_is_log), snapshot)


class _MseRows(torch.autograd.Function):
    """The public `mse_loss(x_output, x_target, ndim, batch_mean=False)` of module/losses.py:8-27 on EVERY row of x_output:
    xo (L, N, D), x (N, D) -> (L, N) mean squares over D (jvae_mse_rows_{fwd,bwd}_f32: the reconstruction kernels told that
    there is no mean-path row in front of the rows - no copy, no padded row)."""

    @staticmethod
    def forward(ctx, xo, x):
        xo = _c(_f32(xo, 'mse_loss'))
        x = _c(_f32(x, 'mse_loss'))
        Ls, N, D = xo.shape
        if D % 4 == 0:                                   # the kernels use 16-byte loads: an odd view offset gets a copy
            xo = xo if xo.data_ptr() % 16 == 0 else xo.clone()
            x = x if x.data_ptr() % 16 == 0 else x.clone()
        wmse = torch.empty((Ls, N), device=x.device, dtype=torch.float32)
        rc = L.load().jvae_mse_rows_fwd_f32(L.ptr(xo), L.ptr(x), L.ptr(wmse), Ls, N, D, L.stream_ptr())
        L.check(rc, 'jvae_mse_rows_fwd_f32')
        ctx.save_for_backward(xo, x)
        return wmse

    @staticmethod
    def backward(ctx, g):
        xo, x = ctx.saved_tensors
        Ls, N, D = xo.shape
        g = _c(g)
        gx = torch.empty_like(xo)
        rc = L.load().jvae_mse_rows_bwd_f32(L.ptr(xo), L.ptr(x), L.ptr(g), L.ptr(gx), Ls, N, D, L.stream_ptr())
        L.check(rc, 'jvae_mse_rows_bwd_f32')
        return gx, None


def mse_rows(xo, x):
    """xo (L, N, D), x (N, D) -> (L, N) mean squares over D of every row (no copy of xo)."""
    return _MseRows.apply(xo, x)


class _Elbo(torch.autograd.Function):
    """wmse_s (L,N), kl (N,), ce (N,)|None, sigma -> (wmse, cross_x, total, mse) (cvae.py:662-670,773-791,887-902)."""

    @staticmethod
    def forward(ctx, wmse_s, kl, ce, sigma, mode, D, beta, cw):
        wmse_s, kl, sigma = _c(wmse_s), _c(kl), _c(sigma)
        ce = None if ce is None else _c(ce)
        mode = int(mode)
        Ls, N = wmse_s.shape
        _check_sigma(sigma, mode, N)
        wmse, cx, tot, mse = (torch.empty(N, device=kl.device, dtype=torch.float32) for _ in range(4))
        rc = L.load().jvae_elbo_fwd_f32(L.ptr(wmse_s), L.ptr(kl), L.ptr(ce), L.ptr(sigma), mode, L.ptr(wmse),
                                        L.ptr(cx), L.ptr(tot), L.ptr(mse), Ls, N, int(D), float(beta), float(cw),
                                        L.stream_ptr())
        L.check(rc, 'jvae_elbo_fwd_f32')
        ctx.save_for_backward(wmse_s if mode == SIGMA_RMSE else sigma)
        ctx.cfg = (mode, Ls, N, int(D), float(beta), float(cw), ce is not None)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(mse)
        return wmse, cx, tot, mse

    @staticmethod
    def backward(ctx, g_wmse, g_cx, g_tot, _g_mse):
        sigma, = ctx.saved_tensors                  # kind 3: the forward's wmse_s
        mode, Ls, N, D, beta, cw, has_ce = ctx.cfg
        dev = sigma.device
        g_ws = torch.empty((Ls, N), device=dev, dtype=torch.float32)
        g_kl = torch.empty(N, device=dev, dtype=torch.float32)
        g_ce = torch.empty(N, device=dev, dtype=torch.float32) if has_ce else None
        gs = torch.empty_like(sigma) if (ctx.needs_input_grad[3] and mode != SIGMA_RMSE) else None
        ws = L.workspace(4 * N + 16, dev)

        def opt(t):
            return None if t is None else _c(t)
        rc = L.load().jvae_elbo_bwd_f32(L.ptr(opt(g_wmse)), L.ptr(opt(g_cx)), L.ptr(opt(g_tot)), L.ptr(sigma), mode,
                                        L.ptr(g_ws), L.ptr(g_kl), L.ptr(g_ce), L.ptr(gs), 0, Ls, N, D, beta, cw,
                                        L.ptr(ws), ws.numel(), L.stream_ptr())
        L.check(rc, 'jvae_elbo_bwd_f32')
        return g_ws, g_kl, g_ce, gs, None, None, None, None


def elbo(wmse_s, kl, ce, sigma, sigma_is_log, D, beta, cw, with_mse=False):
    wmse, cx, tot, mse = _Elbo.apply(wmse_s, kl, ce, sigma, int(sigma_is_log), D, beta, cw)
    return (wmse, cx, tot, mse) if with_mse else (wmse, cx, tot)


def iws(wmse_s, eps, log_var, log_pz, sigma, sigma_is_log, D):
    """Importance-weighted bound of the evaluation path (no gradient): wmse_s (L,N), eps (L,N,K), log_var (N,K),
    log_pz (L,C,N) or (L,N) -> (C,N) or (N,)  (reference cvae.py:793-873)."""
    Ls, N, K = eps.shape
    conditional = log_pz.dim() == 3
    C = log_pz.shape[1] if conditional else 1
    rows = torch.empty((Ls, N), device=eps.device, dtype=torch.float32)
    out = torch.empty((C, N), device=eps.device, dtype=torch.float32)
    rc = L.load().jvae_iws_f32(L.ptr(_c(_f32(wmse_s, 'iws'))), L.ptr(_c(eps)), L.ptr(_c(log_var)), L.ptr(_c(log_pz)), L.ptr(_c(sigma)),
                              int(sigma_is_log), Ls, N, K, C, int(D), L.ptr(rows), L.ptr(out), L.stream_ptr())
    L.check(rc, 'jvae_iws_f32')
    return out if conditional else out[0]


def measures(x, wmse, zdist, var_kl, sigma, sigma_is_log, means, flag, scratch, prev=None, batch=0):
    """-> device tensor of 16 floats (layout in csrc/loss.hip measures_kernel); `scratch`: 1-float device tensor;
    `prev`: the tensor returned for the previous batch (running means are continued on the device)."""
    lib = L.load()
    x = _c(x)
    sqnorm_accum(x, scratch, True)
    out = torch.empty(16, device=x.device, dtype=torch.float32)
    C, K = (means.shape if means is not None else (0, 0))
    rc = lib.jvae_measures_f32(L.ptr(scratch), x.numel(), L.ptr(_c(wmse)), L.ptr(_c(zdist)), L.ptr(_c(var_kl)), wmse.numel(), zdist.numel(),
                               L.ptr(_c(sigma)), int(sigma_is_log), L.ptr(None if means is None else _c(means)), C, K,
                               L.ptr(flag), L.ptr(prev), int(batch), L.ptr(out), L.stream_ptr())
    L.check(rc, 'jvae_measures_f32')
    return out


class _Xent(torch.autograd.Function):
    """Row-wise cross entropy, target y[r % N] (F.cross_entropy(reduction='none'), losses.py:76-86)."""

    @staticmethod
    def forward(ctx, logits, y):
        lg = _c(_f32(logits, 'xent').reshape(-1, logits.shape[-1]))
        y = _c(y)
        R, C = lg.shape
        ce = torch.empty(R, device=lg.device, dtype=torch.float32)
        L.check(L.load().jvae_xent_fwd_f32(L.ptr(lg), L.ptr(y), L.ptr(ce), R, y.numel(), C, L.stream_ptr()), 'xent_fwd')
        ctx.save_for_backward(lg, y)
        ctx.shape = logits.shape
        return ce.reshape(logits.shape[:-1])

    @staticmethod
    def backward(ctx, g):
        lg, y = ctx.saved_tensors
        R, C = lg.shape
        g = _c(g.reshape(-1))
        gl = torch.empty_like(lg)
        L.check(L.load().jvae_xent_bwd_f32(L.ptr(lg), L.ptr(y), L.ptr(g), L.ptr(gl), R, y.numel(), C, L.stream_ptr()),
                'xent_bwd')
        return gl.reshape(ctx.shape), None


def cross_entropy_rows(logits, y):
    return _Xent.apply(logits, y)


# ------------------------------------------------------------------------------------------- optimiser
def sqnorm_accum(g, acc, reset):
    lib = L.load()
    ws = L.workspace(lib.jvae_sqnorm_workspace_bytes(), g.device)
    L.check(lib.jvae_sqnorm_accum_f32(L.ptr(g), g.numel(), L.ptr(acc), int(reset), L.ptr(ws), ws.numel(), L.stream_ptr()), 'sqnorm')


def clip_scale(g, sqnorm, max_norm):
    L.check(L.load().jvae_clip_scale_f32(L.ptr(g), g.numel(), L.ptr(sqnorm), float(max_norm), L.stream_ptr()), 'clip_scale')


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, max_norm=0., sqnorm=None, flag=None):
    rc = L.load().jvae_adam_step_f32(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), lr, beta1, beta2, eps,
                                     weight_decay, step, float(max_norm or 0.), L.ptr(sqnorm), L.ptr(flag), L.stream_ptr())
    L.check(rc, 'jvae_adam_step_f32')


def sgd_step(p, g, buf, lr, momentum, nesterov, weight_decay, first, max_norm=0., sqnorm=None, flag=None):
    """torch.optim.SGD update of one flat buffer (clip coefficient and NaN/Inf flag fused as for Adam)."""
    L.check(L.load().jvae_sgd_step_f32(L.ptr(p), L.ptr(g), L.ptr(buf), p.numel(), float(lr), float(momentum), int(bool(nesterov)),
                                      float(weight_decay), int(bool(first)), float(max_norm or 0.), L.ptr(sqnorm), L.ptr(flag),
                                      L.stream_ptr()), 'jvae_sgd_step_f32')


def adam_step_dev(p, g, m, v, hyper, advance, eps, weight_decay, max_norm=0., sqnorm=None, flag=None):
    """Adam with step count / lr / betas read from the 6-float device block `hyper` (capturable: no host state)."""
    rc = L.load().jvae_adam_step_dev_f32(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), L.ptr(hyper), int(advance),
                                         eps, weight_decay, float(max_norm or 0.), L.ptr(sqnorm), L.ptr(flag),
                                         L.stream_ptr())
    L.check(rc, 'jvae_adam_step_dev_f32')


# ------------------------------------------------------------------------------------------- input pipeline
def augment_batch(images_u8, flip=None, dy=None, dx=None, pad=0, nhwc=True):
    """uint8 batch (N,H,W,C) [or (N,C,H,W) with nhwc=False] -> float32 (N,C,H,W) in [0,1] with the reference's training
    augmentation (utils/torch_load.py:405-426): horizontal flip where flip[n], then edge padding by `pad` and a crop at
    (dy[n], dx[n]) in [0, 2*pad].  flip: uint8/bool (N,), dy/dx: int32 (N,), all on the device (None = identity)."""
    if images_u8.dtype != torch.uint8:
        raise L.JvaeHipError('augment_batch expects a uint8 batch')
    images_u8 = _c(images_u8)
    if nhwc:
        N, H, W, C = images_u8.shape
    else:
        N, C, H, W = images_u8.shape
    out = torch.empty((N, C, H, W), device=images_u8.device, dtype=torch.float32)
    f = None if flip is None else _c(flip.to(torch.uint8))
    a = None if dy is None else _c(dy.to(torch.int32))
    b = None if dx is None else _c(dx.to(torch.int32))
    rc = L.load().jvae_augment_u8_f32(L.ptr(images_u8), L.ptr(f), L.ptr(a), L.ptr(b), L.ptr(out), N, C, H, W, int(pad),
                                      int(nhwc), L.stream_ptr())
    L.check(rc, 'jvae_augment_u8_f32')
    return out


def draw_augmentation(N, pad, device, generator=None, flip=True, crop=True):
    """Random decisions of RandomHorizontalFlip(p=0.5) / RandomCrop(padding=pad) for N images, on the device."""
    f = (torch.rand(N, device=device, generator=generator) < 0.5) if flip else None
    dy = torch.randint(0, 2 * pad + 1, (N,), device=device, generator=generator, dtype=torch.int32) if crop else None
    dx = torch.randint(0, 2 * pad + 1, (N,), device=device, generator=generator, dtype=torch.int32) if crop else None
    return f, dy, dx
