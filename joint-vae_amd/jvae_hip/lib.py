"""ctypes binding of libjvae_hip.so (C ABI: include/jvae_hip.h).

There is NO fallback: if the shared library is missing or a tensor is not resident on an AMD GPU every
op raises.  The library is built in-tree by `__graft_entry__.build()` / `make -C joint-vae_amd/csrc`.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_long, c_longlong, c_size_t, c_void_p, POINTER

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('JVAE_HIP_LIB') or os.path.join(_HERE, 'libjvae_hip.so')      # JVAE_HIP_LIB: A/B builds

_lib = None

P = c_void_p
_SIGS = {
    'jvae_version': (c_char_p, []),
    'jvae_gemm_f32': (c_int, [c_int] * 4 + [P, c_long, c_long, c_long] * 3 + [P, c_int, c_int, c_int, P]),
    'jvae_splitk_fold_f32': (c_int, [P, P, P, c_int, c_long, c_int, c_int, c_int, P]),
    'jvae_conv2d_workspace_bytes': (c_size_t, [c_int] * 11),
    'jvae_conv2d_set_split_bf16': (c_int, [c_int]),
    'jvae_conv2d_set_split_shape16': (c_int, [c_int]),
    'jvae_conv2d_out_shape': (c_int, [c_int] * 8 + [POINTER(c_int), POINTER(c_int)]),
    'jvae_conv2d_fwd_f32': (c_int, [P, P, P, P] + [c_int] * 11 + [P, c_size_t, P]),
    'jvae_conv2d_stats_splits': (c_int, [c_int] * 11),
    'jvae_conv2d_fwd_stats_f32': (c_int, [P, P, P, P, P, POINTER(c_int)] + [c_int] * 11 + [P, c_size_t, P]),
    'jvae_conv2d_dgrad_f32': (c_int, [P, P, P] + [c_int] * 11 + [P, c_size_t, P]),
    'jvae_conv2d_wgrad_f32': (c_int, [P, P, P, P, c_int] + [c_int] * 11 + [P, c_size_t, P]),
    'jvae_b8_pack_f32': (c_int, [P, P, c_int, c_int, c_long, P]),
    'jvae_b8_unpack_f32': (c_int, [P, P, c_int, c_int, c_long, c_int, P]),
    'jvae_conv2d_native_b8': (c_int, [c_int] * 11),
    'jvae_conv2d_workspace_bytes_b8': (c_size_t, [c_int] * 11),
    'jvae_conv2d_stats_splits_b8': (c_int, [c_int] * 11),
    'jvae_conv2d_fwd_b8': (c_int, [P, P, P, P, c_int, P, POINTER(c_int)] + [c_int] * 11 + [P, c_size_t, P]),
    'jvae_conv2d_dgrad_b8': (c_int, [P, P, P] + [c_int] * 11 + [P, c_size_t, P]),
    'jvae_conv2d_wgrad_b8': (c_int, [P, P, P, P, c_int] + [c_int] * 11 + [P, c_size_t, P]),
    'jvae_bn_workspace_bytes_b8': (c_size_t, [c_int]),
    'jvae_bn_plan_b8': (c_int, [c_int, c_int, c_long, POINTER(c_int), POINTER(c_int)]),
    'jvae_bn_fwd_b8': (c_int, [P] * 9 + [c_int, c_int, c_long, c_float, c_float, c_int, c_int, P, c_int, P, P, c_size_t, P]),
    'jvae_bn_bwd_b8': (c_int, [P] * 9 + [c_int, c_int, c_int, c_long, c_int, P, c_size_t, P]),
    'jvae_bn_finalize_b8': (c_int, [P] * 9 + [c_int, c_int, c_long, c_float, c_float, c_int, P, c_int, P, P, c_size_t, P]),
    'jvae_bn_sums_b8': (c_int, [P, P, P, c_int, c_int, c_long, P, c_size_t, P]),
    'jvae_bn_fwd_sync_b8': (c_int, [P] * 9 + [c_int, c_int, c_long, c_float, c_float, c_int, P, P, c_int, P, c_size_t, P]),
    'jvae_bn_bwd_sums_b8': (c_int, [P] * 7 + [c_int, c_int, c_long, c_int, P, c_size_t, P]),
    'jvae_bn_bwd_sync_b8': (c_int, [P] * 8 + [c_int, P, P, P, c_int, c_int, c_int, c_long, c_int, P, c_size_t, P]),
    'jvae_conv2d_affine_ok_b8': (c_int, [c_int] * 11),
    'jvae_conv2d_fwd_aff_b8': (c_int, [P, P, P, P, c_int, P, POINTER(c_int), P, P, c_int] + [c_int] * 11 + [P, c_size_t, P]),
    'jvae_conv2d_wgrad_aff_b8': (c_int, [P, P, P, P, c_int, P, P, c_int] + [c_int] * 11 + [P, c_size_t, P]),
    'jvae_relu_fwd_b8': (c_int, [P, P, c_long, P]),
    'jvae_relu_bwd_b8': (c_int, [P, P, P, c_long, P]),
    'jvae_bn_finalize_f32': (c_int, [P] * 10 + [c_int, c_int, c_int, c_float, c_float, c_int, P, c_int, P, P, c_size_t, P]),
    'jvae_conv2d_affine_ok': (c_int, [c_int] * 11),
    'jvae_conv2d_fwd_aff_f32': (c_int, [P, P, P, P, P, POINTER(c_int), P, P, c_int] + [c_int] * 11 + [P, c_size_t, P]),
    'jvae_conv2d_wgrad_aff_f32': (c_int, [P, P, P, P, c_int, P, P, c_int] + [c_int] * 11 + [P, c_size_t, P]),
    'jvae_pack_cache_configure': (c_int, [P, c_size_t]),
    'jvae_pack_cache_begin': (c_int, [P, c_longlong, P, c_int]),
    'jvae_pack_cache_end': (c_int, []),
    'jvae_pack_cache_pin': (c_int, []),
    'jvae_pack_cache_reset': (c_int, []),
    'jvae_pack_cache_stats': (c_int, [POINTER(c_int), POINTER(c_longlong), POINTER(c_longlong), POINTER(c_longlong)]),
    'jvae_channel_sum_workspace_bytes': (c_size_t, [c_int]),
    'jvae_channel_sum_f32': (c_int, [P, P, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    'jvae_bn_workspace_bytes': (c_size_t, [c_int]),
    'jvae_bn_plan': (c_int, [c_int, c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    'jvae_image_range': (c_int, [c_int, c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    'jvae_bn_fwd_f32': (c_int, [P] * 9 + [c_int, c_int, c_int, c_float, c_float, c_int, c_int, P, c_size_t, P]),
    'jvae_bn_fwd_ext_f32': (c_int, [P] * 9 + [c_int, c_int, c_int, c_float, c_float, c_int, c_int, P, c_int, P, P, c_size_t, P]),
    'jvae_bn_sums_f32': (c_int, [P, P, P, c_int, c_int, c_int, P, c_size_t, P]),
    'jvae_bn_fwd_sync_f32': (c_int, [P] * 9 + [c_int, c_int, c_int, c_float, c_float, c_int, P, P, c_int, P, c_size_t, P]),
    'jvae_bn_bwd_sums_f32': (c_int, [P] * 7 + [c_int, c_int, c_int, c_int, P, c_size_t, P]),
    'jvae_bn_bwd_sync_f32': (c_int, [P] * 8 + [c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'jvae_bn_bwd_f32': (c_int, [P] * 9 + [c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    'jvae_act_fwd_f32': (c_int, [P, P, c_long, c_int, P]),
    'jvae_act_bwd_f32': (c_int, [P, P, P, c_long, c_int, P]),
    'jvae_dict_stats_f32': (c_int, [P, P, c_int, c_int, P]),
    'jvae_latent_fwd_f32': (c_int, [P] * 13 + [c_int] * 6 + [c_float] * 3 + [c_int, c_int, c_float, P]),
    'jvae_latent_bwd_f32': (c_int, [P] * 17 + [c_int] * 6 + [c_float] * 3 + [c_int, c_int, P, c_size_t, P]),
    'jvae_recon_fwd_f32': (c_int, [P, P, P, c_int, P, c_int, c_int, c_int, P]),
    'jvae_recon_bwd_f32': (c_int, [P, P, P, c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    'jvae_mse_rows_fwd_f32': (c_int, [P, P, P, c_int, c_int, c_int, P]),
    'jvae_mse_rows_bwd_f32': (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    'jvae_elbo_fwd_f32': (c_int, [P, P, P, P, c_int, P, P, P, P, c_int, c_int, c_int, c_float, c_float, P]),
    'jvae_elbo_bwd_f32': (c_int, [P, P, P, P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_float, P, c_size_t, P]),
    'jvae_measures_f32': (c_int, [P, c_long, P, P, P, c_int, c_int, P, c_int, P, c_int, c_int, P, P, c_int, P, P]),
    'jvae_dropout_f32': (c_int, [P, P, c_long, c_float, c_long, P]),
    'jvae_dropout_dev_f32': (c_int, [P, P, c_long, c_float, P, c_long, P]),
    'jvae_iws_f32': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    'jvae_xent_fwd_f32': (c_int, [P, P, P, c_int, c_int, c_int, P]),
    'jvae_xent_bwd_f32': (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    'jvae_augment_u8_f32': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'jvae_sqnorm_workspace_bytes': (c_size_t, []),
    'jvae_sqnorm_accum_f32': (c_int, [P, c_long, P, c_int, P, c_size_t, P]),
    'jvae_clip_scale_f32': (c_int, [P, c_long, P, c_float, P]),
    'jvae_adam_step_dev_f32': (c_int, [P, P, P, P, c_long, P, c_int, c_float, c_float, c_float, P, P, P]),
    'jvae_adam_step_f32': (c_int, [P, P, P, P, c_long] + [c_float] * 5 + [c_long, c_float, P, P, P]),
    'jvae_sgd_step_f32': (c_int, [P, P, P, c_long, c_float, c_float, c_int, c_float, c_int, c_float, P, P, P]),
    'jvae_pool2d_out_shape': (c_int, [c_int] * 5 + [POINTER(c_int), POINTER(c_int)]),
    'jvae_pool2d_fwd_f32': (c_int, [P, P, P, c_long] + [c_int] * 6 + [P]),
    'jvae_pool2d_bwd_f32': (c_int, [P, P, P, c_long] + [c_int] * 6 + [P]),
    'jvae_upsample_nearest_fwd_f32': (c_int, [P, P, c_long, c_int, c_int, c_int, P]),
    'jvae_upsample_nearest_bwd_f32': (c_int, [P, P, c_long, c_int, c_int, c_int, P]),
}


class JvaeHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises JvaeHipError when the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise JvaeHipError(f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                           'or `make -C joint-vae_amd/csrc` (there is no CPU/PyTorch fallback)')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)            # AttributeError here = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def exported_symbols():
    return list(_SIGS)


_ERR = {-1: 'invalid argument', -2: 'unsupported configuration', -3: 'workspace too small'}


def check(rc, what):
    if rc != 0:
        msg = _ERR.get(rc, f'hipError_t {rc}')
        raise JvaeHipError(f'{what} failed: {msg}')


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_ptr(device_index=None):
    """hipStream_t of the current stream (the raw-handle query: this is called by every op)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device() if device_index is None else device_index)
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a dense fp32/int64 CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise JvaeHipError('jvae_hip ops need tensors resident on the GPU (no CPU fallback); got device '
                           + str(t.device))
    if not t.is_contiguous():
        raise JvaeHipError('jvae_hip ops need contiguous tensors')
    return t.data_ptr()


_workspaces = {}
_side_streams = {}


def side_stream(device):
    """Second HIP stream of a device: weight-gradient kernels (MFMA-bound, off the critical path of backward) run
    here so that they overlap the HBM-bound BatchNorm-backward passes of the main stream."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _side_streams.get(idx)
    if st is None:
        _side_streams[idx] = st = torch.cuda.Stream(device=idx)
    return st


def join_side_stream(device=None):
    """Make the current stream wait for everything queued on the side stream(s) (called before gradients are read:
    all-reduce, clip, Adam)."""
    for idx, st in _side_streams.items():
        if device is None or device.index in (None, idx):
            torch.cuda.current_stream(idx).wait_stream(st)


def workspace(nbytes, device):
    """Stream-ordered scratch shared by all ops of one (device, stream) (grown on demand, never shrunk)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, stream_ptr(idx))
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        _workspaces[key] = ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
    return ws


# ---- per-step cache of the re-packed convolution weights (csrc/pack_cache.hip) ---------------------------------------------
PACK_CACHE_BYTES = int(os.environ.get('JVAE_PACK_CACHE_MB', '64')) << 20       # JVAE_PACK_CACHE_MB=0: off (A/B switch)
_pack_cache = {}
span_depth = 0        # > 0 while a model-level forward()/evaluate() is running (cvae._constant_weights): see ops._Conv.forward


def pack_cache_begin(weights, device):
    """Start of a span with constant weights (evaluate() ... end of backward): ONE launch on the current stream re-packs every
    registered convolution weight, the convolutions of the span then skip their own pack launches.  `weights`: the tensors the
    span vouches for - their addresses are both the owner key (another model, .to() or a re-flattened optimiser buffer change
    it) and the only addresses the cache will create entries for (see jvae_pack_cache_begin)."""
    if not PACK_CACHE_BYTES:
        return
    lib = load()
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _pack_cache:
        if _pack_cache:                           # one cache per process (one process per GPU): another device switches it off
            check(lib.jvae_pack_cache_configure(None, 0), 'jvae_pack_cache_configure')
            _pack_cache[idx] = None
        else:
            buf = torch.empty(PACK_CACHE_BYTES + 256, dtype=torch.uint8, device=device)
            off = (-buf.data_ptr()) % 256
            _pack_cache[idx] = buf
            check(lib.jvae_pack_cache_configure(buf.data_ptr() + off, PACK_CACHE_BYTES), 'jvae_pack_cache_configure')
    if _pack_cache[idx] is None:
        return
    addrs = tuple(w.data_ptr() for w in weights)
    arr = (ctypes.c_void_p * max(len(addrs), 1))(*addrs)
    check(lib.jvae_pack_cache_begin(stream_ptr(idx), hash(addrs), arr, len(addrs)), 'jvae_pack_cache_begin')


def pack_cache_end():
    """Backward is over, the weights are about to change (optimiser, load_state_dict, .to()) or are no longer vouched for:
    convolutions pack per call again."""
    if _lib is not None and _pack_cache:
        _lib.jvae_pack_cache_end()


def pack_cache_pin():
    """Inside a bracket: keep this owner's cache region for good (a HIP graph is being captured with its slot addresses)."""
    if _lib is not None and _pack_cache and PACK_CACHE_BYTES:
        _lib.jvae_pack_cache_pin()


def pack_cache_stats():
    lib = load()
    e, h, m, r = c_int(), c_longlong(), c_longlong(), c_longlong()
    lib.jvae_pack_cache_stats(ctypes.byref(e), ctypes.byref(h), ctypes.byref(m), ctypes.byref(r))
    return {'entries': e.value, 'hits': h.value, 'misses': m.value, 'refreshes': r.value}
