"""GPU box: how fast is an HBM-bound BatchNorm-backward pass while an MFMA-bound weight-gradient kernel runs on another
stream?  (largest layer of config 2: 1024 x 32 x 32 x 32 fp32)"""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import ops, lib as L
N, C, H = 1024, 32, 32
dev = 'cuda'
spec = ops.ConvSpec(C, C, 5, 1, 2, 0, transposed=True)
x = torch.randn(N, C, H, H, device=dev); gy = torch.randn(N, C, H, H, device=dev)
w = torch.randn(C, C, 5, 5, device=dev) * 0.03
z = torch.randn(N, C, H, H, device=dev); dy = torch.randn(N, C, H, H, device=dev)
gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
mean = z.mean((0, 2, 3)); invstd = (z.var((0, 2, 3), unbiased=False) + 1e-5).rsqrt()
dz = torch.empty_like(z); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
lib = L.load()
ws_bn = torch.empty(lib.jvae_bn_workspace_bytes(C), dtype=torch.uint8, device=dev)
def bn_bwd():
    L.check(lib.jvae_bn_bwd_f32(L.ptr(dy), L.ptr(z), L.ptr(gamma), L.ptr(beta), L.ptr(mean), L.ptr(invstd), L.ptr(dz), L.ptr(dg),
                                L.ptr(db), 0, N, C, H * H, 1, L.ptr(ws_bn), ws_bn.numel(), L.stream_ptr()), 'bn')
gw = torch.zeros(C, C, 5, 5, device=dev)
def wgrad():
    ops.conv_wgrad_raw(x, gy, spec, w.shape, False, gw, None)
def dgrad():
    ops.conv_dgrad_raw(gy, w, spec, x.shape)
def t(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print('alone: bn_bwd %.0f us, wgrad %.0f us, dgrad %.0f us' % (t(bn_bwd), t(wgrad), t(dgrad)))
side = torch.cuda.Stream()
def both(main_f, side_f, reps=10):
    torch.cuda.synchronize()
    e0, e1, s1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    main = torch.cuda.current_stream()
    e0.record()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for _ in range(reps): side_f()
        s1.record()
    for _ in range(reps): main_f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, e0.elapsed_time(s1) / reps * 1e3
for nm, f in (('wgrad', wgrad), ('dgrad', dgrad)):
    both(bn_bwd, f)          # first use of the side stream: workspace allocation, kernel attributes (NOT timed: an earlier
    m, s = both(bn_bwd, f)   # version of this probe timed it and reported a 2.3x collapse that does not exist)
    print('bn_bwd (main) beside %s (side): bn %.0f us per call, %s %.0f us per call' % (nm, m, nm, s))
m, s = both(dgrad, wgrad)
print('dgrad (main) beside wgrad (side): dgrad %.0f, wgrad %.0f us per call' % (m, s))
# which property of the BatchNorm-backward kernels makes them crawl beside the weight gradients?  A plain streaming copy and
# a streaming reduction of the same tensors, for comparison
def copy():
    dz.copy_(z)
def tsum():
    return z.sum()
print('alone: copy %.0f us, sum %.0f us' % (t(copy), t(tsum)))
for nm, f in (('copy', copy), ('sum', tsum)):
    m, s = both(f, wgrad)
    print('%s (main) beside wgrad (side): %s %.0f us per call, wgrad %.0f us per call' % (nm, nm, m, s))
m, s = both(wgrad, bn_bwd)
print('wgrad (main) beside bn_bwd (side): wgrad %.0f, bn %.0f us per call' % (m, s))
# stream roles: which stream carries the chain (BatchNorm backward) and which the weight gradients?
def pair(sa, sb, fa, fb, reps=10):
    torch.cuda.synchronize()
    e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    cur = torch.cuda.current_stream()
    e0.record()
    for s_ in (sa, sb):
        if s_ is not None: s_.wait_stream(cur)
    with torch.cuda.stream(sb if sb is not None else cur):
        for _ in range(reps): fb()
        eb.record()
    with torch.cuda.stream(sa if sa is not None else cur):
        for _ in range(reps): fa()
        ea.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(ea) / reps * 1e3, e0.elapsed_time(eb) / reps * 1e3
lo, hi = -1, 0
try:
    from ctypes import CDLL, c_int, byref
    a_, b_ = c_int(), c_int()
    CDLL('libamdhip64.so').hipDeviceGetStreamPriorityRange(byref(a_), byref(b_))
    print('stream priority range: least %d greatest %d' % (a_.value, b_.value))
except Exception as e:
    print('priority range query failed', e)
A, B = torch.cuda.Stream(), torch.cuda.Stream()
HP = torch.cuda.Stream(priority=-1)
for nm, sa, sb in (('bn default / wgrad created', None, B), ('bn created / wgrad created', A, B), ('bn created / wgrad default', A, None),
                   ('bn high-priority created / wgrad created', HP, B), ('bn high-priority created / wgrad default', HP, None)):
    for _ in range(2):
        m, s = pair(sa, sb, bn_bwd, wgrad)
    print('%-46s bn %.0f us, wgrad %.0f us per call' % (nm + ':', m, s))
