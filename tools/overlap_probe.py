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
    m, s = both(bn_bwd, f)
    print('bn_bwd (main) beside %s (side): bn %.0f us per call, %s %.0f us per call' % (nm, m, nm, s))
m, s = both(dgrad, wgrad)
print('dgrad (main) beside wgrad (side): dgrad %.0f, wgrad %.0f us per call' % (m, s))
