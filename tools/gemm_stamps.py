"""GPU box: phase times of gemm_x3_kernel's K step (diagnostic build libjvae_stamps.so, -DJVAE_GEMM_STAMPS) on the shapes of the
7x7 head of conv32: forward (M=512, N=200, K=3136 in 4 K slices, 4 positions) as the step launches it."""
import ctypes, os, sys, torch
REPO = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
os.environ['JVAE_HIP_LIB'] = os.path.join(REPO, 'joint-vae_amd', 'jvae_hip', 'libjvae_stamps.so')
from jvae_hip import lib as L, ops
lib = L.load()
dbg = ctypes.CDLL(os.environ['JVAE_HIP_LIB']).jvae_debug_gemm_stamps
dbg.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
def stamps(reset=True):
    a = (ctypes.c_ulonglong * 8)()
    assert dbg(a, int(reset)) == 0
    return list(a)
names = ['barrier 1', 'registers->LDS (+ wait for the stage loads)', 'barrier 2', 'issue next loads', 'fragment reads + MFMA issue', 'K steps', 'prologue', 'epilogue (stores + drain)']
spec = ops.ConvSpec(64, 200, 7, 1, 0, 0, False)
x = torch.randn(512, 64, 8, 8, device='cuda'); w = torch.randn(200, 64, 7, 7, device='cuda') * 0.02; b = torch.zeros(200, device='cuda')
for _ in range(3):
    y = ops.conv_fwd_raw(x, w, b, spec)
stamps()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    y = ops.conv_fwd_raw(x, w, b, spec)
e1.record(); torch.cuda.synchronize()
s = stamps()
steps = s[5]
print('7x7 head forward (unfold + K-sliced product + fold): %.1f us per call' % (e0.elapsed_time(e1) * 100))
print('wave-0 cycles per K step (mean over %d workgroup-steps):' % steps)
for n, v in zip(names, s):
    if n != 'K steps':
        print('  %-48s %8.1f' % (n, v / steps if n not in ('prologue', 'epilogue (stores + drain)') else v / (steps / 25.0)))
