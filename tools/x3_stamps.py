"""GPU box, DIAGNOSTIC build only (make -C joint-vae_amd/csrc stamps; JVAE_HIP_LIB=.../libjvae_stamps.so): anatomy of
conv5_x3_kernel on imager.15 forward (1024 x 32 x 32 x 32, deferred BatchNorm in, bias, BatchNorm sums out) from s_memtime stamps
of wave 0 of every workgroup.  Prints mean cycles per phase, per-group detail, the in-kernel clock (s_memtime / s_memrealtime)."""
import ctypes, os, sys, time
import numpy as np
import torch
REPO = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import ops, lib as L
lib = L.load()
raw = ctypes.CDLL(L.LIB_PATH)
raw.jvae_x3_set_stamp_buffer.argtypes = [ctypes.c_void_p]
N, C, H = 1024, 32, 32
spec = ops.ConvSpec(C, C, 5, 1, 2, 0, True)
torch.manual_seed(0)
x = torch.randn(N, C, H, H, device='cuda'); w = torch.randn(C, C, 5, 5, device='cuda') * 0.05; b = torch.randn(C, device='cuda')
aff = (torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.1, True)
for sh in (1,):          # the stamped epilogue is the 16x16x32 form's
    lib.jvae_conv2d_set_split_shape16(sh)
    f = (lambda: ops.conv_fwd_aff_raw(x, w, b, spec, aff, True)) if os.environ.get('AFF', '1') != '0' else (lambda: ops.conv_fwd_stats_raw(x, w, b, spec))
    print('deferred BatchNorm on the input:', os.environ.get('AFF', '1') != '0')
    raw.jvae_x3_set_stamp_buffer(None)
    t0 = time.time()
    while time.time() - t0 < 2.0:                    # clock settles under load
        for _ in range(20): f()
        torch.cuda.synchronize()
    nwg = N * H * H // 256
    dbg = torch.zeros(nwg * 96, dtype=torch.int64, device='cuda')
    raw.jvae_x3_set_stamp_buffer(ctypes.c_void_p(dbg.data_ptr()))
    for _ in range(5): f()
    torch.cuda.synchronize()
    raw.jvae_x3_set_stamp_buffer(None)
    d = dbg.cpu().numpy().reshape(nwg, 96).astype(np.int64)
    NG = 14 if sh else 10
    life = d[:, 7] - d[:, 0]
    rt = (d[:, 65] - d[:, 64]).astype(np.float64)                       # 100 MHz ticks
    clk = np.median(life[rt > 0] / rt[rt > 0]) * 100e6
    print(f'== shape {"16x16x32" if sh else "32x32x16"}: {nwg} workgroups, life of a workgroup mean {life.mean():.0f} cycles (median {np.median(life):.0f}), '
          f'in-kernel clock {clk / 1e9:.3f} GHz, kernel span {(d[:, 7].max() - d[:, 0].min()) / clk * 1e6:.1f} us')
    def m(a, bb): return float((d[:, bb] - d[:, a]).mean())
    print(f'  entry -> zero fill issued            {m(0, 1):8.0f}')
    print(f'  first loads + __syncthreads          {m(1, 2):8.0f}')
    print(f'  first split/stores + barrier         {m(2, 3):8.0f}   (prologue total {m(0, 3):.0f})')
    st = np.array([[float((d[:, 8 + 4 * g + k + 1] - d[:, 8 + 4 * g + k]).mean()) for k in range(3)] for g in range(NG)])
    tail = np.array([float((d[:, (8 + 4 * (g + 1)) if g + 1 < NG else 5] - d[:, 11 + 4 * g]).mean()) for g in range(NG)])
    print(f'  per group (mean over {NG}): store W + issue loads {st[:, 0].mean():6.0f} | fragment reads + MFMAs {st[:, 1].mean():6.0f} | barrier wait {st[:, 2].mean():6.0f} | after barrier (K-step change) {tail.mean():6.0f}')
    for g in range(NG):
        print(f'    group {g:2d}: {st[g, 0]:6.0f} {st[g, 1]:6.0f} {st[g, 2]:6.0f} {tail[g]:6.0f}')
    print(f'  main loop total                      {m(3, 5):8.0f}   (ideal MFMA issue: {NG and (13 if sh else 25) * 2 * 6 * (8 if sh else 2) * (16 if sh else 32)} cycles)')
    print(f'  accumulate + bias + stores issued    {m(5, 6):8.0f}')
    print(f'  BatchNorm sums + final fold          {m(6, 7):8.0f}   (epilogue total {m(5, 7):.0f})')
