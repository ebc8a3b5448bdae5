"""GPU box: raw strided products of the dense / unfolded layers of config 2 (E4 = 7x7 head as a GEMM, D0, the latent heads)
for K-slice counts 1..32, split-bf16 (default) vs fp32-MFMA kernel (JVAE_X3=0)."""
import os, sys, torch
REPO = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import ops
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
SHAPES = [('E4 fwd  NT', 2048, 200, 3136, 'nt'), ('E4 dgrad NN', 2048, 3136, 200, 'nn'), ('E4 wgrad TN', 200, 3136, 2048, 'tn'),
          ('D0 fwd  NN', 1024, 4096, 64, 'nn'), ('D0 wgrad TN', 64, 4096, 1024, 'tn'), ('head fwd NT', 512, 64, 800, 'nt'),
          ('head wgrad TN', 64, 800, 512, 'tn')]
for name, M, N, K, lay in SHAPES:
    A = torch.randn((M, K) if lay[0] == 'n' else (K, M), device='cuda')
    B = torch.randn((K, N) if lay[1] == 'n' else (N, K), device='cuda')
    sA = (K, 1) if lay[0] == 'n' else (1, M)
    sB = (N, 1) if lay[1] == 'n' else (1, K)
    line = f'{name} M={M} N={N} K={K}: '
    for S in (1, 2, 4, 8, 16, 32):
        if K % S or K // S < 64: continue
        Ks = K // S
        part = torch.empty((S, M, N), device='cuda')
        t = timeit(lambda: ops.gemm(M, N, Ks, A, (sA[0], sA[1], Ks * sA[1]), B, (sB[0], sB[1], Ks * sB[0]), part, (N, 1, M * N), batch=S))
        line += f'S={S}: {t:6.1f} us ({2.0 * M * N * K / t / 1e6:5.1f} TF)  '
    print(line)
