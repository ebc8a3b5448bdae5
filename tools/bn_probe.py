"""GPU box: launch the BatchNorm(+ReLU) backward pair of the largest activation (imager.16: 1024 x 32 x 32 x 32 fp32) a few times.
Probe for tools/prof_kernel.sh (kernel trace, then --pmc FETCH_SIZE / WRITE_SIZE passes): HBM bytes vs the algorithmic 20 B/element."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import lib as L
lib = L.load()
N, C, H = 1024, 32, 32
dev = 'cuda'
x = torch.randn(N, C, H, H, device=dev); dy = torch.randn_like(x); dx = torch.empty_like(x)
gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
mean, invstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
gg, gb = torch.empty(C, device=dev), torch.empty(C, device=dev)
ws = L.workspace(lib.jvae_bn_workspace_bytes(C), x.device)
for _ in range(10):
    L.check(lib.jvae_bn_bwd_f32(L.ptr(dy), L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(mean), L.ptr(invstd), L.ptr(dx),
                                L.ptr(gg), L.ptr(gb), 0, N, C, H * H, 1, L.ptr(ws), ws.numel(), L.stream_ptr()), 'jvae_bn_bwd_f32')
torch.cuda.synchronize()
print('done', float(dx[0, 0, 0, 0]))
