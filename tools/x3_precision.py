"""GPU box: error of the stride-1 5x5 kernels against an fp64 convolution - fp32 MFMA, split-bf16 32x32x16, split-bf16 16x16x32."""
import math, os, sys, torch
import torch.nn.functional as F
REPO = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import ops, lib as L
lib = L.load()
def rel(a, b):
    return float((a.double().cpu() - b).abs().max() / b.abs().max())
def rms(a, b):
    return float(((a.double().cpu() - b) ** 2).mean().sqrt() / (b ** 2).mean().sqrt())
for (cin, cout, tr, H, N) in ((32, 32, True, 32, 6), (32, 64, False, 16, 9), (64, 32, True, 16, 5), (64, 64, True, 8, 16)):
    for wide in (0, 1):
        g = torch.Generator().manual_seed(cin * 7 + cout + H)
        x = torch.randn(N, cin, H, H, generator=g)
        if wide: x = x * torch.exp(2 * torch.randn(N, cin, 1, 1, generator=g))
        w = torch.randn((cin, cout, 5, 5) if tr else (cout, cin, 5, 5), generator=g) / math.sqrt(cin * 25)
        b = torch.randn(cout, generator=g)
        ref = (F.conv_transpose2d if tr else F.conv2d)(x.double(), w.double(), b.double(), padding=2)
        spec = ops.ConvSpec(cin, cout, 5, 1, 2, 0, tr)
        xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
        line = f'cin {cin} cout {cout} H {H} wide {wide}:'
        for name, x3, sh in (('f32mfma', 0, 0), ('x3 32x32x16', 1, 0), ('x3 16x16x32', 1, 1)):
            lib.jvae_conv2d_set_split_bf16(x3); lib.jvae_conv2d_set_split_shape16(sh)
            y = ops.conv_fwd_raw(xd, wd, bd, spec)
            line += f'  {name}: max {rel(y, ref):.2e} rms {rms(y, ref):.2e}'
        print(line)
lib.jvae_conv2d_set_split_bf16(1); lib.jvae_conv2d_set_split_shape16(1)
