"""GPU box: per-parameter gradient error of the conv stacks in bf16 mode vs fp32 mode (same weights, same input)."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from module.vae_layers.conv import build_de_conv_layers
torch.manual_seed(0)
N = int(os.environ.get('N', 6))
for where, shape, name in (('input', (3, 64, 64), 'conv32+'), ('output', (8, 5, 5), 'deconv32+')):
    a = build_de_conv_layers(shape, name, batch_norm=True, where=where).cuda()
    b = build_de_conv_layers(shape, name, batch_norm=True, where=where).cuda()
    b.load_state_dict(a.state_dict())
    b.compute_dtype = 'bf16'
    x = torch.rand(N, *shape, device='cuda')
    ya, yb = a(x), b(x)
    print(name, 'fwd rel L2', float((ya - yb).norm() / ya.norm()))
    g = torch.randn_like(ya)
    ya.backward(g); yb.backward(g)
    for (n_, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if pa.grad is None: continue
        e = float((pa.grad - pb.grad).norm() / max(float(pa.grad.norm()), 1e-30))
        cos = float((pa.grad * pb.grad).sum() / (pa.grad.norm() * pb.grad.norm() + 1e-30))
        print(f'   {n_:12s} |g| {float(pa.grad.norm()):10.4g} rel {e:8.4f} cos {cos:7.4f}')
