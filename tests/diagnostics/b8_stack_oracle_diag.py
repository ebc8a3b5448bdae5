"""Diagnostic (not collected): one conv stack of config 5 in the product's bf16 mode against the oracle's bf16 emulation, layer by
layer in the forward pass (the product's tape holds the tensor in front of module i; a deferred BatchNorm is applied here the way
the consuming kernel does) and per parameter in the backward pass."""
import os, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
for p in (REPO, os.path.join(REPO, 'joint-vae_amd')):
    sys.path.insert(0, p)
from oracle import jvae_oracle as O                                   # noqa: E402
from module.vae_layers.conv import build_de_conv_layers              # noqa: E402

torch.manual_seed(0)
N = int(os.environ.get('N', 6))
for where, shape, name in (('input', (3, 64, 64), 'conv32+'), ('output', (8, 5, 5), 'deconv32+')):
    stack = build_de_conv_layers(shape, name, batch_norm=True, where=where).cuda()
    with torch.no_grad():
        for p_ in stack.parameters():
            if p_.dim() == 1:
                p_.add_(0.1 * torch.randn_like(p_))
    stack.compute_dtype = 'bf16'
    stack._debug_tape = tape = []
    x = torch.rand(N, *shape, device='cuda')
    y = stack(x)
    g = torch.randn_like(y)
    y.backward(g)
    layers = O.parse_stack(name, shape, where == 'output')
    P = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point and not k.endswith(('running_mean', 'running_var')))
         for k, v in stack.state_dict().items()}
    P = {('s.' + k): v for k, v in P.items()}
    # running statistics were already updated by the product's forward: use fresh copies for the oracle
    for k in list(P):
        if k.endswith('running_mean'): P[k] = torch.zeros_like(P[k])
        if k.endswith('running_var'): P[k] = torch.ones_like(P[k])
    O.TAPE = otape = []
    with O.bf16_convs():
        yo = O.run_stack(P, 's', layers, True, x.cpu(), 'linear' if where == 'output' else None, True)
    O.TAPE = None
    yo.backward(g.cpu())
    print(f'== {name}: output rel L2 {float((y.detach().cpu() - yo.detach()).norm() / yo.detach().norm()):.2e}')
    od = dict(otape)
    for (i, t, aff) in tape:
        key = f's.{i}'
        if key not in od:
            continue
        t = t.cpu()
        ref = od[key].detach()
        if aff is not None:         # the product holds the pre-BatchNorm tensor; the consumer applies relu(x*sc+sh) and rounds to bf16
            continue
        print(f'   after module {i:2d}: rel L2 {float((t - ref).norm() / ref.norm()):.2e}  max {float((t - ref).abs().max() / ref.abs().max()):.2e}')
    for n_, p_ in stack.named_parameters():
        if p_.grad is None:
            continue
        go = P['s.' + n_].grad
        if go is None or float(go.norm()) == 0:
            continue
        print(f'   grad {n_:12s} |g| {float(go.norm()):10.4g}  rel L2 {float((p_.grad.cpu() - go).norm() / go.norm()):.2e}')
