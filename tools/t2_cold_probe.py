"""GPU box: the 4-phase kernel launched COLD every time - a 512 MB write + a few ms of idle in front of every launch - on D4 / D2 forward
(plain and with the deferred BatchNorm), 16 launches each; launches that differ from the majority output, and where."""
import os, sys, time, torch
REPO = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import ops
torch.manual_seed(0)
tag = os.path.basename(os.environ.get('JVAE_HIP_LIB', 'libjvae_hip.so')) + (' V1' if os.environ.get('JVAE_T2_V1') == '1' else '')
junk = torch.empty(128 * 1024 * 1024, device='cuda')
tot_bad = 0
for name, N, cin, cout, H in (('D4', 1024, 32, 32, 16), ('D2', 1024, 64, 64, 8), ('D4 n512', 512, 32, 32, 16)):
    spec = ops.ConvSpec(cin, cout, 5, 2, 2, 1, True)
    x = torch.randn(N, cin, H, H, device='cuda'); w = torch.randn(cin, cout, 5, 5, device='cuda') * 0.05
    b = torch.randn(cout, device='cuda')
    aff = (torch.rand(cin, device='cuda') + 0.5, torch.randn(cin, device='cuda') * 0.3, True)
    for k, f in (('plain', lambda: ops.conv_fwd_raw(x, w, b, spec)), ('aff', lambda: ops.conv_fwd_aff_raw(x, w, b, spec, aff, False)[0])):
        ys = []
        for r in range(16):
            junk.fill_(float(r)); torch.cuda.synchronize(); time.sleep(0.01)
            ys.append(f().clone())
        keys = [int(y.view(torch.int32).to(torch.int64).sum()) for y in ys]
        major = max(set(keys), key=keys.count)
        good = ys[keys.index(major)]
        odd = [i for i, kk in enumerate(keys) if kk != major]
        tot_bad += len(odd)
        n_el = [int((ys[i] != good).sum()) for i in odd]
        ch = sorted({int(c) for i in odd for c in (ys[i] != good).nonzero()[:, 1].tolist()})
        print(f'{tag} {name} {k}: cold launches that differ from the majority: {len(odd)} of 16, elements {n_el[:8]}, channels {ch[:8]}')
print(f'{tag}: TOTAL {tot_bad} bad launches of 96')
