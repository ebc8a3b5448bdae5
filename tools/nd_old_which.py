"""GPU box: WHAT the round-4 nondeterministic staging variant (nd_old/libjvae_old_V.so, tools/nd_old_variants.patch (-DND_VARIANT=0..3)) gets wrong.  Inputs that
make every output name the coefficient-table entry it was built from: x = 1 (or 0), one centre tap that copies input channel o % Cin
to output channel o, scale[c] = c + 1 (or shift[c] = c + 1): the expected output is (o % Cin) + 1 everywhere, exactly.
usage: JVAE_HIP_LIB=nd_old/libjvae_old_V.so python tools/nd_old_which.py"""
import collections, os, sys, torch
REPO = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, 'nd_old', 'pkg')]
from jvae_hip import ops
tag = os.path.basename(os.environ.get('JVAE_HIP_LIB', '?'))
for (name, N, cin, cout, H, tr) in (('D5', 1024, 32, 32, 32, True), ('E2', 512, 32, 64, 16, False)):
    spec = ops.ConvSpec(cin, cout, 5, 1, 2, 0, tr)
    w = torch.zeros((cin, cout, 5, 5) if tr else (cout, cin, 5, 5), device='cuda')
    for o in range(cout):
        if tr: w[o % cin, o, 2, 2] = 1.
        else: w[o, o % cin, 2, 2] = 1.
    b = torch.zeros(cout, device='cuda')
    ramp = torch.arange(1, cin + 1, device='cuda', dtype=torch.float32)
    want = ramp[torch.arange(cout, device='cuda') % cin].view(1, -1, 1, 1)
    for mode in ('scale', 'shift', 'pixel'):
        if mode == 'scale':
            x = torch.ones(N, cin, H, H, device='cuda'); aff = (ramp.clone(), torch.zeros(cin, device='cuda'), True); exp = want.expand(N, cout, H, H)
        elif mode == 'shift':
            x = torch.zeros(N, cin, H, H, device='cuda'); aff = (torch.ones(cin, device='cuda'), ramp.clone(), True); exp = want.expand(N, cout, H, H)
        else:       # every input element names its own (image, channel, row, column): a wrong pixel shows as another element's code
            x = (torch.arange(N * cin * H * H, device='cuda', dtype=torch.float32) % 65521.).view(N, cin, H, H)
            aff = (torch.ones(cin, device='cuda'), torch.zeros(cin, device='cuda'), True)
            exp = x[:, torch.arange(cout, device='cuda') % cin]
        for launch in range(3):
            y = ops.conv_fwd_aff_raw(x, w, b, spec, aff, True)[0]
            bad = (y != exp).nonzero()
            print(f'{tag} {name} {mode} launch {launch}: {bad.shape[0]} wrong of {y.numel()}')
            if bad.shape[0]:
                yb = y[bad[:, 0], bad[:, 1], bad[:, 2], bad[:, 3]]; eb = exp[bad[:, 0], bad[:, 1], bad[:, 2], bad[:, 3]]
                pairs = collections.Counter(zip(eb.tolist(), yb.tolist()))
                print('   (expected, got) x count, top 12:', pairs.most_common(12))
                for dim, nm in enumerate(('image', 'out channel', 'row', 'column')):
                    c = collections.Counter(bad[:, dim].tolist())
                    print(f'   by {nm}: {len(c)} distinct; top 8', c.most_common(8))
                print('   image % 8:', sorted(collections.Counter((bad[:, 0] % 8).tolist()).items()))
                print('   first 6:', [(tuple(bad[i].tolist()), float(eb[i]), float(yb[i])) for i in range(min(6, bad.shape[0]))])
