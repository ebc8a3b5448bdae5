import os, sys, numpy as np, torch
REPO = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from oracle.cases import get_case
from oracle.det_init import det_inputs, load_det_state
from cvae import ClassificationVariationalNetwork as Net
name = 'b2_n8_vib'
g = np.load(os.path.join(REPO, 'tests', 'golden', name + '.npz'))
case = get_case(name); kw = case['net']
net = Net(**kw); load_det_state(net, seed=0); net.to('cuda').train()
x, y, eps = (t.cuda() for t in det_inputs(case['N'], kw['input_shape'], kw['num_labels'], 1, kw['latent_dim']))
net.optimizer.zero_grad()
out = net.evaluate(x, y, with_beta=True, kl_var_weighting=case.get('kl_var_weighting', 1.), gamma_weighting=case.get('gamma_weighting', 1.), epsilon=eps)
out[2]['total'].mean().backward()
torch.cuda.synchronize()
tot = 0.
for n_, p in net.named_parameters():
    if p.grad is None or 'grad.' + n_ not in g.files: continue
    ref = g['grad.' + n_].astype(np.float64); mine = p.grad.double().cpu().numpy()
    d = mine - ref
    nz = np.abs(d) > 1e-4 * np.abs(ref).max()
    print(f'{n_:28s} |ref| {np.linalg.norm(ref):10.4f} rel L2 {np.linalg.norm(d)/max(np.linalg.norm(ref),1e-30):.2e}  norm ratio-1 {np.linalg.norm(mine)/max(np.linalg.norm(ref),1e-30)-1:+.2e}  elems off {int(nz.sum())}/{d.size}')
