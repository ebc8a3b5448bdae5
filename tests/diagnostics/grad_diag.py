"""Debug helper (GPU box): per-parameter gradient error of one golden case."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from oracle.cases import get_case
from oracle.det_init import det_inputs, load_det_state
from cvae import ClassificationVariationalNetwork as Net
name = sys.argv[1] if len(sys.argv) > 1 else 'c2_n8'
g = np.load(os.path.join(REPO, 'tests', 'golden', name + '.npz'))
case = get_case(name); kw = case['net']
net = Net(**kw); load_det_state(net, 0); net.to('cuda'); net.train()
x, y, eps = det_inputs(case['N'], kw['input_shape'], kw['num_labels'], net.latent_sampling, kw['latent_dim'],
                       uniform_eps=kw['prior'].get('distribution') == 'uniform')
net.optimizer.zero_grad()
out = net.evaluate(x.cuda(), y.cuda(), with_beta=True, kl_var_weighting=case['kl_var_weighting'],
                   gamma_weighting=case['gamma_weighting'], epsilon=eps.cuda())
def _r(a, b):
    return float(np.abs(a.detach().cpu().numpy().astype(np.float64) - b).max() / np.abs(b).max())
out7 = net.evaluate(x.cuda(), y.cuda(), with_beta=True, kl_var_weighting=case['kl_var_weighting'],
                   gamma_weighting=case['gamma_weighting'], epsilon=eps.cuda(), z_output=True)
print('fwd rel err: mu %.1e log_var %.1e z %.1e x_reco %.1e total %.1e kl %.1e' % (
    _r(out7[4], g['mu']), _r(out7[5], g['log_var']), _r(out7[6], g['z']), _r(out7[0], g['x_reco']),
    _r(out7[2]['total'], g['loss.total']), _r(out7[2]['kl'], g['loss.kl'])))
net.optimizer.zero_grad()
out[2]['total'].mean().backward()
for n, p in net.named_parameters():
    if p.grad is None: continue
    gn = float(p.grad.double().norm()); ref = float(g['gnorm.' + n]) if 'gnorm.' + n in g.files else float('nan')
    line = f'{n:40s} |g|={gn:12.5e} ref={ref:12.5e} relnorm={abs(gn-ref)/max(ref,1e-30):.2e}'
    if 'grad.' + n in g.files:
        r = g['grad.' + n]; d = np.abs(p.grad.cpu().numpy() - r).max() / max(np.abs(r).max(), 1e-30)
        line += f' maxrel={d:.2e}'
    print(line)
