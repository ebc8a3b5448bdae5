"""Diagnostic (not collected): the 24-step bf16-vs-fp32 trajectories of test_b8_training_sequence_tracks_fp32 for several
(data seed, noise seed) choices - worst / second-worst / median same-step difference, last-six-steps difference, largest per-sample
var_kl seen in the fp32 run (the exp(log sigma^2) outlier that made one seed spike)."""
import os, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
for p in (REPO, os.path.join(REPO, 'joint-vae_amd')):
    sys.path.insert(0, p)
from oracle.cases import get_case
from oracle.det_init import load_det_state
from cvae import ClassificationVariationalNetwork as Net
kw = get_case('c5_n4')['net']
N = 32
for dseed, nseed in ((1, 7), (2, 7), (3, 11), (4, 5), (5, 3), (6, 9), (11, 1), (12, 2)):
    torch.manual_seed(dseed)
    data = torch.rand(4, N, *kw['input_shape'], device='cuda')
    lab = torch.randint(0, kw['num_labels'], (4, N), device='cuda')
    def run(dtype):
        net = Net(**kw); load_det_state(net, seed=0); net.to('cuda').train(); net.set_compute_dtype(dtype)
        torch.manual_seed(nseed); torch.cuda.manual_seed(nseed)
        hist, vk = [], 0.
        for step in range(24):
            losses, _ = net.train_step(data[step % 4], lab[step % 4])
            hist.append(float(losses['total'].detach().mean()))
            vk = max(vk, float(losses['var_kl'].detach().max() / losses['var_kl'].detach().mean()))
        return hist, vk
    h32, vk32 = run('fp32'); h16, vk16 = run('bf16')
    diffs = sorted(abs(a - b) / abs(a) for a, b in zip(h32, h16))
    tail = abs(sum(h16[-6:]) - sum(h32[-6:])) / sum(h32[-6:])
    print(f'data {dseed:2d} noise {nseed:2d}: worst {diffs[-1]:.3f} second {diffs[-2]:.3f} median {diffs[12]:.3f} tail {tail:.4f} max var_kl/mean fp32 {vk32:.1f} bf16 {vk16:.1f} fall {h32[-1]/h32[0]:.2f}')
