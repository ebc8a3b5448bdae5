#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — checkpoint fixture.  Build container only (needs /root/reference).

A tiny MLP CVAE of the *reference* takes one optimiser step, is saved with the reference's own `save()`
(params.json, train_params.json, state.pth, optimizer.pth, ...) into tests/golden/ckpt_ref/, then takes a SECOND
step whose losses / parameters are stored in next_step.npz.  The GPU test loads the reference's checkpoint into the
drop-in model (`ClassificationVariationalNetwork.load`), repeats that second step and must land on the same numbers;
it then saves with OUR save() and checks that the files have the keys / shapes the reference wrote.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
from oracle.gen_golden import import_reference, inject_eps      # noqa: E402
from oracle.det_init import load_det_state, det_inputs          # noqa: E402

KW = dict(input_shape=(1, 8, 8), num_labels=4, type='cvae', features=None, upsampler=None, encoder=[24], decoder=[24],
          classifier=[], batch_norm=False, latent_dim=6, latent_sampling=1, test_latent_sampling=1, sigma={'value': 0.5},
          gamma=0., beta=1., output_activation='sigmoid',
          prior=dict(distribution='gaussian', init_mean=0., learned_means=True, var_dim='diag', freeze_means=0),
          optimizer=dict(optim_type='adam', lr=1e-3, weight_decay=3e-5, grad_clipping=100))
N = 5


def step(net, seed):
    x, y, eps = det_inputs(N, KW['input_shape'], 4, 1, 6, seed=seed)
    net.optimizer.zero_grad()
    with inject_eps(eps):
        out = net.evaluate(x, y, with_beta=True)
    out[2]['total'].mean().backward()
    net.optimizer.clip(net.parameters())
    net.optimizer.step()
    return out


def main():
    Net = import_reference()
    out_dir = os.path.join(REPO, 'tests', 'golden', 'ckpt_ref')
    net = Net(**KW)
    load_det_state(net, 0)
    net.train()
    step(net, 1234)
    net.trained = 1                      # save() writes state.pth / optimizer.pth only for trained models
    net.save(out_dir)
    o = step(net, 4321)
    res = {'loss.' + k: v.detach().numpy() for k, v in o[2].items()}
    for n_, p in net.named_parameters():
        res['param_after.' + n_] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(out_dir, 'next_step.npz'), **res)
    for f in sorted(os.listdir(out_dir)):
        print(f, os.path.getsize(os.path.join(out_dir, f)))


if __name__ == '__main__':
    main()
