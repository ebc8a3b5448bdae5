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


def main_resume():
    """Second fixture, tests/golden/ckpt_ref_e2: lr_decay = 0.1, two 'epochs' written the way train_model does it
    (cvae.py:2295-2296,2485-2493: history entry, epochs += 1, trained += 1, update_lr), saved by the reference, then
    RE-LOADED by the reference's own load() (cvae.py:2677-2857) - which restores history / test / ood records, sets
    trained = history['epochs'] and fast-forwards the lr scheduler - followed by one more step.  resume.npz holds what the
    reference had after its load (lr, trained) and after that step."""
    Net = import_reference()
    out_dir = os.path.join(REPO, 'tests', 'golden', 'ckpt_ref_e2')
    kw = dict(KW, optimizer=dict(KW['optimizer'], lr_decay=0.1))
    net = Net(**kw)
    load_det_state(net, 0)
    net.train()
    for epoch in range(2):
        o = step(net, 100 + epoch)
        net.train_history[epoch] = {'train_loss': {k: float(v.mean()) for k, v in o[2].items()},
                                    'train_measures': dict(o[3]), 'lr': net.optimizer.lr}
        net.train_history['epochs'] += 1
        net.trained += 1
        net.optimizer.update_lr()
    net.testing = {0: {'iws': {'n': 7, 'epochs': 2, 'accuracy': 0.25}}}
    net.save(out_dir)
    lr_at_save = net.optimizer.lr
    again = Net.load(out_dir, load_state=True)
    again.train()
    res = {'lr_at_save': np.float64(lr_at_save), 'lr_after_load': np.float64(again.optimizer.lr),
           'trained_after_load': np.int64(again.trained), 'history_epochs': np.int64(again.train_history['epochs'])}
    o = step(again, 4321)
    for k, v in o[2].items():
        res['loss.' + k] = v.detach().numpy()
    for n_, p in again.named_parameters():
        res['param_after.' + n_] = p.detach().numpy().copy()
    again.optimizer.update_lr()
    res['lr_after_next_epoch'] = np.float64(again.optimizer.lr)
    np.savez_compressed(os.path.join(out_dir, 'resume.npz'), **res)
    print({k: (float(v) if v.shape == () else v.shape) for k, v in res.items() if not k.startswith('param')})
    for f in sorted(os.listdir(out_dir)):
        print(f, os.path.getsize(os.path.join(out_dir, f)))


if __name__ == '__main__':
    if sys.argv[1:] == ['resume']:
        main_resume()
    else:
        main()
