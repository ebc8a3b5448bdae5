#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — golden-vector generator.  Runs ONLY in the build container (needs /root/reference).

Imports the *reference* implementation (moxime/joint-vae, read-only at /root/reference) with inert
`torchvision` / `setproctitle` placeholder modules (neither is installed; neither is on the hot path, see
SURVEY.md §8c), builds the model of every case in oracle/cases.py, loads deterministic weights
(oracle/det_init.py), runs ONE training step exactly as cvae.py:2424-2461 does

    zero_grad -> evaluate(x, y, with_beta=True, kl_var_weighting=w, gamma_weighting=g) -> total.mean().backward()
    -> optimizer.clip(params) -> optimizer.step()

with epsilon injected (cvae's Sampling draws from the global RNG, layers.py:235), and writes the inputs'
seeds and the expected outputs to tests/golden/<case>.npz.  Only DATA is written - no reference source.

Usage:  python oracle/gen_golden.py [case ...]
"""
import os
import sys
import types
import contextlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = '/root/reference'
sys.path.insert(0, REPO)

from oracle.cases import CASES, DSL_EVAL_CASES, FULL_CASES, EVAL_CASES, EVAL_OOD_METHODS, WIM_CASES, get_case            # noqa: E402
from oracle.det_init import load_det_state, det_inputs  # noqa: E402

FULL_GRAD_MAX = 8192      # parameters up to this many elements get their full gradient stored


def import_reference():
    class _Dummy:
        def __init__(self, *a, **k):
            raise RuntimeError('placeholder for a package that is not installed')

    def _stub(name):
        m = types.ModuleType(name)

        def _ga(attr):
            if attr.startswith('__'):
                raise AttributeError(attr)
            return type(attr, (_Dummy,), {})
        m.__getattr__ = _ga
        return m
    tv = _stub('torchvision')
    for s in ('models', 'datasets', 'transforms', 'utils'):
        sub = _stub('torchvision.' + s)
        setattr(tv, s, sub)
        sys.modules['torchvision.' + s] = sub
    sys.modules['torchvision'] = tv
    sys.modules['setproctitle'] = _stub('setproctitle')
    sys.path.insert(0, REF)
    os.chdir(REF)                       # data/sets.ini, utils/filters.ini are cwd-relative
    from cvae import ClassificationVariationalNetwork
    return ClassificationVariationalNetwork


@contextlib.contextmanager
def inject_eps(eps):
    """Make the next torch.randn / torch.rand of shape eps.shape return `eps` (layers.py:233-237)."""
    real_randn, real_rand = torch.randn, torch.rand
    inv = 1.0 / float(np.sqrt(12))

    def fake_randn(*size, **kw):
        size = size[0] if len(size) == 1 and not isinstance(size[0], int) else size
        assert tuple(size) == tuple(eps.shape), (size, eps.shape)
        return eps.clone()

    def fake_rand(*size, **kw):          # reference computes (rand - 0.5) * sqrt(12)
        size = size[0] if len(size) == 1 and not isinstance(size[0], int) else size
        assert tuple(size) == tuple(eps.shape), (size, eps.shape)
        return eps.clone() * inv + 0.5
    torch.randn, torch.rand = fake_randn, fake_rand
    try:
        yield
    finally:
        torch.randn, torch.rand = real_randn, real_rand


def run_case(Net, name, compact=False):
    """compact=True (the full-size cases): the big tensors (x_reco, z, ...) are reduced to checksums - per-sample losses,
    measures, mu / log_var, gradient norms, small gradients and parameter norms are kept."""
    case = get_case(name)
    kw = case['net']
    N = case['N']
    torch.manual_seed(0)
    net = Net(**kw)
    load_det_state(net, seed=0)
    net.train()
    L = net.latent_sampling
    K = kw['latent_dim']
    C = kw['num_labels']
    uniform = kw['prior'].get('distribution') == 'uniform'
    x, y, eps = det_inputs(N, kw['input_shape'], C, L, K, uniform_eps=uniform)

    out = {}
    net.optimizer.zero_grad()
    with inject_eps(eps):
        x_reco, y_est, losses, measures, mu, log_var, z = net.evaluate(
            x, y, batch=0, with_beta=True, kl_var_weighting=case['kl_var_weighting'],
            gamma_weighting=case['gamma_weighting'], current_measures=None, z_output=True)
    if compact or x_reco.numel() > (1 << 20):
        # per-image mean and L2 norm of the reconstruction instead of the (L+1, N, C, H, W) tensor itself (also for the
        # (L+1, N, 256, C, H, W) level logits of a categorical decoder: 25 MB at N = 4)
        xr = x_reco.detach().double().flatten(2)
        out['x_reco_mean'] = xr.mean(-1).numpy()
        out['x_reco_norm'] = xr.norm(dim=-1).numpy()
        out['z_norm'] = z.detach().double().norm(dim=-1).numpy()
    else:
        out['x_reco'] = x_reco.detach().numpy()
        out['z'] = z.detach().numpy()
    out['y_est'] = y_est.detach().numpy()
    out['mu'] = mu.detach().numpy()
    out['log_var'] = log_var.detach().numpy()
    for k, v in losses.items():
        out['loss.' + k] = v.detach().numpy()
    for k, v in measures.items():
        out['measure.' + k] = np.float64(v)
    loss = losses['total'].mean()
    out['batch_loss'] = np.float64(loss.item())
    loss.backward()

    names = [n for n, _ in net.named_parameters()]
    has_grad = []
    for n_, p in net.named_parameters():
        if p.grad is None:
            continue
        has_grad.append(n_)
        g = p.grad.detach()
        out['gnorm.' + n_] = np.float64(g.double().norm().item())
        if g.numel() <= FULL_GRAD_MAX:
            out['grad.' + n_] = g.numpy().copy()
    out['param_names'] = np.array(names)
    out['grad_names'] = np.array(has_grad)
    total_norm = torch.sqrt(sum(p.grad.double().pow(2).sum() for p in net.parameters() if p.grad is not None))
    out['total_grad_norm'] = np.float64(total_norm.item())

    net.optimizer.clip(net.parameters())
    net.optimizer.step()
    for n_, p in net.named_parameters():
        out['pnorm_after.' + n_] = np.float64(p.detach().double().norm().item())
        if p.numel() <= FULL_GRAD_MAX:
            out['param_after.' + n_] = p.detach().numpy().copy()
    for n_, b in net.named_buffers():
        if b.dtype.is_floating_point:
            out['buffer_after.' + n_] = b.detach().numpy().copy()
        else:
            out['buffer_after.' + n_] = b.detach().numpy().copy()
    out['state_keys'] = np.array(list(net.state_dict().keys()))
    out['state_shapes'] = np.array([','.join(str(s) for s in v.shape) for v in net.state_dict().values()])
    out['nparams'] = np.int64(sum(p.numel() for p in net.parameters()))

    if compact:
        # The same backward by the reference in DOUBLE precision (net.double()): at N = 512 the reference's own fp32
        # gradients sit 1e-4 ... 1.5e-3 (relative L2 per tensor) from these - ReLU pre-activations within fp32 rounding
        # of zero take the other branch - so "distance to the fp64 gradient, relative to the reference's own" is the
        # meaningful bar for an independent fp32 implementation.
        torch.manual_seed(0)
        net64 = Net(**kw)
        load_det_state(net64, seed=0)
        net64.double()
        net64.train()
        net64.optimizer.zero_grad()
        with inject_eps(eps.double()):
            o64 = net64.evaluate(x.double(), y, batch=0, with_beta=True, kl_var_weighting=case['kl_var_weighting'],
                                 gamma_weighting=case['gamma_weighting'], current_measures=None, z_output=True)
        o64[2]['total'].mean().backward()
        for k, v in o64[2].items():
            out['loss64.' + k] = v.detach().numpy()
        for n_, p in net64.named_parameters():
            if p.grad is None:
                continue
            out['gnorm64.' + n_] = np.float64(p.grad.norm().item())
            if p.grad.numel() <= FULL_GRAD_MAX:
                out['grad64.' + n_] = p.grad.numpy().astype(np.float32)     # fp64 values rounded once
        out['total_grad_norm64'] = np.float64(torch.sqrt(sum(p.grad.pow(2).sum() for p in net64.parameters()
                                                              if p.grad is not None)).item())

    # second evaluate in eval mode on the updated model is NOT part of the train step -> not recorded
    path = os.path.join(REPO, 'tests', 'golden', name + '.npz')
    np.savez_compressed(path, **out)
    kb = os.path.getsize(path) / 1024
    print(f'{name}: total={out["batch_loss"]:.6f} kl={out["loss.kl"].mean():.5f} '
          f'|g|={out["total_grad_norm"]:.5f} nparams={out["nparams"]} -> {path} ({kb:.0f} KiB)')


def run_eval_case(Net, name):
    """evaluate(x) without labels in eval mode (cvae.py:548-600,793-873) + predict_after_evaluate (:938-970) +
    batch_dist_measures (:972-1085) of the reference."""
    case = get_case(name)
    kw = case['net']
    N = case['N']
    torch.manual_seed(0)
    net = Net(**kw)
    load_det_state(net, seed=0)
    net.eval()
    L = net.latent_sampling
    x, y, eps = det_inputs(N, kw['input_shape'], kw['num_labels'], L, kw['latent_dim'])
    out = {}
    with torch.no_grad(), inject_eps(eps):
        x_reco, y_est, losses, measures = net.evaluate(x, batch=0)
    out['L'] = np.int64(L)
    if L >= 8 or x_reco.numel() > (1 << 20):   # compact: per-image mean / norm of the (L+1, N, ...) reconstruction
        xr = x_reco.double().flatten(2)
        out['x_reco_mean'] = xr.mean(-1).numpy()
        out['x_reco_norm'] = xr.norm(dim=-1).numpy()
    else:
        out['x_reco'] = x_reco.numpy()
    out['y_est'] = y_est.numpy()
    for k, v in losses.items():
        out['loss.' + k] = v.numpy()
    for k, v in measures.items():
        out['measure.' + k] = np.float64(v)
    for m in net.predict_methods:
        out['predict.' + m] = net.predict_after_evaluate(y_est, losses, method=m).numpy()
    methods = EVAL_OOD_METHODS if name in EVAL_CASES else [m for m in EVAL_OOD_METHODS if m in net.ood_methods]
    out['ood_methods'] = np.array(methods)
    dm = net.batch_dist_measures(y_est, losses, methods)
    for k, v in dm.items():
        out['ood.' + k] = v.numpy()
    out['predict_methods'] = np.array(net.predict_methods)
    path = os.path.join(REPO, 'tests', 'golden', name + '.npz')
    np.savez_compressed(path, **out)
    probe = 'loss.iws' if 'loss.iws' in out else 'loss.total'
    print(f'{name}: L={L} {probe}[:3]={out[probe].reshape(-1)[:3]} -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)')


def run_wim_case(Net, name):
    """The WIM fine-tuning step with the reference's model class and prior factory, written out as ft/wim.py:215-255
    + ft/job.py:380-399 do it (the WIMJob class itself needs the job / dataset machinery of ft/)."""
    from module.priors import build_prior
    case = get_case(name)
    kw = case['net']
    N, K = case['N'], kw['latent_dim']
    torch.manual_seed(0)
    net = Net(**kw)
    load_det_state(net, seed=0)
    original = net.encoder.prior
    alternate = build_prior(dim=K, num_priors=1, **case['alternate_prior'])
    for p in alternate.parameters():
        p.requires_grad_(False)
    x_in, y_in, eps_in = det_inputs(N, kw['input_shape'], kw['num_labels'], 1, K, seed=1234)
    x_mix, _, eps_mix = det_inputs(N, kw['input_shape'], kw['num_labels'], 1, K, seed=777)
    y_mix = torch.zeros(N, dtype=int)
    out = {}
    net.optimizer.zero_grad()
    net.train()
    with inject_eps(eps_in):
        _, _, in_loss, _ = net.evaluate(x_in, y_in, batch=0, with_beta=True)
    L = in_loss['total'].mean()
    net.encoder.prior = alternate
    net.num_labels = 1
    net.train()
    with inject_eps(eps_mix):
        _, _, mix_loss, mix_meas = net.evaluate(x_mix, y_mix, batch=0, with_beta=True)
    L = L + case['alpha'] * mix_loss['total'].mean()
    net.encoder.prior = original
    net.num_labels = kw['num_labels']
    out['L'] = np.float64(L.item())
    for k, v in in_loss.items():
        out['in.' + k] = v.detach().numpy()
    for k, v in mix_loss.items():
        out['mix.' + k] = v.detach().numpy()
    for k, v in mix_meas.items():
        out['mixmeasure.' + k] = np.float64(v)
    L.backward()
    names = []
    for n_, p in net.named_parameters():
        if p.grad is not None:
            names.append(n_)
            out['gnorm.' + n_] = np.float64(p.grad.double().norm().item())
    out['grad_names'] = np.array(names)
    out['total_grad_norm'] = np.float64(torch.sqrt(sum(p.grad.double().pow(2).sum() for p in net.parameters()
                                                       if p.grad is not None)).item())
    net.optimizer.step()
    net.optimizer.clip(net.parameters())
    for n_, p in net.named_parameters():
        out['pnorm_after.' + n_] = np.float64(p.detach().double().norm().item())
    for n_, b in net.named_buffers():
        out['buffer_after.' + n_] = b.detach().numpy().copy()
    path = os.path.join(REPO, 'tests', 'golden', name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}: L={out["L"]:.5f} mix kl={out["mix.kl"].mean():.4f} -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)')


def main():
    names = sys.argv[1:] or (list(CASES) + list(EVAL_CASES) + list(WIM_CASES))
    Net = import_reference()
    torch.set_num_threads(8)
    for n in names:
        if n in FULL_CASES:
            run_case(Net, n, compact=True)
        elif n in WIM_CASES:
            run_wim_case(Net, n)
        elif n in EVAL_CASES or n in DSL_EVAL_CASES:
            run_eval_case(Net, n)
        else:
            run_case(Net, n)


if __name__ == '__main__':
    main()
