"""Pins the CPU oracle (oracle/jvae_oracle.py) against the reference's own outputs (tests/golden/*.npz,
written by oracle/gen_golden.py which ran moxime/joint-vae itself).  Tolerance: 2e-5 relative fp32."""
import os

import numpy as np
import pytest
import torch

from oracle import jvae_oracle as O
from oracle.cases import CASES, EVAL_CASES, EVAL_OOD_METHODS, FULL_CASES, WIM_CASES, get_case
from oracle.det_init import det_inputs

RTOL = 2e-5


def _close(a, b, rtol=RTOL, floor=1e-30, what=''):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), floor)
    err = np.abs(a - b).max() / scale
    assert err <= rtol, f'{what}: rel err {err:.3e} (scale {scale:.3e})'


def _dead_bias(key, state_keys):
    """A conv bias directly in front of a BatchNorm has an exactly-zero true gradient (BN removes the mean):
    what the reference stores for it is rounding noise, so element-wise comparison is meaningless."""
    parts = key.split('.')
    if parts[-1] != 'bias' or not parts[-2].isdigit():
        return False
    nxt = '.'.join(parts[:-2] + [str(int(parts[-2]) + 1), 'running_mean'])
    return nxt in set(state_keys)


@pytest.mark.parametrize('name', list(CASES) + list(FULL_CASES))
def test_oracle_matches_reference(name, golden_dir):
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    case = get_case(name)
    sp = O.make_spec(**case['net'])
    P = O.init_state(sp, seed=0)
    # state_dict contract: same keys / shapes as the reference model
    keys = [k for k, _ in O.param_keys(sp)]
    assert keys == list(g['state_keys']), (keys, list(g['state_keys']))
    shapes = [','.join(str(s) for s in sh) for _, sh in O.param_keys(sp)]
    assert shapes == list(g['state_shapes'])
    uniform = case['net']['prior'].get('distribution') == 'uniform'
    x, y, eps = det_inputs(case['N'], sp['input_shape'], sp['C'], sp['L'], sp['K'], uniform_eps=uniform)
    opt = O.AdamState(sp)
    out, grads, gn = O.train_step(sp, P, opt, x, y, eps, case['kl_var_weighting'], case['gamma_weighting'])
    x_reco, y_est, losses, meas, mu, log_var, z = out
    _close(mu.detach(), g['mu'], what='mu')
    _close(log_var.detach(), g['log_var'], what='log_var')
    if 'x_reco' in g.files:
        _close(z.detach(), g['z'], what='z')
        _close(x_reco.detach(), g['x_reco'], what='x_reco')
    else:           # compact goldens of the full-size workloads: per-image checksums of the big tensors
        xr = x_reco.detach().double().flatten(2)
        _close(xr.mean(-1), g['x_reco_mean'], what='x_reco_mean')
        _close(xr.norm(dim=-1), g['x_reco_norm'], what='x_reco_norm')
        _close(z.detach().double().norm(dim=-1), g['z_norm'], what='z_norm')
    _close(y_est.detach(), g['y_est'], rtol=1e-4, what='y_est')
    for k in [f[5:] for f in g.files if f.startswith('loss.')]:
        if k == 'var_kl' and np.abs(g['loss.var_kl']).max() == 0:
            continue
        _close(losses[k].detach(), g['loss.' + k], what='loss.' + k)
    for k in [f[8:] for f in g.files if f.startswith('measure.')]:
        assert abs(meas[k] - float(g['measure.' + k])) <= 1e-4 * max(1.0, abs(float(g['measure.' + k]))), k
    assert set(grads) == set(g['grad_names'])
    _close(gn, g['total_grad_norm'], what='total grad norm')
    for k in g['grad_names']:
        gnorm = float(grads[k].double().norm())
        ref = float(g['gnorm.' + k])
        assert abs(gnorm - ref) <= 1e-4 * max(ref, 1e-3 * float(g['total_grad_norm'])), (k, gnorm, ref)
        if 'grad.' + k in g.files:
            if _dead_bias(k, g['state_keys']):
                continue
            _close(grads[k], g['grad.' + k], rtol=2e-4, floor=1e-6 * float(g['total_grad_norm']), what='grad.' + k)
    for k in g['param_names']:
        if 'param_after.' + k in g.files and not _dead_bias(k, g['state_keys']):   # Adam amplifies the noise sign
            _close(P[k].detach(), g['param_after.' + k], rtol=1e-5, what='param_after.' + k)
    for f in g.files:
        if f.startswith('buffer_after.'):
            _close(P[f[13:]].detach().double(), g[f], rtol=1e-5, what=f)


@pytest.mark.parametrize('name', list(EVAL_CASES))
def test_oracle_eval_path_matches_reference(name, golden_dir):
    """evaluate(x) without labels in eval mode: all-class losses, iws, predictions, OOD scores."""
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    case = get_case(name)
    sp = O.make_spec(**case['net'])
    P = O.init_state(sp, seed=0)
    L = int(g['L'])
    x, y, eps = det_inputs(case['N'], sp['input_shape'], sp['C'], L, sp['K'])
    with torch.no_grad():
        x_reco, y_est, losses, meas = O.evaluate_all_classes(sp, P, x, eps)
    if 'x_reco' in g.files:
        _close(x_reco, g['x_reco'], what='x_reco')
    else:
        xr = x_reco.double().flatten(2)
        _close(xr.mean(-1), g['x_reco_mean'], what='x_reco_mean')
        _close(xr.norm(dim=-1), g['x_reco_norm'], what='x_reco_norm')
    _close(y_est, g['y_est'], rtol=1e-4, what='y_est')
    for k in [f[5:] for f in g.files if f.startswith('loss.')]:
        assert tuple(losses[k].shape) == g['loss.' + k].shape, k
        _close(losses[k], g['loss.' + k], what='loss.' + k)
    for k in [f[8:] for f in g.files if f.startswith('measure.')]:
        assert abs(meas[k] - float(g['measure.' + k])) <= 1e-4 * max(1.0, abs(float(g['measure.' + k]))), k
    for m in g['predict_methods']:
        assert np.array_equal(O.predict(losses, y_est, str(m)).numpy(), g['predict.' + str(m)]), m
    scores = O.ood_scores(losses, sp['C'], EVAL_OOD_METHODS)
    for m in EVAL_OOD_METHODS:
        _close(scores[m], g['ood.' + m], rtol=5e-5, what='ood.' + m)


@pytest.mark.parametrize('name', list(WIM_CASES))
def test_oracle_wim_step_matches_reference(name, golden_dir):
    """the WIM fine-tuning step (two evaluate passes under the original / alternate prior)."""
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    case = get_case(name)
    kw = case['net']
    sp = O.make_spec(**kw)
    P = O.init_state(sp, seed=0)
    N, K = case['N'], kw['latent_dim']
    x_in, y_in, eps_in = det_inputs(N, sp['input_shape'], sp['C'], 1, K, seed=1234)
    x_mix, _, eps_mix = det_inputs(N, sp['input_shape'], sp['C'], 1, K, seed=777)
    alt = {'mean': torch.full((1, K), float(case['alternate_prior']['mean_shift'])), 'T': torch.ones(1)}
    o_in, o_mix, L, grads, gn = O.wim_step(sp, P, O.AdamState(sp), x_in, y_in, eps_in, x_mix, eps_mix, alt, case['alpha'])
    assert abs(L - float(g['L'])) <= 2e-5 * abs(float(g['L']))
    for k in [f[3:] for f in g.files if f.startswith('in.')]:
        _close(o_in[2][k].detach(), g['in.' + k], what='in.' + k)
    mix_keys = [f[4:] for f in g.files if f.startswith('mix.')]
    assert 'dzdist' not in mix_keys and set(mix_keys) == set(o_mix[2])
    for k in mix_keys:
        _close(o_mix[2][k].detach(), g['mix.' + k], what='mix.' + k)
    assert set(o_mix[3]) == {f[11:] for f in g.files if f.startswith('mixmeasure.')}
    _close(gn, g['total_grad_norm'], what='grad norm')
    for k in g['grad_names']:
        ref = float(g['gnorm.' + k])
        assert abs(float(grads[k].double().norm()) - ref) <= 1e-4 * max(ref, 1e-3 * float(g['total_grad_norm'])), k
    for f in g.files:
        if f.startswith('buffer_after.'):
            _close(P[f[13:]].detach().double(), g[f], rtol=1e-5, what=f)


def test_oracle_bf16_mode_rounds_where_the_product_does():
    """The bf16-emulating mode of the oracle (bf16_convs; the reference has no bf16 mode - this models the PRODUCT's arithmetic
    for BASELINE configs[4]): around 5x5 convolutions the stored activations are bf16 values, BatchNorm takes its batch statistics
    from the fp32 accumulators, other kernel sizes stay fp32, activation gradients are rounded where they are stored, parameter
    gradients are not; outside the context manager the oracle is the plain fp32 restatement again (bit for bit)."""
    import torch
    from oracle import jvae_oracle as O

    def is_bf16(t):
        return bool(torch.equal(t, t.to(torch.bfloat16).to(torch.float32)))
    torch.manual_seed(0)
    layers = O.parse_stack('32x5+2-64x5+2:2-8x3', (3, 16, 16), False)
    P = {}
    i = 0
    for d in layers:
        P[f's.{i}.weight'] = (torch.randn(d['c'], d['cin'], d['k'], d['k']) / (d['cin'] * d['k'] ** 2) ** 0.5).requires_grad_(True)
        P[f's.{i}.bias'] = (0.1 * torch.randn(d['c'])).requires_grad_(True)
        i += 1
        P[f's.{i}.weight'] = (1 + 0.1 * torch.randn(d['c'])).requires_grad_(True)
        P[f's.{i}.bias'] = (0.1 * torch.randn(d['c'])).requires_grad_(True)
        P[f's.{i}.running_mean'], P[f's.{i}.running_var'] = torch.zeros(d['c']), torch.ones(d['c'])
        P[f's.{i}.num_batches_tracked'] = torch.zeros((), dtype=torch.long)
        i += 2
    x = torch.rand(5, 3, 16, 16)

    def run(bf16):
        for k in list(P):
            if k.endswith('running_mean'): P[k] = torch.zeros_like(P[k])
            if k.endswith('running_var'): P[k] = torch.ones_like(P[k])
            if P[k].requires_grad: P[k].grad = None
        O.TAPE = tape = []
        try:
            if bf16:
                with O.bf16_convs():
                    y = O.run_stack(P, 's', layers, True, x, None, True)
            else:
                y = O.run_stack(P, 's', layers, True, x, None, True)
        finally:
            O.TAPE = None
        y.square().sum().backward()
        return y.detach().clone(), dict(tape), {k: v.grad.clone() for k, v in P.items() if v.requires_grad}
    y0, t0, g0 = run(False)
    y1, t1, g1 = run(True)
    y2, t2, g2 = run(False)
    assert torch.equal(y0, y2) and all(torch.equal(g0[k], g2[k]) for k in g0)        # the switch leaves no trace
    assert not O.BF16_CONV
    # 5x5 layers: stored conv output and stored activation are bf16 values; the 3x3 tail is fp32 again
    assert is_bf16(t1['s.0'].detach()) and is_bf16(t1['s.2'].detach()) and is_bf16(t1['s.3'].detach()) and is_bf16(t1['s.5'].detach())
    assert not is_bf16(t1['s.6'].detach()) and not is_bf16(y1)
    assert not is_bf16(t0['s.0'].detach())
    # close to fp32 at bf16 level, not equal; parameter gradients are fp32 values
    rel = float((y1 - y0).norm() / y0.norm())
    assert 1e-4 < rel < 3e-2, rel
    assert not is_bf16(g1['s.0.weight']) and float((g1['s.0.weight'] - g0['s.0.weight']).norm() / g0['s.0.weight'].norm()) < 0.1
    # BatchNorm statistics come from the fp32 accumulators: the running mean equals momentum * mean of the UNROUNDED conv output
    with O.bf16_convs():
        w = P['s.0.weight'].detach().to(torch.bfloat16).float()
        acc = torch.nn.functional.conv2d(x.to(torch.bfloat16).float(), w, P['s.0.bias'].detach(), padding=2)
    run(True)
    assert torch.allclose(P['s.1.running_mean'], 0.1 * acc.mean((0, 2, 3)), rtol=1e-5, atol=1e-7)
    assert not torch.allclose(P['s.1.running_mean'], 0.1 * acc.to(torch.bfloat16).float().mean((0, 2, 3)), rtol=1e-7, atol=0)


def test_bf16_emulation_fixture_is_what_the_oracle_computes(golden_dir):
    """tests/golden/c5_n256_bf16emu.npz (oracle/gen_bf16_golden.py) against the oracle run again at a batch that takes seconds:
    the same code path at N = 8 is deterministic, its fp32 twin reproduces the plain oracle, and the stored N = 256 file carries
    both the bf16-emulated and the fp32 step of the same inputs (the GPU test compares the product's bf16 mode with the former
    and reports its distance to the latter)."""
    import torch
    from oracle import jvae_oracle as O
    from oracle.det_init import det_inputs
    g = np.load(os.path.join(golden_dir, 'c5_n256_bf16emu.npz'))
    ref = np.load(os.path.join(golden_dir, 'c5_n256.npz'))
    # the fixture's fp32 half is the oracle's plain step: it agrees with the REFERENCE's own golden of the same case
    for k in ('total', 'cross_x', 'kl'):
        a, b = g['loss.' + k + '.fp32'].astype(np.float64), ref['loss.' + k].astype(np.float64)
        assert float(np.abs(a - b).max() / np.abs(b).max()) < 2e-5, k
    assert abs(float(g['total_grad_norm.fp32']) / float(ref['total_grad_norm']) - 1) < 1e-4
    # the bf16 half sits at bf16 distance from it: not equal, not far
    d = float((np.abs(g['loss.total'].astype(np.float64) - g['loss.total.fp32']) / np.abs(g['loss.total.fp32'])).max())
    assert 1e-4 < d < 2e-2, d
    case = get_case('c5_n256')
    kw = case['net']
    sp = O.make_spec(**kw)
    x, y, eps = det_inputs(8, kw['input_shape'], kw['num_labels'], 1, kw['latent_dim'])
    outs = []
    for _ in range(2):
        P = O.init_state(sp, seed=0)
        with O.bf16_convs():
            o, grads, gn = O.train_step(sp, P, O.AdamState(sp), x, y, eps)
        outs.append((o[2]['total'].detach().clone(), gn))
    assert torch.equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]
