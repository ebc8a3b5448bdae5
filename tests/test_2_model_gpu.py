"""End-to-end GPU parity of the drop-in model (cvae.ClassificationVariationalNetwork on HIP kernels):
  * against the reference's own outputs (tests/golden/*.npz, produced by oracle/gen_golden.py) on every case,
  * against the CPU oracle on larger seeded batches,
  * size-independent properties at BASELINE.json's full batch (N=512).
Tolerance (north_star): per-sample ELBO / KL / reconstruction within 1e-4 relative fp32."""
import os

import numpy as np
import pytest
import torch

from oracle import jvae_oracle as O
from oracle.cases import CASES, DSL_CASES, FULL_CASES, get_case, full_config
from oracle.det_init import det_inputs, load_det_state

pytestmark = pytest.mark.gpu
DEV = 'cuda'
RTOL = 1e-4


def rel(a, b, floor=1e-30):
    a = np.asarray(a.detach().double().cpu() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b.detach().double().cpu() if torch.is_tensor(b) else b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), floor))


def dead_bias(key, state_keys):
    parts = key.split('.')
    if parts[-1] != 'bias' or not parts[-2].isdigit():
        return False
    return '.'.join(parts[:-2] + [str(int(parts[-2]) + 1), 'running_mean']) in set(state_keys)


def build(case):
    from cvae import ClassificationVariationalNetwork as Net
    net = Net(**case['net'])
    load_det_state(net, seed=0)
    net.to(DEV)
    net.train()
    return net


@pytest.mark.parametrize('name', list(CASES) + list(DSL_CASES))
def test_train_step_matches_reference_golden(name, golden_dir):
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    case = get_case(name)
    net = build(case)
    kw = case['net']
    uniform = kw['prior'].get('distribution') == 'uniform'
    x, y, eps = det_inputs(case['N'], kw['input_shape'], kw['num_labels'], net.latent_sampling, kw['latent_dim'],
                           uniform_eps=uniform)
    x, y, eps = x.to(DEV), y.to(DEV), eps.to(DEV)
    net.optimizer.zero_grad()
    x_reco, y_est, losses, meas, mu, log_var, z = net.evaluate(
        x, y, batch=0, with_beta=True, kl_var_weighting=case['kl_var_weighting'],
        gamma_weighting=case['gamma_weighting'], z_output=True, epsilon=eps)
    assert bool((y.cpu() == det_inputs(case['N'], kw['input_shape'], kw['num_labels'], 1, 1)[1]).all())   # labels bit-exact
    assert rel(mu, g['mu']) < RTOL and rel(log_var, g['log_var']) < RTOL
    if 'x_reco' in g.files:
        assert rel(z, g['z']) < RTOL and rel(x_reco, g['x_reco']) < RTOL
    else:                       # large decoder outputs (256-level logits) are stored as per-image mean / norm
        xr = x_reco.detach().double().flatten(2)
        assert tuple(x_reco.shape[2:]) == tuple(net._reco_shape())
        assert rel(xr.mean(-1), g['x_reco_mean']) < RTOL and rel(xr.norm(dim=-1), g['x_reco_norm']) < RTOL
        assert rel(z.detach().double().norm(dim=-1), g['z_norm']) < RTOL
    assert rel(y_est, g['y_est']) < 5e-4
    for k in [f[5:] for f in g.files if f.startswith('loss.')]:
        if np.abs(g['loss.' + k]).max() == 0:
            assert float(losses[k].abs().max()) == 0.
            continue
        assert rel(losses[k], g['loss.' + k]) < RTOL, k
    for k in [f[8:] for f in g.files if f.startswith('measure.')]:
        ref = float(g['measure.' + k])
        assert abs(meas[k] - ref) <= 2e-4 * max(1.0, abs(ref)), (k, meas[k], ref)
    losses['total'].mean().backward()
    tot = float(g['total_grad_norm'])
    got = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
    assert set(g['grad_names']) <= set(got)
    # Gradients: two fp32 implementations cannot agree bit-wise on the forward pass, so a handful of ReLU
    # pre-activations within ~1e-6 of zero take the other branch (measured with tests/diagnostics/layer_diag.py: every
    # forward tensor agrees to <3e-6, the backward differs on isolated elements only).  Per-element max-abs
    # comparison is therefore meaningless for the backward; the bar is the relative L2 error per tensor
    # (<= 5e-2 at these tiny batches of 4-8 images, typically 1e-4) and the global gradient norm (<= 1e-4).  Kernel-level exactness is pinned
    # separately by tests/test_0_ops_gpu.py (1e-5 against PyTorch on random data).
    for k in g['grad_names']:
        ref = float(g['gnorm.' + k])
        assert abs(float(got[k].double().norm()) - ref) <= 1e-2 * max(ref, 1e-3 * tot), k
        if 'grad.' + k in g.files and not dead_bias(k, g['state_keys']):
            d = got[k].detach().double().cpu().numpy() - g['grad.' + k]
            assert np.linalg.norm(d) <= 5e-2 * max(ref, 1e-6 * tot), (k, np.linalg.norm(d), ref)
    net.optimizer.clip(net.parameters())
    assert abs(float(net.optimizer.grad_norm()) - tot) <= 1e-4 * tot
    net.optimizer.step()
    params = dict(net.named_parameters())
    lr = kw['optimizer']['lr']
    for k in g['param_names']:
        if 'param_after.' + k in g.files and not dead_bias(k, g['state_keys']):
            # first Adam step moves every weight by ~lr*sign(g): a gradient whose sign differs (see above) is off
            # by at most 2*lr; everything else must agree to fp32 rounding
            d = np.abs(params[k].detach().cpu().numpy().astype(np.float64) - g['param_after.' + k])
            assert d.max() <= 2.2 * lr, k
            assert (d > 1e-5 * max(1.0, np.abs(g['param_after.' + k]).max())).mean() <= 0.05, k
    bufs = dict(net.named_buffers())
    for f in g.files:
        if f.startswith('buffer_after.'):
            assert rel(bufs[f[13:]].double(), g[f]) < 2e-5, f


@pytest.mark.parametrize('name', list(FULL_CASES))
def test_full_size_step_matches_reference_golden(name, golden_dir):
    """BASELINE.json's full-size workloads (configs[1] / configs[2] at N = 512, the per-rank batch of configs[4] at
    N = 256 in fp32) against the REFERENCE's own training step at that size (compact goldens of oracle/gen_golden.py):
    every per-sample loss <= 1e-4 relative, labels bit-exact, the global gradient norm <= 1e-4, per-tensor gradient norms
    <= 1e-3, and every stored gradient no further from the reference's fp64 gradient than 3 x the reference's own fp32
    gradient is (see the comment at the backward bar)."""
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    case = get_case(name)
    net = build(case)
    kw = case['net']
    N = case['N']
    x, y, eps = det_inputs(N, kw['input_shape'], kw['num_labels'], net.latent_sampling, kw['latent_dim'])
    x, y, eps = x.to(DEV), y.to(DEV), eps.to(DEV)
    net.optimizer.zero_grad()
    x_reco, y_est, losses, meas, mu, log_var, z = net.evaluate(
        x, y, batch=0, with_beta=True, kl_var_weighting=case['kl_var_weighting'],
        gamma_weighting=case['gamma_weighting'], z_output=True, epsilon=eps)
    assert rel(mu, g['mu']) < RTOL and rel(log_var, g['log_var']) < RTOL
    xr = x_reco.detach().double().flatten(2)
    assert rel(xr.mean(-1), g['x_reco_mean']) < RTOL and rel(xr.norm(dim=-1), g['x_reco_norm']) < RTOL
    assert rel(z.detach().double().norm(dim=-1), g['z_norm']) < RTOL
    for k in [f[5:] for f in g.files if f.startswith('loss.')]:
        if np.abs(g['loss.' + k]).max() == 0:
            assert float(losses[k].abs().max()) == 0.
            continue
        assert tuple(losses[k].shape) == g['loss.' + k].shape, k
        assert rel(losses[k], g['loss.' + k]) < RTOL, k
    for k in [f[8:] for f in g.files if f.startswith('measure.')]:
        ref = float(g['measure.' + k])
        assert abs(meas[k] - ref) <= 2e-4 * max(1.0, abs(ref)), (k, meas[k], ref)
    losses['total'].mean().backward()
    tot = float(g['total_grad_norm'])
    got = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
    assert set(g['grad_names']) <= set(got)
    # The backward bar.  The goldens also hold the reference's own backward in DOUBLE precision (gen_golden.py): its
    # fp32 gradients sit 1e-4 ... 1.5e-3 (relative L2 per tensor, largest at the first encoder layer, the far end of the
    # chain) from the fp64 ones because ReLU pre-activations within fp32 rounding of zero take the other branch.  An
    # independent fp32 implementation cannot be closer to the reference's fp32 gradient than both are to the exact one,
    # so each tensor must be (a) within 1e-3 of the reference's fp32 norm (or no further from the fp64 norm than that
    # is), (b) for every stored tensor, element-wise L2:
    # no further from the reference's fp64 gradient than max(3 x the reference's own fp32 distance, 3e-4).  Measured
    # (tests/diagnostics/full_grad_diag.py): 0.7-2.0 x the reference's distance at N = 512 (worst 2.4e-3 vs 1.2e-3 on
    # features.1.bias), 0.75-1.3 x on the 64x64 geometry; norms within 6.6e-4.
    worst, worst_ref = 0., 0.
    for k in g['grad_names']:
        ref32, ref64 = float(g['gnorm.' + k]), float(g['gnorm64.' + k])
        mine = float(got[k].double().norm())
        if dead_bias(k, g['state_keys']):       # exact-zero gradient: the reference holds rounding noise, we hold 0
            assert mine <= max(1e-6 * tot, 2 * ref32), k
            continue
        # (a) the norm: within 1e-3 of the reference's fp32 value - or at least as close to the reference's fp64 value as
        # the reference's own fp32 value is.  Second clause (round 4): on c5_n256 the reference's fp32 norm of features.1.bias
        # is itself 1.11e-3 off its fp64 norm; the 32x32x16 kernels land at -2.6e-4 of fp64 (8.6e-4 from ref32), the 16x16x32
        # ones at +1.4e-4 of fp64 (1.25e-3 from ref32): closer to the exact gradient, further from the reference's rounding.
        # The second clause is bounded: even then the norm stays within 2e-3 of the reference's fp32 value.
        assert (abs(mine - ref32) <= 1e-3 * max(ref32, 1e-3 * tot)
                or (abs(mine - ref64) <= abs(ref32 - ref64)
                    and abs(mine - ref32) <= 2e-3 * max(ref32, 1e-3 * tot))), (k, mine, ref32, ref64)
        if 'grad64.' + k in g.files:
            g64 = g['grad64.' + k].astype(np.float64)
            d_ref = np.linalg.norm(g['grad.' + k].astype(np.float64) - g64) / max(ref64, 1e-4 * tot)
            d = np.linalg.norm(got[k].detach().double().cpu().numpy() - g64) / max(ref64, 1e-4 * tot)
            worst, worst_ref = max(worst, d), max(worst_ref, d_ref)
            assert d <= max(3 * d_ref, 3e-4), (k, d, d_ref)
    assert abs(tot - float(g['total_grad_norm64'])) <= 1e-5 * tot        # the reference's fp32 / fp64 norms agree ...
    net.optimizer.clip(net.parameters())
    assert abs(float(net.optimizer.grad_norm()) - tot) <= 1e-4 * tot
    net.optimizer.step()
    for n_, p in net.named_parameters():
        ref = float(g['pnorm_after.' + n_])
        # the first Adam step moves every weight by ~lr * sign(g): elements whose (rounding-level) gradient has the other
        # sign shift the norm by up to 2 lr |w| each - 1e-4 relative covers that (measured 2e-5); conv biases under a
        # BatchNorm (exact-zero gradient here, rounding noise in the reference) move by lr in unrelated directions
        if not dead_bias(n_, g['state_keys']):
            assert abs(float(p.detach().double().norm()) - ref) <= 1e-4 * max(ref, 1.0), n_
    bufs = dict(net.named_buffers())
    for f in g.files:
        if f.startswith('buffer_after.'):
            assert rel(bufs[f[13:]].double(), g[f]) < 2e-5, f
    print(f'{name}: worst per-tensor distance to the fp64 gradient: ours {worst:.2e}, reference fp32 {worst_ref:.2e}')


def test_per_dimension_sigma_is_refused_like_the_reference_fails():
    """a per-dimension sigma (learned, or coded as a mask) cannot run in the reference either - cvae.py:789
    adds a (C,H,W) log sigma to the (N,) wmse: RuntimeError for both kinds (probed on the reference in the build
    container).  The drop-in refuses it at construction; every scalar kind is built (goldens c2_n8_{rmse,coded,decay})."""
    from cvae import ClassificationVariationalNetwork as Net
    for sg in ({'value': 1.0, 'learned': True, 'sdim': (3, 32, 32)}, {'input_dim': (3, 32, 32), 'sdim': (3, 32, 32)}):
        with pytest.raises(NotImplementedError):
            Net(**dict(get_case('c2_n8')['net'], sigma=sg))


# N = 37, 49, 53: ragged last-batch sizes whose BatchNorm launch plans have EMPTY trailing image parts (the reference never
# drops the last batch, cvae.py:2245-2249; tests/test_abi_and_host.py sweeps the plans of every n <= 512 on the host)
@pytest.mark.parametrize('which,N', [(2, 64), (3, 48), (2, 37), (2, 49), (3, 53)])
def test_three_steps_against_oracle(which, N):
    """Three consecutive optimiser steps on a larger batch: losses track the CPU oracle step by step."""
    case = full_config(which, N)
    kw = case['net']
    net = build(case)
    sp = O.make_spec(**kw)
    P = O.init_state(sp, seed=0)
    opt = O.AdamState(sp)
    meas = None
    for step in range(3):
        x, y, eps = det_inputs(N, kw['input_shape'], kw['num_labels'], 1, kw['latent_dim'], seed=100 + 10 * step)
        out, grads, gn = O.train_step(sp, P, opt, x, y, eps)
        losses, meas = net.train_step(x.to(DEV), y.to(DEV), batch=step, current_measures=meas, epsilon=eps.to(DEV))
        # step 0 is a pure forward comparison (1e-4).  Adam's first updates are ~lr*sign(g) for EVERY weight, so
        # weights whose gradient is rounding noise (conv biases under BatchNorm, ReLU-flip neighbourhoods) move
        # the other way on a different fp32 implementation: later steps track per sample to a few 1e-2 (KL terms)
        # and to 2e-3 on the batch-mean ELBO.
        for k in ('total', 'cross_x', 'kl', 'zdist', 'var_kl', 'wmse', 'dzdist'):
            assert rel(losses[k], out[2][k]) < (RTOL if step == 0 else 5e-2), (step, k)
        assert abs(float(losses['total'].mean()) - float(out[2]['total'].mean())) < 2e-3 * float(out[2]['total'].mean())
        assert abs(float(net.optimizer.grad_norm()) - gn) < (2e-4 if step == 0 else 1e-2) * gn
        if step == 0:       # larger batch: the flipped-ReLU elements are a vanishing share of every gradient
            for n_, p_ in net.named_parameters():
                if n_ in grads and not dead_bias(n_, [k_ for k_, _ in O.param_keys(sp)]):
                    d = (p_.grad.detach().double().cpu() - grads[n_].double()).norm()
                    assert float(d) <= 1e-2 * max(float(grads[n_].double().norm()), 1e-6 * gn), n_


def test_full_batch_properties():
    """BASELINE config 2 at N=512: properties that do not need the oracle at full size."""
    case = full_config(2, 512)
    kw = case['net']
    net = build(case)
    x, y, eps = det_inputs(512, kw['input_shape'], 10, 1, 64, seed=7)
    x, y, eps = x.to(DEV), y.to(DEV), eps.to(DEV)
    _, _, l1, _ = net.evaluate(x, y, with_beta=True, epsilon=eps)
    # (1) the ELBO decomposes exactly: total = cross_x + beta * kl, kl = (zdist + var_kl) / 2
    assert rel(l1['total'], l1['cross_x'] + l1['kl']) < 1e-6
    assert rel(l1['kl'], 0.5 * (l1['zdist'] + l1['var_kl'])) < 1e-5
    # (2) a permutation of the batch permutes the per-sample losses (BatchNorm statistics are symmetric)
    perm = torch.randperm(512, device=DEV)
    _, _, l2, _ = net.evaluate(x[perm], y[perm], with_beta=True, epsilon=eps[:, perm])
    assert rel(l2['total'], l1['total'][perm]) < 2e-5
    assert rel(l2['kl'], l1['kl'][perm]) < 2e-5
    # (3) eps = 0 makes both decoded rows identical (row 0 is the mean path)
    xr, _, _, _ = net.evaluate(x, y, with_beta=True, epsilon=torch.zeros_like(eps))
    assert rel(xr[1], xr[0]) < 1e-6
    # (4) a step lowers the loss on the same batch (lr 1e-3, clip 100)
    before = float(l1['total'].mean())
    for _ in range(5):
        net.train_step(x, y, epsilon=eps)
    _, _, l3, _ = net.evaluate(x, y, with_beta=True, epsilon=eps)
    assert float(l3['total'].mean()) < before
    assert all(bool(torch.isfinite(p).all()) for p in net.parameters())


def test_checkpoint_roundtrip(tmp_path):
    case = get_case('c2_n8')
    kw = case['net']
    net = build(case)
    x, y, eps = det_inputs(8, kw['input_shape'], 10, 1, 64)
    x, y, eps = x.to(DEV), y.to(DEV), eps.to(DEV)
    net.train_step(x, y, epsilon=eps)
    json_only = ['history.json', 'ood.json', 'params.json', 'test.json', 'train_params.json']
    net.save(str(tmp_path / 'untrained'))                 # cvae.py:2667: tensors are written `if self.trained` only
    assert sorted(os.listdir(tmp_path / 'untrained')) == json_only
    net.trained = 1
    net.save(str(tmp_path / 'nostate'), except_state=True)
    assert sorted(os.listdir(tmp_path / 'nostate')) == json_only
    net.save(str(tmp_path / 'noopt'), except_optimizer=True)
    assert sorted(os.listdir(tmp_path / 'noopt')) == sorted(json_only + ['state.pth'])
    tmp_path = tmp_path / 'full'
    net.save(str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == ['history.json', 'ood.json', 'optimizer.pth', 'params.json', 'state.pth',
                                            'test.json', 'train_params.json']      # the reference's job-directory files
    other = build(case)
    other.load_weights(str(tmp_path))
    for (n1, p1), (n2, p2) in zip(net.named_parameters(), other.named_parameters()):
        assert torch.equal(p1, p2), n1                   # the checkpoint restores the parameters bit for bit
    la, _ = net.train_step(x, y, epsilon=eps)
    lb, _ = other.train_step(x, y, epsilon=eps)
    assert rel(la['total'], lb['total']) < 1e-6          # same state, same batch -> same forward
    # no float atomics on the training path (test_training_step_is_bit_reproducible): the restored model takes the
    # SAME step bit for bit
    for (n1, p1), (n2, p2) in zip(net.named_parameters(), other.named_parameters()):
        assert torch.equal(p1, p2), n1
    _, _, l1, _ = net.evaluate(x, y, with_beta=True, epsilon=eps)
    _, _, l2, _ = other.evaluate(x, y, with_beta=True, epsilon=eps)
    assert abs(float(l1['total'].mean()) - float(l2['total'].mean())) < 2e-3 * float(l2['total'].mean())


@pytest.mark.parametrize('name', ['c2_n8', 'c5_n4'])
def test_train_step_on_fp32_mfma_kernels_matches_golden(name, golden_dir):
    """The same parity bar with the split-bf16 convolution switched off (`jvae_conv2d_set_split_bf16(0)`): the stride-1
    layers then run on v_mfma_f32_32x32x2_f32 (conv_mfma.hip), the A/B partner of conv_x3.hip."""
    from jvae_hip import lib
    old = lib.load().jvae_conv2d_set_split_bf16(0)
    try:
        test_train_step_matches_reference_golden(name, golden_dir)
    finally:
        lib.load().jvae_conv2d_set_split_bf16(old)


from oracle.cases import DSL_EVAL_CASES, EVAL_CASES, EVAL_OOD_METHODS   # noqa: E402


@pytest.mark.parametrize('name', list(EVAL_CASES) + list(DSL_EVAL_CASES))
def test_eval_path_matches_reference_golden(name, golden_dir):
    """evaluate(x) without labels in eval mode (BatchNorm on running statistics, L = test sampling):
    all-class losses (C,N), the importance-weighted bound, predictions (bit-exact) and OOD scores."""
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    case = get_case(name)
    kw = case['net']
    from cvae import ClassificationVariationalNetwork as Net
    net = Net(**kw)
    load_det_state(net, seed=0)
    net.to(DEV)
    net.eval()
    L = int(g['L'])
    assert net.latent_sampling == L
    x, y, eps = det_inputs(case['N'], kw['input_shape'], kw['num_labels'], L, kw['latent_dim'])
    x_reco, y_est, losses, meas = net.evaluate(x.to(DEV), epsilon=eps.to(DEV))
    if 'x_reco' in g.files:
        assert rel(x_reco, g['x_reco']) < RTOL
    else:
        xr = x_reco.double().flatten(2)
        assert rel(xr.mean(-1), g['x_reco_mean']) < RTOL and rel(xr.norm(dim=-1), g['x_reco_norm']) < RTOL
    assert rel(y_est, g['y_est']) < 5e-4
    for k in [f[5:] for f in g.files if f.startswith('loss.')]:
        assert tuple(losses[k].shape) == g['loss.' + k].shape, k
        assert rel(losses[k], g['loss.' + k]) < RTOL, k
    for k in [f[8:] for f in g.files if f.startswith('measure.')]:
        ref = float(g['measure.' + k])
        assert abs(meas[k] - ref) <= 2e-4 * max(1.0, abs(ref)), (k, meas[k], ref)
    for m in g['predict_methods']:
        assert np.array_equal(net.predict_after_evaluate(y_est, losses, method=str(m)).cpu().numpy(), g['predict.' + str(m)])
    methods = [str(m) for m in g['ood_methods']] if 'ood_methods' in g.files else EVAL_OOD_METHODS
    scores = net.batch_dist_measures(y_est, losses, methods)
    for m in methods:
        assert rel(scores[m], g['ood.' + m]) < RTOL, m


def test_reference_checkpoint_interop(tmp_path, golden_dir):
    """a job directory written by the REFERENCE's save() (tests/golden/ckpt_ref, produced by
    oracle/gen_ckpt_fixture.py) loads into the drop-in model; the next optimiser step reproduces the reference's own
    next step (Adam moments, step count, lr restored); our save() writes the same files / keys / shapes back."""
    import json
    from cvae import ClassificationVariationalNetwork as Net
    src = os.path.join(golden_dir, 'ckpt_ref')
    g = np.load(os.path.join(src, 'next_step.npz'))
    net = Net.load(src, device=DEV)
    net.train()
    ref_state = torch.load(os.path.join(src, 'state.pth'), map_location='cpu')
    assert list(net.state_dict().keys()) == list(ref_state.keys())
    for k, v in net.state_dict().items():
        assert torch.equal(v.cpu(), ref_state[k]), k
    x, y, eps = det_inputs(5, (1, 8, 8), 4, 1, 6, seed=4321)
    losses, _ = net.train_step(x.to(DEV), y.to(DEV), epsilon=eps.to(DEV))
    for k in [f[5:] for f in g.files if f.startswith('loss.')]:
        assert rel(losses[k], g['loss.' + k]) < RTOL, k
    for n_, p in net.named_parameters():
        assert rel(p, g['param_after.' + n_], floor=1e-6) < 2e-5, n_     # second Adam step: moments came from the file
    assert net.trained == 0                              # history.json of the fixture: no finished epoch
    net.trained = 1                                      # as the fixture's generator did: tensors are saved `if self.trained`
    net.save(str(tmp_path))
    for f in ('params.json', 'train_params.json', 'test.json', 'ood.json', 'history.json', 'state.pth', 'optimizer.pth'):
        assert os.path.exists(os.path.join(tmp_path, f)), f
    mine, theirs = json.load(open(os.path.join(tmp_path, 'params.json'))), json.load(open(os.path.join(src, 'params.json')))
    assert mine == theirs
    o_m = torch.load(os.path.join(tmp_path, 'optimizer.pth'), map_location='cpu')
    o_r = torch.load(os.path.join(src, 'optimizer.pth'), map_location='cpu')
    assert sorted(o_m['state'].keys()) == sorted(o_r['state'].keys())
    for i in o_r['state']:
        assert o_m['state'][i]['exp_avg'].shape == o_r['state'][i]['exp_avg'].shape
        assert float(o_m['state'][i]['step']) == float(o_r['state'][i]['step']) + 1
    assert set(o_r['param_groups'][0]) <= set(o_m['param_groups'][0])


def test_wim_finetune_step_matches_reference_golden(golden_dir):
    """SURVEY.md §8f-4 (ft/wim.py:215-255, ft/job.py:380-399): the WIM fine-tuning step on the drop-in model, exactly as
    the reference's WIMJob drives its base class: swap `encoder.prior` for a non-conditional alternate prior
    (`build_prior(num_priors=1, ...)`, `num_labels = 1`), two evaluate passes, one backward, step-then-clip."""
    from oracle.cases import WIM_CASES
    from module.priors import build_prior
    name = 'w2_n8'
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    case = get_case(name)
    kw = case['net']
    net = build(case)
    N, K = case['N'], kw['latent_dim']
    x_in, y_in, eps_in = det_inputs(N, kw['input_shape'], kw['num_labels'], 1, K, seed=1234)
    x_mix, _, eps_mix = det_inputs(N, kw['input_shape'], kw['num_labels'], 1, K, seed=777)
    original = net.encoder.prior
    alternate = build_prior(dim=K, num_priors=1, **case['alternate_prior']).to(DEV)
    for p in alternate.parameters():
        p.requires_grad_(False)
    net.optimizer.zero_grad()
    _, _, in_loss, _ = net.evaluate(x_in.to(DEV), y_in.to(DEV), batch=0, with_beta=True, epsilon=eps_in.to(DEV))
    L = in_loss['total'].mean()
    net.encoder.prior, net.num_labels = alternate, 1
    _, _, mix_loss, mix_meas = net.evaluate(x_mix.to(DEV), torch.zeros(N, dtype=torch.int64, device=DEV), batch=0,
                                            with_beta=True, epsilon=eps_mix.to(DEV))
    L = L + case['alpha'] * mix_loss['total'].mean()
    net.encoder.prior, net.num_labels = original, kw['num_labels']
    assert abs(float(L) - float(g['L'])) <= RTOL * abs(float(g['L']))
    for k in [f[3:] for f in g.files if f.startswith('in.')]:
        assert rel(in_loss[k], g['in.' + k]) < RTOL, k
    mix_keys = [f[4:] for f in g.files if f.startswith('mix.')]
    assert set(mix_keys) == set(mix_loss)                       # no dictionary terms under a non-conditional prior
    for k in mix_keys:
        assert rel(mix_loss[k], g['mix.' + k]) < RTOL, k
    assert set(mix_meas.keys()) == {f[11:] for f in g.files if f.startswith('mixmeasure.')}
    for k in mix_meas.keys():
        ref = float(g['mixmeasure.' + k])
        assert abs(mix_meas[k] - ref) <= 2e-4 * max(1.0, abs(ref)), k
    L.backward()
    tot = float(g['total_grad_norm'])
    got = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
    assert set(g['grad_names']) <= set(got)
    for k in g['grad_names']:
        ref = float(g['gnorm.' + k])
        assert abs(float(got[k].double().norm()) - ref) <= 1e-2 * max(ref, 1e-3 * tot), k
    net.optimizer.step()                                        # sic: step, THEN clip (ft/job.py:397-399)
    net.optimizer.clip(net.parameters())
    assert abs(float(net.optimizer.grad_norm()) - tot) <= 1e-4 * tot
    bufs = dict(net.named_buffers())
    for f in g.files:
        if f.startswith('buffer_after.'):                       # BatchNorm saw two batches: running stats updated twice
            assert rel(bufs[f[13:]].double(), g[f]) < 2e-5, f


@pytest.mark.parametrize('name', ['c2_n8', 'c3_n8_diag', 'c3_n8_full', 'c2_n8_gamma'])
def test_training_step_is_bit_reproducible(name):
    """Two identical models on the same batch / epsilon: gradients after the first backward and parameters after two
    optimiser steps are BIT-identical (no float atomics anywhere on the path: K-sliced products and channel / class /
    norm reductions are folded in a fixed order; weight gradients on the side stream write disjoint slots; the gradient of
    a diag / full prior whitening factor is folded per class in sample order)."""
    case = get_case(name)
    kw = case['net']
    a, b = build(case), build(case)
    x, y, eps = (t.to(DEV) for t in det_inputs(case['N'], kw['input_shape'], kw['num_labels'], 1, kw['latent_dim']))
    for step in range(2):
        for net in (a, b):
            net.optimizer.zero_grad()
            out = net.evaluate(x, y, with_beta=True, epsilon=eps)
            out[2]['total'].mean().backward()
        torch.cuda.synchronize()
        for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
            if p.grad is not None:
                assert torch.equal(p.grad, q.grad), (step, k)
        for net in (a, b):
            net.optimizer.clip(net.parameters())
            net.optimizer.step()
        torch.cuda.synchronize()
        for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
            assert torch.equal(p, q), (step, k)


def test_adam_with_device_resident_step_matches_host_version():
    """jvae_adam_step_dev_f32 (step count / lr / betas in device memory) against jvae_adam_step_f32: bit-identical
    parameters and moments over 4 steps, including a mid-way learning-rate change written to the device block."""
    from jvae_hip import ops
    g = torch.Generator().manual_seed(3)
    n = 10_007
    p0 = torch.randn(n, generator=g)
    ps = [p0.clone().to(DEV) for _ in range(2)]
    ms = [torch.zeros(n, device=DEV) for _ in range(2)]
    vs = [torch.zeros(n, device=DEV) for _ in range(2)]
    lr = 1e-3
    hyper = torch.tensor([lr, 0.9, 0.999, 0., 0., 0.], device=DEV)
    for step in range(1, 5):
        grad = torch.randn(n, generator=g).to(DEV)
        if step == 3:
            lr = 5e-4
            hyper[0] = lr
        ops.adam_step(ps[0], grad, ms[0], vs[0], lr, 0.9, 0.999, 1e-8, 3e-5, step)
        ops.adam_step_dev(ps[1], grad, ms[1], vs[1], hyper, True, 1e-8, 3e-5)
        assert torch.equal(ps[0], ps[1]) and torch.equal(ms[0], ms[1]) and torch.equal(vs[0], vs[1]), step
    assert float(hyper[3]) == 4.0


def test_graph_captured_training_step():
    """graph_train_step(): the whole step (two streams, ~250 launches) captured into one HIP graph.  Replays keep
    training (loss falls), the device-resident Adam step count follows the host's, and an eager evaluation afterwards
    sees the trained weights."""
    case = full_config(2, 64)
    kw = case['net']
    net = build(case)
    x, y, _ = det_inputs(64, kw['input_shape'], kw['num_labels'], 1, kw['latent_dim'])
    x, y = x.to(DEV), y.to(DEV)
    step = net.graph_train_step(x, y, warmup=2)
    first = None
    for i in range(12):
        losses, meas = step(x, y)
        if first is None:
            first = float(losses['total'].detach().mean())
    last = float(losses['total'].detach().mean())
    assert np.isfinite(last) and last < first
    g0 = net.optimizer._groups[0]
    assert g0.step == 2 + 12 and float(g0.hyper[3]) == g0.step          # warm-up steps + replays, on both sides
    assert 0 < meas['rmse'] < 10 and meas['sigma'] > 0
    # the replayed graph really updates THE model's parameters: an eager evaluation (batch statistics, as in training:
    # after 14 steps the running statistics of eval mode are still far from them) sees the trained weights
    with torch.no_grad():
        _, _, l_eval, _ = net.evaluate(x, y, with_beta=True)
    assert float(l_eval['total'].mean()) < first


@pytest.mark.parametrize('where,spec,shape', [('input', '[x3-Mx2]8-M-16-A-Ax1', (3, 16, 16)),
                                              ('output', '[!x3+1-U:2]U-!8-U-!3', (4, 4, 4))])
@pytest.mark.parametrize('bn', [False, True])
def test_pooling_and_upsampling_stacks_match_torch(where, spec, shape, bn):
    """vgg-style (`M`, `A`) and ivgg-style (`U`, `!C`) layer strings (conv-models.ini:13-18,28-30) build and run on the
    HIP kernels: forward / backward against the same stack made of stock torch modules with the same state_dict."""
    import torch.nn as nn
    from module.vae_layers.conv import build_de_conv_layers
    torch.manual_seed(3)
    net = build_de_conv_layers(shape, spec, batch_norm=bn, where=where, output_activation='sigmoid').to('cuda')
    mods = []
    for m in net:
        n = type(m).__name__
        if isinstance(m, nn.Conv2d):
            mods.append(nn.Conv2d(m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding))
        elif isinstance(m, nn.BatchNorm2d):
            mods.append(nn.BatchNorm2d(m.num_features))
        elif n == 'HipPool2d':
            mods.append((nn.MaxPool2d if m.letter == 'M' else nn.AvgPool2d)(m.kernel_size, m.stride, m.padding))
        elif n == 'HipUpsamplingNearest2d':
            mods.append(nn.UpsamplingNearest2d(scale_factor=m.scale_factor))
        elif 'sigmoid' in n.lower():
            mods.append(nn.Sigmoid())
        elif 'relu' in n.lower():
            mods.append(nn.ReLU())
        else:
            mods.append(nn.Identity())
    ref = nn.Sequential(*mods)
    ref.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()})
    assert net.output_shape[1:] == tuple(ref(torch.zeros(2, *shape)).shape[2:])
    x = torch.randn(6, *shape)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    gy = torch.randn(yr.shape)
    yr.backward(gy)
    xd = x.cuda().requires_grad_(True)
    yd = net(xd)
    assert yd.shape == yr.shape
    yd.backward(gy.cuda())
    rel = lambda a, b: float((a.detach().cpu() - b.detach()).abs().max() / b.detach().abs().max())
    assert rel(yd, yr) < 2e-5 and rel(xd.grad, xr.grad) < 1e-4
    for (k, p), q in zip(net.named_parameters(), ref.parameters()):
        if bn and k.endswith('bias') and q.grad.abs().max() < 1e-4:
            # a conv bias in front of a train-mode BatchNorm: exact zero here, rounding noise in torch
            assert float(p.grad.abs().max()) < 1e-4, k
            continue
        assert rel(p.grad, q.grad) < 2e-4, k


@pytest.mark.parametrize('var_dim', ['scalar', 'diag', 'full'])
def test_prior_kl_dictionary_has_the_reference_keys(var_dim):
    """GaussianPrior.kl returns {trace, log_det_prior, log_det, distance, var_kl, kl} in the reference's order
    (module/priors.py:287-324); the three diagnostic terms (derived on first access) rebuild the kernel's var_kl."""
    from module.priors import build_prior
    K, C, N = 16, 5, 32
    torch.manual_seed(1)
    pr = build_prior(dim=K, distribution='gaussian', num_priors=C, init_mean=1., learned_means=True, var_dim=var_dim).to(DEV)
    with torch.no_grad():
        pr._var_parameter.add_(0.05 * torch.randn_like(pr._var_parameter))
    mu, lv = torch.randn(N, K, device=DEV), 0.3 * torch.randn(N, K, device=DEV)
    y = torch.randint(0, C, (N,), device=DEV)
    d = pr.kl(mu, lv, y, var_weighting=0.5)
    assert list(d) == ['trace', 'log_det_prior', 'log_det', 'distance', 'var_kl', 'kl']
    assert all(tuple(v.shape) == (N,) for v in d.values())
    assert rel(d['trace'] - d['log_det'] + d['log_det_prior'] - K, d['var_kl'], floor=1.) < 2e-5
    assert rel(0.5 * (d['distance'] + 0.5 * d['var_kl']), d['kl']) < 2e-5
    # all-class form: y (C, N) against mu (N, K)
    y_all = torch.arange(C, device=DEV).unsqueeze(1).expand(C, N)
    d2 = pr.kl(mu, lv, y_all)
    assert tuple(d2['kl'].shape) == (C, N) and tuple(d2['trace'].shape) == (C, N)


def test_resume_from_reference_checkpoint_with_history(tmp_path, golden_dir):
    """ADVICE r1 / SURVEY §8f-3: a job directory the REFERENCE saved after two epochs with lr_decay (tests/golden/ckpt_ref_e2,
    oracle/gen_ckpt_fixture.py resume) and then re-loaded itself.  load() must restore what the reference's load() restores
    (cvae.py:2745-2851): history / test records, trained = history['epochs'], the Adam state, and the learning rate after its
    scheduler fast-forward (sic: the reference decays the already decayed rate `trained` more times); the next step and the
    next epoch's rate must be the reference's; save() must keep the earlier epochs in history.json."""
    import json
    from cvae import ClassificationVariationalNetwork as Net
    src = os.path.join(golden_dir, 'ckpt_ref_e2')
    g = np.load(os.path.join(src, 'resume.npz'))
    net = Net.load(src, device=DEV)
    assert net.trained == int(g['trained_after_load']) == 2 and net.train_history['epochs'] == 2
    assert set(net.train_history) == {'epochs', 0, 1} and set(net.train_history[0]) == {'train_loss', 'train_measures', 'lr'}
    assert net.testing[0]['iws'] == {'n': 7, 'epochs': 2, 'accuracy': 0.25}
    assert net.optimizer.lr == pytest.approx(float(g['lr_after_load']), rel=1e-12)
    net.train()
    x, y, eps = det_inputs(5, (1, 8, 8), 4, 1, 6, seed=4321)
    losses, _ = net.train_step(x.to(DEV), y.to(DEV), epsilon=eps.to(DEV))
    for k in [f[5:] for f in g.files if f.startswith('loss.')]:
        assert rel(losses[k], g['loss.' + k]) < RTOL, k
    for n_, p in net.named_parameters():
        assert rel(p, g['param_after.' + n_], floor=1e-6) < 2e-5, n_        # third Adam step at the resumed rate
    net.optimizer.update_lr()
    assert net.optimizer.lr == pytest.approx(float(g['lr_after_next_epoch']), rel=1e-12)
    net.save(str(tmp_path))
    hist = json.load(open(os.path.join(tmp_path, 'history.json')))
    assert hist['epochs'] == 2 and set(hist) == {'epochs', '0', '1'}
    assert json.load(open(os.path.join(tmp_path, 'test.json')))['0']['iws']['accuracy'] == 0.25


def test_train_model_loop_saves_and_resumes(tmp_path):
    """train_model (cvae.py:2081-2547; the hot loop :2424-2501): two epochs on a synthetic TensorDataset with a KL warm-up,
    lr decay and frozen-then-thawed prior means; the console hook gets real running batch means (not NaN); history.json has
    the reference's entries; load() resumes at epoch 2 with the decayed rate and trains on to epoch 3."""
    import json
    from cvae import ClassificationVariationalNetwork as Net
    torch.manual_seed(0)
    kw = dict(get_case('c2_n8')['net'])
    kw['optimizer'] = dict(kw['optimizer'], lr_decay=0.1)
    kw['prior'] = dict(kw['prior'], freeze_means=1, init_mean=0.5)
    net = Net(**kw).to(DEV)
    data = torch.utils.data.TensorDataset(torch.rand(96, 3, 32, 32), torch.randint(0, 10, (96,)))

    class Out:
        rows = []

        def results(self, i, per_epoch, epoch, epochs, **k):
            self.rows.append((i, epoch, dict(k['losses']), dict(k['metrics'])))
    out = Out()
    m0 = net.encoder.prior.mean.detach().clone()
    hist = net.train_model(data, epochs=2, batch_size=32, warmup=[0, 1], save_dir=str(tmp_path), outputs=out,
                           report_every=2, validation=0, device=DEV)
    # cvae.py:2293-2296: the loop runs to epoch == epochs, whose entry holds the closing test / validation results only
    assert hist['epochs'] == 2 and net.trained == 2 and set(hist) == {'epochs', 0, 1, 2} and hist[2] == {}
    assert set(hist[0]) == {'train_loss', 'train_measures', 'lr'}
    assert set(hist[0]['train_loss']) >= {'total', 'cross_x', 'kl', 'zdist', 'var_kl', 'wmse'}
    assert all(np.isfinite(v) for v in hist[1]['train_loss'].values())
    assert hist[0]['lr'] == pytest.approx(1e-3) and hist[1]['lr'] == pytest.approx(9e-4) and net.optimizer.lr == pytest.approx(8.1e-4)
    assert hist[1]['train_loss']['total'] < hist[0]['train_loss']['total']
    assert len(out.rows) == 6 and out.rows[0][1] == 1 and out.rows[-1][1] == 2
    last = out.rows[-1]
    assert np.isfinite(last[2]['total']) and last[2]['total'] == pytest.approx(hist[1]['train_loss']['total'], rel=1e-6)
    assert np.isfinite(last[3]['rmse'])
    # freeze_means=1: the dictionary does not move in epoch 0 and does from epoch 1 on (priors.py:105-106,134-140)
    assert not torch.equal(net.encoder.prior.mean.detach(), m0)
    on_disk = json.load(open(os.path.join(tmp_path, 'history.json')))
    assert on_disk['epochs'] == 2 and set(on_disk) == {'epochs', '0', '1', '2'}
    again = Net.load(str(tmp_path), device=DEV)
    assert again.trained == 2 and again.train_history['epochs'] == 2
    for (k, p), (_, q) in zip(net.state_dict().items(), again.state_dict().items()):
        assert torch.equal(p, q), k
    hist2 = again.train_model(data, epochs=3, batch_size=32, warmup=[0, 1], save_dir=str(tmp_path), validation=0, device=DEV)
    assert hist2['epochs'] == 3 and set(hist2) == {'epochs', 0, 1, 2, 3} and again.trained == 3
    assert all(np.isfinite(v) for v in hist2[2]['train_loss'].values())


def test_train_model_epoch_with_a_ragged_last_batch(tmp_path):
    """The reference's DataLoader has no drop_last (cvae.py:2245-2249): 1 061 samples at batch_size 512 give batches of
    512, 512 and 37 - the last one a size whose BatchNorm plans leave trailing workgroups without images.  The epoch must
    finish with finite losses, every sample counted, and parameters that moved."""
    from cvae import ClassificationVariationalNetwork as Net
    torch.manual_seed(1)
    kw = dict(get_case('c2_n8')['net'])
    net = Net(**kw).to(DEV)
    data = torch.utils.data.TensorDataset(torch.rand(1061, 3, 32, 32), torch.randint(0, 10, (1061,)))
    seen = []

    class Out:
        def results(self, i, per_epoch, epoch, epochs, **k):
            seen.append((i, per_epoch, k.get('batch_size')))
    w0 = {k: v.detach().clone() for k, v in net.state_dict().items() if v.dtype.is_floating_point}
    hist = net.train_model(data, epochs=1, batch_size=512, save_dir=str(tmp_path), outputs=Out(), report_every=1, validation=0, device=DEV)
    torch.cuda.synchronize()
    assert hist['epochs'] == 1 and net.trained == 1
    assert all(np.isfinite(v) for v in hist[0]['train_loss'].values()), hist[0]['train_loss']
    assert seen and seen[-1][1] == 3                       # three batches per epoch: the ragged one is not dropped
    assert any(not torch.equal(v, w0[k]) for k, v in net.state_dict().items() if k in w0)
    assert all(bool(torch.isfinite(v).all()) for v in net.state_dict().values() if v.dtype.is_floating_point)
    # the ragged batch alone, against the oracle's forward (per-sample ELBO terms at 1e-4)
    case = full_config(2, 37)
    net2 = build(case)
    x, y, eps = det_inputs(37, (3, 32, 32), 10, 1, 64, seed=77)
    sp = O.make_spec(**case['net'])
    P = O.init_state(sp, seed=0)
    out, grads, gn = O.train_step(sp, P, O.AdamState(sp), x, y, eps)
    losses, _ = net2.train_step(x.to(DEV), y.to(DEV), epsilon=eps.to(DEV))
    for k in ('total', 'cross_x', 'kl'):
        assert rel(losses[k], out[2][k]) < RTOL, k
    assert abs(float(net2.optimizer.grad_norm()) - gn) < 2e-4 * gn


class _RawImages(torch.utils.data.Dataset):
    """uint8 HWC images + labels, as torchvision's CIFAR10 `.data` / `.targets`; remembers the order it was read in."""

    def __init__(self, n, side=32, classes=10, seed=0, name='cifar10'):
        rng = np.random.default_rng(seed)
        self.data = rng.integers(0, 256, size=(n, side, side, 3), dtype=np.uint8)
        self.targets = rng.integers(0, classes, size=n)
        self.name = name
        self.order = []

    def __len__(self):
        return len(self.data)

    def __getitem__(self, i):
        self.order.append(int(i))
        return torch.from_numpy(self.data[i]), int(self.targets[i])


def test_train_model_augments_on_the_device_bit_exact(monkeypatch):
    """VERDICT r3 item 3 / SURVEY.md §8f-2: train_model(data_augmentation=['flip', 'crop']) on a raw uint8 dataset feeds the
    step the batches that the reference's transform chain (RandomHorizontalFlip, RandomCrop(32, padding=4, 'edge'), ToTensor:
    utils/torch_load.py:405-426) would produce for the same per-image decisions - bit for bit, against
    oracle/augment_oracle.py.  The decisions are those of ops.draw_augmentation (recorded here), the batches those handed to
    train_step (recorded here); a ragged last batch included."""
    from cvae import ClassificationVariationalNetwork as Net
    from jvae_hip import ops
    from oracle.augment_oracle import augment
    torch.manual_seed(3)
    net = Net(**dict(get_case('c2_n8')['net'])).to(DEV)
    data = _RawImages(83, seed=11)
    draws, batches = [], []
    real_draw, real_step = ops.draw_augmentation, net.train_step

    def draw(*a, **k):
        out = real_draw(*a, **k)
        draws.append(tuple(None if t is None else t.cpu().numpy().copy() for t in out))
        return out

    def step(x, y, **k):
        batches.append((x.detach().cpu().numpy().copy(), y.cpu().numpy().copy()))
        return real_step(x, y, **k)
    monkeypatch.setattr(ops, 'draw_augmentation', draw)
    monkeypatch.setattr(net, 'train_step', step)
    hist = net.train_model(data, epochs=1, batch_size=32, data_augmentation=['flip', 'crop'], validation=0, device=DEV)
    assert hist['epochs'] == 1 and net.training_parameters['data_augmentation'] == ['flip', 'crop']
    data.order.pop(0)                                      # train_model's look at what the dataset yields (uint8 or float)
    assert [b[0].shape[0] for b in batches] == [32, 32, 19] and len(draws) == 3 and sorted(data.order) == list(range(83))
    at, flipped, shifted = 0, 0, 0
    for (xb, yb), (flip, dy, dx) in zip(batches, draws):
        idx = data.order[at:at + xb.shape[0]]
        at += xb.shape[0]
        ref = augment(data.data[idx], flip, dy, dx, 4)
        assert xb.dtype == np.float32 and np.array_equal(xb, ref)
        assert np.array_equal(yb, data.targets[idx])
        flipped += int(flip.sum())
        shifted += int(((dy != 4) | (dx != 4)).sum())
    assert 0 < flipped < 83 and shifted > 40               # both transforms really drew
    # no augmentation: the uint8 images are only converted (ToTensor)
    batches.clear(); draws.clear(); data.order.clear()
    net2 = Net(**dict(get_case('c2_n8')['net'])).to(DEV)
    step2 = net2.train_step
    monkeypatch.setattr(net2, 'train_step', lambda x, y, **k: (batches.append(x.detach().cpu().numpy().copy()), step2(x, y, **k))[1])
    net2.train_model(data, epochs=1, batch_size=64, validation=0, device=DEV)
    data.order.pop(0)
    assert np.array_equal(batches[0], data.data[data.order[:64]].transpose(0, 3, 1, 2).astype(np.float32) / np.float32(255))
    # 'crop' on an imagenet set pads by 0 (utils/torch_load.py:410): flips only
    batches.clear(); draws.clear(); data.order.clear()
    data.name = 'imagenet20'
    net3 = Net(**dict(get_case('c2_n8')['net'])).to(DEV)
    step3 = net3.train_step
    monkeypatch.setattr(net3, 'train_step', lambda x, y, **k: (batches.append(x.detach().cpu().numpy().copy()), step3(x, y, **k))[1])
    net3.train_model(data, epochs=1, batch_size=83, data_augmentation=['flip', 'crop'], validation=0, device=DEV)
    data.order.pop(0)
    flip, dy, dx = draws[0]
    assert dy is None and dx is None and np.array_equal(batches[0], augment(data.data[data.order], flip, None, None, 0))


def test_train_model_refuses_what_it_cannot_honour_and_clamps_the_batch(monkeypatch):
    """A float dataset cannot be augmented the reference's way (it augments the PIL image before ToTensor), an unknown token
    has no transform behind it (utils/torch_load.py:405-413 knows flip and crop): both raise instead of training silently
    un-augmented.  The training batch size is min(batch_size, max_batch_sizes['train']) (cvae.py:2180-2194)."""
    from cvae import ClassificationVariationalNetwork as Net
    net = Net(**dict(get_case('c2_n8')['net'])).to(DEV)
    floats = torch.utils.data.TensorDataset(torch.rand(24, 3, 32, 32), torch.randint(0, 10, (24,)))
    with pytest.raises(ValueError, match='uint8'):
        net.train_model(floats, epochs=1, batch_size=8, data_augmentation=['flip'], validation=0, device=DEV)
    net = Net(**dict(get_case('c2_n8')['net'])).to(DEV)
    with pytest.raises(ValueError, match='rotate'):
        net.train_model(_RawImages(8), epochs=1, batch_size=8, data_augmentation=['rotate'], validation=0, device=DEV)
    net = Net(**dict(get_case('c2_n8')['net'])).to(DEV)
    monkeypatch.setattr(Net, 'max_batch_sizes', property(lambda self: {'train': 8, 'test': 4}))
    sizes = []
    real_step = net.train_step
    monkeypatch.setattr(net, 'train_step', lambda x, y, **k: (sizes.append(x.shape[0]), real_step(x, y, **k))[1])
    net.train_model(floats, epochs=1, batch_size=512, validation=0, device=DEV)
    assert sizes == [8, 8, 8] and net.training_parameters['batch_size'] == 8


class _CifarLike(torch.utils.data.Dataset):
    """What utils/torch_load.get_dataset('cifar10', transformer='default') hands to train.py:236-243: a torchvision-CIFAR-shaped
    object - raw images in `.data` (uint8 NHWC) / `.targets` (list), `__getitem__` -> (ToTensor(image), label) - with the
    attributes the reference's factory adds (`.name .transformer .same_size .classes .heldout`, torch_load.py:489-497)."""
    target_transform = None

    def __init__(self, n, name='cifar10', seed=0, classes=10):
        rng = np.random.default_rng(seed)
        self.data = rng.integers(0, 256, size=(n, 32, 32, 3), dtype=np.uint8)
        self.targets = [int(t) for t in rng.integers(0, classes, size=n)]
        self.name, self.transformer, self.same_size = name, 'default', ['svhn']
        self.classes, self.heldout = [str(i) for i in range(classes)], []

    def __len__(self):
        return len(self.data)

    def __getitem__(self, i):
        return torch.from_numpy(self.data[i]).permute(2, 0, 1).to(torch.float32).div(255), self.targets[i]


def _keep_job(src, name, with_tensors):
    """JVAE_KEEP_JOB_DIR (set by tools/gpu_job.sh): the job directory a test's train_model() wrote is copied there, so that
    oracle/check_dropin_job.py can hand it to the REFERENCE's load() in the build container."""
    keep = os.environ.get('JVAE_KEEP_JOB_DIR')
    if not keep:
        return
    import shutil
    dst = os.path.join(keep, name)
    shutil.rmtree(dst, ignore_errors=True)
    os.makedirs(dst)
    for f in os.listdir(src):
        if f.endswith('.json') or (with_tensors and f.endswith('.pth')):
            shutil.copy(os.path.join(src, f), dst)


def test_train_model_as_train_py_drives_it(tmp_path, monkeypatch, caplog):
    """the call of train.py:333-351 - a FLOAT (ToTensor) CIFAR-shaped dataset object that carries its raw
    images, data_augmentation=['flip', 'crop'], a test set, OOD sets, a validation hold-out - through the drop-in.
    Bookkeeping of cvae.py:2108-2167 (what train.py:224-229 reads back on --resume), the seeded split (the held-out samples
    are never trained on), augmentation of the carried raw images bit-exact against the oracle, the test / validation phases
    of cvae.py:2293-2382 with their history entries and record files, one warning for the OOD phase."""
    import json
    import logging
    from cvae import ClassificationVariationalNetwork as Net
    from jvae_hip import ops
    from oracle.augment_oracle import augment
    torch.manual_seed(5)
    net = Net(**dict(get_case('c2_n8')['net'])).to(DEV)
    trainset, testset, ood = _CifarLike(600, seed=21), _CifarLike(130, seed=22), _CifarLike(40, name='svhn', seed=23)
    index_of = {trainset.data[i].tobytes(): i for i in range(len(trainset))}
    raw, seen_y = [], []
    real_aug, real_step = ops.augment_batch, net.train_step

    def aug(x, flip, dy, dx, **k):
        out = real_aug(x, flip, dy, dx, **k)
        raw.append((x.cpu().numpy().copy(), flip.cpu().numpy().copy(), dy.cpu().numpy().copy(), dx.cpu().numpy().copy(),
                    dict(k), out.cpu().numpy().copy()))
        return out
    monkeypatch.setattr(ops, 'augment_batch', aug)
    monkeypatch.setattr(net, 'train_step', lambda x, y, **k: (seen_y.append(y.cpu().numpy().copy()), real_step(x, y, **k))[1])

    class Out:
        rows = []

        def results(self, i, per_epoch, epoch, epochs, **k):
            self.rows.append((k.get('preambule'), i, per_epoch))

    class Sig:
        sig = 0
    save_dir = str(tmp_path / 'job')
    with caplog.at_level(logging.WARNING):
        hist = net.train_model(trainset, transformer=trainset.transformer, epochs=2, batch_size=64, test_batch_size=50,
                               full_test_every=1, ood_detection_every=1, validation=88, device=DEV, testset=testset,
                               oodsets=[ood], data_augmentation=['flip', 'crop'], fine_tuning=False, warmup=[0, 0],
                               warmup_gamma=[0, 0], validation_sample_size=64, save_dir=save_dir, outputs=Out(),
                               signal_handler=Sig())
    assert sum('OOD' in r.getMessage() for r in caplog.records) == 1
    tp = net.training_parameters
    assert (tp['set'], tp['transformer'], tp['validation'], tp['full_test_every']) == ('cifar10', 'default', 88, 1)
    assert tp['data_augmentation'] == ['flip', 'crop'] and tp['epochs'] == 2 and tp['batch_size'] == 64
    assert 0 <= tp['validation_split_seed'] < 2 ** 12 and tp['warmup'] == [0, 0]
    # the split: 512 of the 600 samples per epoch, the SAME 512 in both epochs, each once; labels travel with their images
    per_epoch = [raw[:8], raw[8:]]
    assert len(raw) == 16 and all(sum(b[0].shape[0] for b in ep) == 512 for ep in per_epoch)
    trained_on = []
    at = 0
    for ep in per_epoch:
        ids = []
        for xb, flip, dy, dx, kw, out in ep:
            assert xb.dtype == np.uint8 and xb.shape[1:] == (32, 32, 3) and kw.get('pad') == 4 and kw.get('nhwc')
            idx = [index_of[im.tobytes()] for im in xb]
            assert np.array_equal(out, augment(trainset.data[idx], flip, dy, dx, 4))            # bit-exact input pipeline
            assert np.array_equal(seen_y[at], np.asarray(trainset.targets)[idx])
            at += 1
            ids += idx
        assert len(set(ids)) == 512
        trained_on.append(set(ids))
    assert trained_on[0] == trained_on[1]
    held_out = set(range(600)) - trained_on[0]
    split = torch.utils.data.random_split(range(600), [88, 512], generator=torch.Generator().manual_seed(tp['validation_split_seed']))
    assert held_out == set(split[0]) and len(held_out) == 88                                    # cvae.py:2164-2167
    # phases and history (cvae.py:2293-2382,2481-2501)
    assert hist['epochs'] == 2 and set(hist) == {'epochs', 0, 1, 2}
    assert set(hist[0]) == {'validation_accuracy', 'validation_measures', 'validation_loss', 'train_loss', 'train_measures', 'lr'}
    for e in (1, 2):
        assert {'test_accuracy', 'test_measures', 'test_loss', 'validation_accuracy'} <= set(hist[e])
        assert set(hist[e]['test_accuracy']) == set(net.predict_methods)
        assert all(0. <= v <= 1. for v in hist[e]['test_accuracy'].values())
        assert np.isfinite(hist[e]['test_loss']['total']) and np.isfinite(hist[e]['validation_loss']['total'])
    assert 'train_loss' in hist[1] and 'train_loss' not in hist[2]
    assert net.testing[2][net.predict_methods[0]]['n'] == 130
    pre = [r[0] for r in Out.rows]
    assert pre.count('train') == 16 and 'TEST' in pre and ('VALID' in pre or 'valid' in pre)
    for d in ('last', '0001', '0002'):
        for s in ('cifar10', 'validation'):
            assert os.path.exists(os.path.join(save_dir, 'samples', d, 'record-{}.pth'.format(s))), (d, s)
    from jvae_compat.recorders import LossRecorder
    rec = LossRecorder.load(os.path.join(save_dir, 'samples', 'last', 'record-cifar10.pth'))
    assert rec.recorded_samples == 130 and np.array_equal(rec['y_true'].cpu().numpy(), np.asarray(testset.targets))
    # what train.py:224-229 reads on --resume, from the files
    on_disk = json.load(open(os.path.join(save_dir, 'train_params.json')))
    for k in ('set', 'transformer', 'validation', 'data_augmentation', 'latent_sampling'):
        assert on_disk[k] == tp[k], k
    again = Net.load(save_dir, device=DEV)
    assert again.trained == 2 and again.training_parameters['set'] == 'cifar10' and again.training_parameters['validation'] == 88
    _keep_job(save_dir, 'cifar_conv32', with_tensors=False)
    # a float dataset WITHOUT raw images, or one whose own transform is more than ToTensor, still raises
    net = Net(**dict(get_case('c2_n8')['net'])).to(DEV)
    class Inverted(_CifarLike):
        def __getitem__(self, i):
            x, y = super().__getitem__(i)
            return 1 - x, y
    bad = Inverted(64, seed=3)
    with pytest.raises(ValueError, match='more than ToTensor'):
        net.train_model(bad, epochs=1, batch_size=8, data_augmentation=['flip'], validation=0, device=DEV)


def test_train_model_job_of_a_small_model_for_the_reference_to_load(tmp_path):
    """A complete job directory (JSON files + state.pth + optimizer.pth) written by the drop-in's train_model() for a model
    small enough to commit: oracle/check_dropin_job.py loads it with the REFERENCE's own load() (cvae.py:2677-2857) in the
    build container.  Here: it is written, and the drop-in resumes from it."""
    from cvae import ClassificationVariationalNetwork as Net
    kw = dict(input_shape=(1, 8, 8), num_labels=4, type='cvae', features=None, upsampler=None, encoder=[24], decoder=[24],
              classifier=[], batch_norm=False, latent_dim=6, latent_sampling=1, test_latent_sampling=2, sigma={'value': 0.5},
              gamma=0., beta=1., output_activation='sigmoid',
              prior=dict(distribution='gaussian', init_mean=0., learned_means=True, var_dim='diag', freeze_means=0),
              optimizer=dict(optim_type='adam', lr=1e-3, weight_decay=3e-5, grad_clipping=100, lr_decay=0.1))
    torch.manual_seed(2)
    net = Net(**kw).to(DEV)
    g = torch.Generator().manual_seed(0)
    data = torch.utils.data.TensorDataset(torch.rand(80, 1, 8, 8, generator=g), torch.randint(0, 4, (80,), generator=g))
    data.name, data.transformer = 'toy8', 'simple'
    test = torch.utils.data.TensorDataset(torch.rand(20, 1, 8, 8, generator=g), torch.randint(0, 4, (20,), generator=g))
    test.name = 'toy8'
    save_dir = str(tmp_path / 'job')
    hist = net.train_model(data, epochs=2, batch_size=16, test_batch_size=10, validation=16, testset=test, full_test_every=1,
                           save_dir=save_dir, device=DEV)
    assert hist['epochs'] == 2 and net.training_parameters['set'] == 'toy8' and net.training_parameters['transformer'] == 'simple'
    for f in ('params.json', 'train_params.json', 'test.json', 'ood.json', 'history.json', 'state.pth', 'optimizer.pth'):
        assert os.path.exists(os.path.join(save_dir, f)), f
    again = Net.load(save_dir, device=DEV)
    # (load() fast-forwards the restored learning rate by `trained` epochs once more, as the reference's does: ckpt_ref_e2)
    assert again.trained == 2 and again.optimizer.lr == pytest.approx(net.optimizer.lr * 0.9 ** 2)
    _keep_job(save_dir, 'toy8_mlp', with_tensors=True)


def test_encoder_value_error_dumps_model_and_batch(tmp_path, monkeypatch):
    """cvae.py:476-488: a ValueError raised by the encoder leaves log/dump-<job> with the model files and x.pt / y.pt,
    and propagates."""
    case = get_case('c2_n8')
    net = build(case)
    net.job_number = 4242
    net.trained = 1                                        # an untrained net dumps its json files only (cvae.py:2667)
    x, y, eps = (t.to(DEV) for t in det_inputs(8, (3, 32, 32), 10, 1, 64))
    monkeypatch.chdir(tmp_path)

    def boom(*a, **k):
        raise ValueError('nan in the trunk')
    monkeypatch.setattr(net.encoder, 'encode', boom)
    with pytest.raises(ValueError, match='nan in the trunk'):
        net.evaluate(x, y, with_beta=True, epsilon=eps)
    where = os.path.join(tmp_path, 'log', 'dump-4242')
    assert {'params.json', 'state.pth', 'optimizer.pth', 'x.pt', 'y.pt'} <= set(os.listdir(where))
    assert torch.equal(torch.load(os.path.join(where, 'x.pt')).cpu(), x.cpu())


def test_label_free_evaluation_with_coded_labels_is_refused_like_the_reference_fails():
    """evaluate(x) without labels for a model whose labels are coded into the encoder (jvae / y_is_coded): the reference
    raises `RuntimeError: shape '[N]' is invalid for input of size C*N` at cvae.py:451 (probed on the reference, conv and MLP
    models); the drop-in refuses as well.  With labels the same model evaluates (golden j2_n8_jvae)."""
    net = build(get_case('j2_n8_jvae'))
    net.eval()
    x, y, eps = (t.to(DEV) for t in det_inputs(8, (3, 32, 32), 10, net.latent_sampling, 64))
    with pytest.raises(NotImplementedError):
        net.evaluate(x)
    out = net.evaluate(x, y, epsilon=eps)
    assert tuple(out[2]['total'].shape) == (8,)


def test_label_free_evaluation_decodes_in_slabs(monkeypatch):
    """The (L+1)*N decoder batch of evaluate(x) goes through the decoder in slabs of rows (bounded memory, no tensor near
    2^31 elements; the reference bounds the same pass by halving the batch, cvae.py:1087-1153).  Running-statistics
    BatchNorm makes rows independent, so a run cut into ragged 7-row slabs must equal the one-slab run BIT FOR BIT; in
    train mode (batch statistics over the whole decoder batch) the slab path must not be taken."""
    case = dict(get_case('e2_n8_L3'))
    net = build(case)
    net.eval()
    kw = case['net']
    N, L = 10, net.latent_sampling
    x, y, eps = det_inputs(N, kw['input_shape'], kw['num_labels'], L, kw['latent_dim'], seed=5)
    xd, ed = x.to(DEV), eps.to(DEV)
    with torch.no_grad():
        whole = net.evaluate(xd, epsilon=ed)
    calls = []
    orig = net._decode_rows
    monkeypatch.setattr(net, '_decode_rows', lambda z: (calls.append(z.numel() // z.shape[-1]), orig(z))[1])
    monkeypatch.setenv('JVAE_EVAL_SLAB_ROWS', '7')
    with torch.no_grad():
        slabbed = net.evaluate(xd, epsilon=ed)
    rows = (L + 1) * N
    assert calls == [7] * (rows // 7) + ([rows % 7] if rows % 7 else [])
    assert torch.equal(whole[0], slabbed[0]) and torch.equal(whole[1], slabbed[1])
    for k in whole[2]:
        assert torch.equal(whole[2][k], slabbed[2][k]), k
    del calls[:]
    net.train()
    with torch.no_grad():
        net.evaluate(xd, epsilon=torch.cat([ed[:1], ed[1:2]]))
    assert calls == [2 * N]                     # train mode: one pass over the whole decoder batch
    mb = net.max_batch_sizes
    assert mb['train'] & (mb['train'] - 1) == 0 and mb['test'] & (mb['test'] - 1) == 0
    assert (net._latent_samplings['eval'] + 1) * mb['test'] * 3072 < 2 ** 31 <= (net._latent_samplings['eval'] + 1) * 2 * mb['test'] * 3072 \
        or mb['test'] == 1 << 16


def test_pack_cache_is_transparent():
    """csrc/pack_cache.hip: evaluate() opens a span of constant weights in which ONE launch refreshes the packed operand
    forms of all convolution weights instead of one pack launch per convolution.  It must be invisible: (i) three optimiser
    steps with the cache give bit-identical parameters and losses to the same steps without it, (ii) once the parameters
    have settled in the optimiser's flat buffer (first step) every convolution is served from the cache, (iii) a weight changed behind PyTorch's back (`.data` arithmetic: no version
    counter moves) is seen by the next evaluate(), (iv) outside a span nothing is served from the cache."""
    from jvae_hip import lib as L
    from jvae_hip import ops
    case = full_config(2, 16)
    kw = case['net']
    x, y, eps = (t.to(DEV) for t in det_inputs(16, kw['input_shape'], 10, 1, 64, seed=9))

    def two_steps(cache_mb):
        old = L.PACK_CACHE_BYTES
        L.PACK_CACHE_BYTES = cache_mb << 20
        if not cache_mb:
            L.load().jvae_pack_cache_configure(None, 0)
            L._pack_cache.clear()
        try:
            net = build(case)
            out = []
            for step in range(3):
                losses, _ = net.train_step(x, y, epsilon=eps)
                out.append(losses['total'].detach().clone())
            torch.cuda.synchronize()
            return net, out
        finally:
            L.PACK_CACHE_BYTES = old
    net0, l0 = two_steps(0)
    L._pack_cache.clear()
    before = L.pack_cache_stats()
    net1, l1 = two_steps(64)
    after = L.pack_cache_stats()
    for a, b in zip(l0, l1):
        assert torch.equal(a, b)
    for (k, p), (_, q) in zip(net0.state_dict().items(), net1.state_dict().items()):
        assert torch.equal(p, q), k
    assert after['entries'] >= 12 and after['refreshes'] - before['refreshes'] >= 1
    hits, misses = after['hits'] - before['hits'], after['misses'] - before['misses']
    # step 1 fills the table; the first Adam moves the parameters into the flat buffer (new addresses = new owner: refilled in
    # step 2); step 3 is served from the cache entirely
    assert misses <= 2 * after['entries'] and hits >= after['entries']
    # (iii) a weight changed through .data
    net1.eval()
    with torch.no_grad():
        ref_a = net1.evaluate(x, y, epsilon=eps)[2]['total'].clone()
        net1.imager[3].weight.data.mul_(0.5)
        ref_b = net1.evaluate(x, y, epsilon=eps)[2]['total'].clone()
    assert not torch.equal(ref_a, ref_b)
    net0.eval()
    net0.imager[3].weight.data.mul_(0.5)
    old = L.PACK_CACHE_BYTES
    L.PACK_CACHE_BYTES = 0
    try:
        with torch.no_grad():
            assert torch.equal(net0.evaluate(x, y, epsilon=eps)[2]['total'], ref_b)
    finally:
        L.PACK_CACHE_BYTES = old
    # (iv) a convolution called outside evaluate() packs for itself
    s0 = L.pack_cache_stats()
    net1.imager(torch.randn(4, *net1.imager.input_shape, device=DEV))
    s1 = L.pack_cache_stats()
    assert (s1['hits'], s1['misses']) == (s0['hits'], s0['misses'])


def test_pack_cache_span_closes_without_an_optimizer_step():
    """a train-mode evaluate() leaves the span of constant weights open for the backward that needs it - and only
    for that.  (i) the backward pass itself closes it (no optimiser step needed); (ii) a loss-only call (no backward) followed
    by a weight change through .data: a stand-alone forward() opens a span of its own and sees the NEW weights; (iii) a
    convolution of a tensor the model did not declare - here a temporary weight, inside an open span - is never registered;
    (iv) load_state_dict() closes the span."""
    from jvae_hip import lib as L
    from jvae_hip import ops
    case = full_config(2, 8)
    net = build(case)
    x, y, eps = (t.to(DEV) for t in det_inputs(8, (3, 32, 32), 10, 1, 64, seed=5))
    net.train_step(x, y, epsilon=eps)                     # parameters settle in the flat buffer
    net.train_step(x, y, epsilon=eps)

    def served(f):
        s0 = L.pack_cache_stats()
        out = f()
        s1 = L.pack_cache_stats()
        return out, s1['hits'] - s0['hits'], s1['misses'] - s0['misses'], s1['entries'] - s0['entries']

    z = torch.randn(4, *net.imager.input_shape, device=DEV)
    # (i) evaluate + backward, no step.  A convolution called OUTSIDE forward() / evaluate() - a submodule invoked directly - is
    # never served from the cache, not even between evaluate() and its backward (the weights may have changed through
    # .data since the cache was armed); it disarms the cache, and the backward that follows packs per call and gives the same
    # gradients bit for bit
    def grads(stray):
        net.optimizer.zero_grad()
        losses = net.evaluate(x, y, epsilon=eps, with_beta=True)[2]
        if stray:
            _, hits, misses, _ = served(lambda: net.imager(z))
            assert (hits, misses) == (0, 0)
        losses['total'].mean().backward()
        torch.cuda.synchronize()
        return net.optimizer._groups[0].g.clone()
    g_plain, g_stray = grads(False), grads(True)
    assert torch.equal(g_plain, g_stray)
    _, hits, misses, _ = served(lambda: net.imager(z))
    assert (hits, misses) == (0, 0)
    # ... and sees a weight change made through .data while the cache was still armed
    net.optimizer.zero_grad()
    net.evaluate(x, y, epsilon=eps, with_beta=True)       # train mode, no backward: the cache stays armed
    with torch.no_grad():
        before = net.imager(z).clone()
    net.evaluate(x, y, epsilon=eps, with_beta=True)
    net.imager[3].weight.data.mul_(2.)
    with torch.no_grad():
        after = net.imager(z).clone()
    L.pack_cache_end()
    with torch.no_grad():
        uncached = net.imager(z).clone()
    net.imager[3].weight.data.mul_(0.5)
    assert not torch.equal(before, after) and torch.equal(after, uncached)
    # (ii) loss-only call, then a .data change, then a stand-alone forward()
    net.evaluate(x, y, epsilon=eps, with_beta=True)
    with torch.no_grad():
        a = net(x, y, epsilon=eps)[0].clone()
    net.evaluate(x, y, epsilon=eps, with_beta=True)       # span left open again (training mode, no backward)
    net.imager[3].weight.data.mul_(0.5)
    with torch.no_grad():
        b = net(x, y, epsilon=eps)[0].clone()
    L.pack_cache_end()
    old = L.PACK_CACHE_BYTES
    L.PACK_CACHE_BYTES = 0
    try:
        with torch.no_grad():
            c = net(x, y, epsilon=eps)[0].clone()         # the uncached answer for the changed weights
    finally:
        L.PACK_CACHE_BYTES = old
    assert not torch.equal(a, b) and torch.equal(b, c)
    # (iii) an undeclared weight inside an open span
    net.evaluate(x, y, epsilon=eps, with_beta=True)
    spec = ops.ConvSpec(32, 32, 5, 1, 2, 0, False)
    xt = torch.randn(8, 32, 16, 16, device=DEV)
    wt = torch.randn(32, 32, 5, 5, device=DEV)
    _, hits, misses, entries = served(lambda: ops.conv_fwd_raw(xt, wt, None, spec))
    assert (hits, misses, entries) == (0, 0, 0)
    # (iv) load_state_dict closes the span
    net.load_state_dict(net.state_dict())
    _, hits, misses, _ = served(lambda: net.imager(z))
    assert (hits, misses) == (0, 0)
    torch.cuda.synchronize()


def test_nonfinite_parameters_end_the_run_before_the_next_backward(capsys):
    """cvae.py:2454-2457: the reference scans every parameter for NaN / Inf between evaluate() and backward() and ends the run
    with `print('GRAD NAN'); sys.exit(1)`.  Here the Adam kernel raises a device flag when an updated parameter is not finite;
    train_step() reads the flag of the PREVIOUS update (a 4-byte copy that completed during this step's forward) at the
    same place - after evaluate(), before backward: the step after the poisoned update exits with nothing back-propagated."""
    case = full_config(2, 8)
    net = build(case)
    x, y, eps = (t.to(DEV) for t in det_inputs(8, (3, 32, 32), 10, 1, 64, seed=3))
    net.train_step(x, y, epsilon=eps)                       # a healthy step: no exit on the next one
    net.train_step(x, y, epsilon=eps)
    net.optimizer._lr = float('inf')                        # the next update writes Inf / NaN parameters
    net.train_step(x, y, epsilon=eps)
    with pytest.raises(SystemExit) as ei:
        net.train_step(x, y, epsilon=eps)
    assert ei.value.code == 1 and 'GRAD NAN' in capsys.readouterr().out
    torch.cuda.synchronize()
    assert float(net.optimizer._groups[0].g.abs().sum()) == 0.       # zero_grad ran, backward did not
    # NaN / Inf present BEFORE the first step (a poisoned checkpoint, an initialisation gone wrong): the reference's scan
    # sees it on the very first step (cvae.py:2454-2457 looks at every parameter); so does the one-off device scan of
    # check_nonfinite() that follows construction, load_state_dict() and .to()
    fresh = build(case)
    with torch.no_grad():
        fresh.imager[3].weight[0, 0, 0, 0] = float('inf')
    with pytest.raises(SystemExit):
        fresh.train_step(x, y, epsilon=eps)
    assert 'GRAD NAN' in capsys.readouterr().out
    healthy = build(case)
    healthy.train_step(x, y, epsilon=eps)
    sd = {k: v.clone() for k, v in healthy.state_dict().items()}
    sd['encoder.dense_mean.weight'][0, 0] = float('nan')
    healthy.load_state_dict(sd)
    with pytest.raises(SystemExit):
        healthy.train_step(x, y, epsilon=eps)
    # a hand-written loop that never asks pays no flag copy
    quiet = build(case)
    quiet.optimizer.zero_grad()
    quiet.evaluate(x, y, epsilon=eps, with_beta=True)[2]['total'].mean().backward()
    quiet.optimizer.clip(quiet.parameters())
    quiet.optimizer.step()
    assert getattr(quiet.optimizer, '_flag_event', None) is None and not quiet.optimizer._flag_wanted
    assert quiet.optimizer.check_nonfinite() is False and quiet.optimizer._flag_wanted


def test_accuracy_loop_records_and_recovers(tmp_path):
    """accuracy() (cvae.py:1187-1452) over a synthetic test set: per-method accuracies, `testing` bookkeeping, the
    per-sample losses recorded into `record-<set>.pth` in the reference's format (the file layout itself is
    pinned on the CPU against a reference-written file), and a second pass RECOVERED from that file giving the same numbers."""
    from cvae import ClassificationVariationalNetwork as Net
    from jvae_compat.recorders import LossRecorder
    kw = dict(get_case('e2_n8_L3')['net'])
    net = Net(**kw)
    load_det_state(net, seed=0)
    net.to(DEV)
    torch.manual_seed(3)
    data = torch.utils.data.TensorDataset(torch.rand(70, 3, 32, 32), torch.randint(0, 10, (70,)))
    data.name = 'synth'
    rec = LossRecorder(32)
    torch.manual_seed(5)
    acc = net.accuracy(data, batch_size=32, recorder=rec, sample_dirs=[str(tmp_path)])
    assert set(acc) == set(net.predict_methods) and all(0. <= a <= 1. for a in acc.values())
    assert net.testing[0]['iws']['n'] == 70 and net.testing[0]['iws']['accuracy'] == acc['iws']
    path = os.path.join(tmp_path, 'record-synth.pth')
    d = torch.load(path, weights_only=False)
    assert {'batch_size', 'last_batch_size', '_num_batch', '_samples', '_recorded_batches', '_tensors', '_seed', 'device'} == set(d)
    assert d['_recorded_batches'] == 3 and d['last_batch_size'] == 6 and d['batch_size'] == 32
    assert tuple(d['_tensors']['iws'].shape) == (10, 70) and tuple(d['_tensors']['cross_x'].shape) == (70,)
    assert tuple(d['_tensors']['logits'].shape) == (10, 70) and torch.equal(d['_tensors']['y_true'].cpu(), data.tensors[1])
    again = LossRecorder.load(path, device=DEV)
    acc2 = net.accuracy(data, batch_size=32, recorder=again)               # recovered from the record: nothing is evaluated
    assert acc2 == acc
    assert isinstance(net.accuracy(data, batch_size=32, method='closest'), float)


def test_batchnorm_as_its_own_pass_matches_the_deferred_form(monkeypatch):
    """HipConvStack.defer_batchnorm (JVAE_DEFER_BN=0 in the environment): BatchNorm + activation as their own kernels instead of
    inside the next convolution's staging and the previous one's epilogue.  Same arithmetic per element (one fmaf, the same
    batch statistics up to the order of their partial sums): losses within 1e-5, every gradient within 2e-4 of its norm - at a
    ragged batch, for ReLU and for the leaky ReLU."""
    from module.vae_layers.conv import HipConvStack
    for act in ('relu', 'leaky'):
        case = full_config(2, 37)
        case['net'] = dict(case['net'], activation=act)
        x, y, eps = (t.to(DEV) for t in det_inputs(37, (3, 32, 32), 10, 1, 64, seed=11))
        out = {}
        for defer in (True, False):
            monkeypatch.setattr(HipConvStack, 'defer_batchnorm', defer)
            net = build(case)
            net.optimizer.zero_grad()
            losses = net.evaluate(x, y, epsilon=eps, with_beta=True)[2]
            losses['total'].mean().backward()
            torch.cuda.synchronize()
            out[defer] = ({k: v.detach().clone() for k, v in losses.items()},
                          {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None})
        for k in ('total', 'cross_x', 'kl'):
            assert rel(out[False][0][k], out[True][0][k]) < 1e-5, (act, k)
        assert out[False][1].keys() == out[True][1].keys()
        for n, g in out[True][1].items():
            d = float((out[False][1][n] - g).norm()) / max(float(g.norm()), 1e-6 * float(torch.cat([t.flatten() for t in out[True][1].values()]).norm()))
            assert d < 2e-4, (act, n, d)
