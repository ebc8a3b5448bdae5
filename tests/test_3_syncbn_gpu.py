"""Data-parallel training on the GPU with SYNCHRONISED BatchNorm: 2 ranks (gloo, sharing the one GPU of
the test box), each with half of a global batch, must reproduce the single-process step on the whole batch — the thing
the single-process reference computes: per-sample ELBO terms of each half, and the parameters after clip + Adam."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_RANK, WORLD = 49, 2         # 49 per rank: BatchNorm-sums launch plans with EMPTY trailing image parts (ragged batches)


def _paths():
    for p in (REPO, os.path.join(REPO, 'joint-vae_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)


def _build():
    from cvae import ClassificationVariationalNetwork as Net
    from oracle.cases import get_case
    from oracle.det_init import load_det_state
    kw = get_case('c2_n8')['net']
    net = Net(**kw)
    load_det_state(net, seed=0)
    net.to('cuda')
    net.train()
    return net, kw


def _worker(rank, port, out_dir):
    _paths()
    import torch.distributed as dist
    from oracle.det_init import det_inputs
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    torch.cuda.set_device(0)
    net, kw = _build()
    net.optimizer.set_distributed(WORLD)
    net.set_sync_batchnorm(WORLD)
    x, y, eps = det_inputs(N_RANK * WORLD, kw['input_shape'], 10, 1, 64, seed=31)
    sl = slice(rank * N_RANK, (rank + 1) * N_RANK)
    losses, _ = net.train_step(x[sl].cuda(), y[sl].cuda(), epsilon=eps[:, sl].cuda())
    torch.save({'losses': {k: v.detach().cpu() for k, v in losses.items()},
                'params': {k: v.detach().cpu() for k, v in net.state_dict().items()},
                'gnorm': float(net.optimizer.grad_norm())}, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_sync_batchnorm_matches_single_process(tmp_path):
    _paths()
    from oracle.det_init import det_inputs
    port = 29700 + os.getpid() % 1000
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    net, kw = _build()
    x, y, eps = det_inputs(N_RANK * WORLD, kw['input_shape'], 10, 1, 64, seed=31)
    full, _ = net.train_step(x.cuda(), y.cuda(), epsilon=eps.cuda())
    ranks = [torch.load(os.path.join(tmp_path, f'rank{r}.pt')) for r in range(WORLD)]

    def rel(a, b):
        return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
    for r in range(WORLD):
        sl = slice(r * N_RANK, (r + 1) * N_RANK)
        for k in ('total', 'cross_x', 'kl', 'wmse', 'zdist'):
            assert rel(ranks[r]['losses'][k], full[k].detach().cpu()[sl]) < 1e-4, (r, k)
    gn = float(net.optimizer.grad_norm())
    assert abs(ranks[0]['gnorm'] - gn) < 2e-4 * gn and abs(ranks[1]['gnorm'] - gn) < 2e-4 * gn
    mine = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    for k, v in mine.items():
        if not v.dtype.is_floating_point:
            assert torch.equal(v, ranks[0]['params'][k]), k
            continue
        assert torch.equal(ranks[0]['params'][k], ranks[1]['params'][k]) or rel(ranks[0]['params'][k], ranks[1]['params'][k]) < 1e-6, k
        d = float((ranks[0]['params'][k].double() - v.double()).norm() / v.double().norm().clamp_min(1e-12))
        assert d < 2e-2, (k, d)          # Adam's first step is ~lr*sign(g): only noise-level gradients may differ
    running = [k for k in mine if k.endswith('running_var')]
    for k in running:                    # the statistics themselves: global-batch values on every rank
        assert rel(ranks[0]['params'][k], mine[k]) < 1e-5, k


def _graph_worker(rank, port, out_dir):
    """2 ranks (gloo, one GPU): replicas built from DIFFERENT seeds are brought together by set_distributed's broadcast;
    the data-parallel step then runs as two captured HIP graphs with one all-reduce between them."""
    _paths()
    import torch.distributed as dist
    from cvae import ClassificationVariationalNetwork as Net
    from oracle.cases import get_case
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    torch.cuda.set_device(0)
    torch.manual_seed(100 + rank)                       # different initial weights on purpose
    kw = get_case('c2_n8')['net']
    net = Net(**kw).to('cuda')
    net.train()
    net.set_distributed(WORLD)
    g = torch.Generator().manual_seed(50 + rank)        # each rank its own shard
    x = torch.rand(16, 3, 32, 32, generator=g).cuda()
    y = torch.randint(0, 10, (16,), generator=g).cuda()
    # the early (decoder) bucket really is exchanged while the encoder's backward is still to come.  Every
    # collective and every dgrad launch of this eager step is logged in host order with a HIP event on the stream it is issued on.
    from jvae_hip import ops as _ops
    order = []
    real_ar, real_dg = dist.all_reduce, _ops.conv_dgrad_raw

    def logged_all_reduce(t, *a, **k):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        order.append(('allreduce', int(t.numel()), bool(k.get('async_op', False)), ev))
        return real_ar(t, *a, **k)

    def logged_dgrad(gy, w, spec, shape):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        order.append(('dgrad', (spec.cin, spec.cout, bool(spec.transposed)), False, ev))
        return real_dg(gy, w, spec, shape)
    net.train_step(x, y)                                 # the first step moves the parameters into the flat buffer; from the
    torch.cuda.synchronize()                             # second on the decoder's slice of it can leave early
    dist.all_reduce, _ops.conv_dgrad_raw = logged_all_reduce, logged_dgrad
    try:
        eager_losses, _ = net.train_step(x, y)           # one eager data-parallel step (early bucket + remainder)
    finally:
        dist.all_reduce, _ops.conv_dgrad_raw = real_ar, real_dg
    torch.cuda.synchronize()
    kinds = [o[0] for o in order]
    i_ar = kinds.index('allreduce')
    from module.vae_layers.conv import HipConv2d, HipConvTranspose2d
    n_dec = sum(isinstance(m, (HipConv2d, HipConvTranspose2d)) for m in net.imager.modules())
    n_enc = sum(isinstance(m, (HipConv2d, HipConvTranspose2d)) for m in net.features.modules())
    dec_elems = sum(p.numel() for m in (net.imager, net.decoder) for p in m.parameters())
    first_enc = next(o for o in order[i_ar + 1:] if o[0] == 'dgrad')
    overlap = {'n_allreduce': kinds.count('allreduce'), 'first_is_async': order[i_ar][2], 'early_elems': order[i_ar][1],
               'dec_elems': dec_elems, 'dgrads_before': kinds[:i_ar].count('dgrad'), 'dgrads_after': kinds[i_ar:].count('dgrad'),
               'n_dec': n_dec, 'n_enc': n_enc,
               # device timeline (reported, not asserted: at 16 images per rank it is launch-bound): the point at which the
               # bucket's gradients are complete on the collective's stream - the upsampler's weight gradients run there -
               # relative to the start of the encoder's first dgrad on the main stream
               'ms_bucket_ready_to_first_encoder_dgrad': order[i_ar][3].elapsed_time(first_enc[3]),
               'later_allreduces_async': [o[2] for o in order[i_ar + 1:] if o[0] == 'allreduce']}
    first = float(eager_losses['total'].detach().mean())
    del eager_losses                                    # frees the eager autograd graph (its AccumulateGrad nodes are bound to the
    torch.cuda.synchronize()                            # default stream and must not survive into the capture)
    step = net.graph_train_step(x, y, warmup=1)
    assert isinstance(step.graph, tuple) and len(step.graph) == 2
    for _ in range(6):
        losses, meas = step(x, y)
    torch.cuda.synchronize()
    last = float(losses['total'].detach().mean())
    # an EAGER data-parallel step AFTER graph_train_step() must exchange gradients again (the capture-only
    # `_external_reduce` flag used to stay set: no join, no all-reduce, replicas silently diverging on their own shards)
    assert not getattr(net.optimizer, '_external_reduce', False)
    late, _ = net.train_step(x, y)
    torch.cuda.synchronize()
    del late
    eps_probe = torch.randn(4, device='cuda')           # the device generator was offset per rank: ranks draw different noise
    torch.save({'params': {k: v.detach().cpu() for k, v in net.state_dict().items()}, 'first': first, 'last': last,
                'opt_step': net.optimizer._groups[0].step, 'eps_probe': eps_probe.cpu(), 'rmse': meas['rmse'],
                'overlap': overlap},
               os.path.join(out_dir, f'g{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_graph_captured_data_parallel_step(tmp_path):
    """graph_train_step() under data parallelism - [zero_grad, forward, backward] and [clip, Adam] as two
    HIP graphs with the flat-gradient all-reduce between them.  Replicas that started from different weights are equal
    after set_distributed() and STAY bit-identical through eager, graph and again eager steps (parameters, BatchNorm running statistics
    excepted: those are per-rank by design), the loss falls, the optimiser's step count follows the replays."""
    port = 29900 + os.getpid() % 1000
    mp.spawn(_graph_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    a, b = (torch.load(os.path.join(tmp_path, f'g{r}.pt')) for r in range(WORLD))
    for k, v in a['params'].items():
        if 'running_' in k or 'num_batches' in k:
            continue
        assert torch.equal(v, b['params'][k]), k
    assert a['opt_step'] == b['opt_step'] == 2 + 1 + 6 + 1    # two eager steps + warm-up + replays + eager step after the graph
    assert a['last'] < a['first'] and b['last'] < b['first']
    assert not torch.equal(a['eps_probe'], b['eps_probe'])
    assert 0 < a['rmse'] < 10
    # the eager data-parallel step: exactly two collectives - the decoder-side bucket, ASYNCHRONOUS, issued when every
    # dgrad of the upsampler has been queued and none of the encoder's (HOST order: the enqueue order of the main stream), and the contiguous remainder after backward
    for r in (a, b):
        ov = r['overlap']
        assert ov['n_allreduce'] == 2 and ov['first_is_async'] and ov['later_allreduces_async'] == [False], ov
        assert ov['dec_elems'] <= ov['early_elems'] <= ov['dec_elems'] + 64, ov      # + the 16-byte alignment gaps of the flat buffer
        assert ov['dgrads_before'] == ov['n_dec'] and ov['dgrads_after'] == ov['n_enc'] - 1, ov    # the first layer has no dgrad
        assert abs(ov['ms_bucket_ready_to_first_encoder_dgrad']) < 1e3, ov      # the events exist and resolve; sign depends on the box


def _b8_worker(rank, port, out_dir):
    _paths()
    import torch.distributed as dist
    from jvae_hip import ops_b8
    from oracle.det_init import det_inputs
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    torch.cuda.set_device(0)
    net, kw = _build()
    net.set_compute_dtype('bf16')
    net.optimizer.set_distributed(WORLD)
    net.set_sync_batchnorm(WORLD)
    calls = [0]
    orig = ops_b8.sync_batchnorm_act

    def counted(*a, **k):
        calls[0] += 1
        return orig(*a, **k)
    ops_b8.sync_batchnorm_act = counted
    x, y, eps = det_inputs(N_RANK * WORLD, kw['input_shape'], 10, 1, 64, seed=31)
    sl = slice(rank * N_RANK, (rank + 1) * N_RANK)
    losses, _ = net.train_step(x[sl].cuda(), y[sl].cuda(), epsilon=eps[:, sl].cuda())
    torch.save({'losses': {k: v.detach().cpu() for k, v in losses.items()},
                'params': {k: v.detach().cpu() for k, v in net.state_dict().items()},
                'gnorm': float(net.optimizer.grad_norm()), 'b8_sync_calls': calls[0]}, os.path.join(out_dir, f'b{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_sync_batchnorm_bf16_layout_matches_single_process(tmp_path):
    """The bf16 (B8) mode with synchronised BatchNorm runs on its own kernels (jvae_bn_sums_b8 / fwd_sync / bwd_sums /
    bwd_sync: no fp32 fall-back between two conversions) and reproduces the single-process bf16 step on the whole batch.
    Tolerances are the bf16 mode's: the two runs round the same activations to bf16 from statistics
    that agree to fp32 rounding, so single bf16 ulps (2^-8) flip."""
    _paths()
    from oracle.det_init import det_inputs
    port = 29800 + os.getpid() % 1000
    mp.spawn(_b8_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    net, kw = _build()
    net.set_compute_dtype('bf16')
    x, y, eps = det_inputs(N_RANK * WORLD, kw['input_shape'], 10, 1, 64, seed=31)
    full, _ = net.train_step(x.cuda(), y.cuda(), epsilon=eps.cuda())
    ranks = [torch.load(os.path.join(tmp_path, f'b{r}.pt')) for r in range(WORLD)]
    assert ranks[0]['b8_sync_calls'] >= 8 and ranks[1]['b8_sync_calls'] == ranks[0]['b8_sync_calls']

    def rel(a, b):
        return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
    for r in range(WORLD):
        sl = slice(r * N_RANK, (r + 1) * N_RANK)
        for k, tol in (('total', 5e-3), ('cross_x', 5e-3), ('kl', 2e-2)):
            assert rel(ranks[r]['losses'][k], full[k].detach().cpu()[sl]) < tol, (r, k)
    gn = float(net.optimizer.grad_norm())
    assert abs(ranks[0]['gnorm'] - gn) < 2e-2 * gn and abs(ranks[1]['gnorm'] - gn) < 2e-2 * gn
    mine = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    for k in mine:
        if k.endswith('running_var') or k.endswith('running_mean'):      # global-batch statistics on every rank
            assert torch.equal(ranks[0]['params'][k], ranks[1]['params'][k]), k
            assert rel(ranks[0]['params'][k], mine[k]) < 2e-2, k
    for k, v in ranks[0]['params'].items():                              # replicas stay identical
        if v.dtype.is_floating_point:
            assert torch.equal(v, ranks[1]['params'][k]), k


@pytest.mark.timeout(600)
@pytest.mark.parametrize('flags', [['--eager'], [], ['--eager', '--sync-bn']], ids=['eager', 'graph', 'sync-bn'])
def test_bench_two_ranks_as_the_driver_launches_it(flags, tmp_path):
    """bench.py at N = 2 exactly as the driver starts it - `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2
    --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...` in fresh child processes (the launcher runs before anything
    touches the GPU) - with the gloo backend standing in for RCCL (JVAE_BENCH_BACKEND=gloo: both ranks share this box's one
    GPU; the collective calls, streams and the two-graph step are the ones RCCL will see).  Eager, captured-graph and
    synchronised-BatchNorm steps: one well-formed JSON line from rank 0, whole-job throughput, and replicas whose parameters
    are BIT-identical after the steps although every rank trained on its own shard."""
    import json
    import subprocess
    port = 29500 + (os.getpid() + len(flags) * 7 + (17 if '--sync-bn' in flags else 0)) % 400
    env = dict(os.environ, JVAE_BENCH_BACKEND='gloo', JVAE_BENCH_NO_PROBES='1', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '2',
           '--no-cpu-baseline'] + flags
    res = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=540)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['metric'] == 'training_images_per_sec' and d['n_gpus'] == 2 and d['steps'] == 3 and d['scaling'] == 'weak'
    assert d['config']['global_batch'] == 1024 and d['config']['parallelism'] == 'dp2' and d['config']['backend'] == 'gloo'
    assert d['config']['launch'].startswith('eager' if '--eager' in flags else 'HIP graph replay')
    assert d['config']['bn_statistics'] == ('synchronised over ranks' if '--sync-bn' in flags else 'per-rank (local)')
    assert d['value'] > 0 and abs(d['value'] - 1024 * 3 / (d['ms_per_step'] * 3e-3)) < 1e-6 * d['value']
    assert d['replicas_identical'] is True and len(d['replica_param_checksums']) == 2
    assert np.isfinite(d['final_loss'])
