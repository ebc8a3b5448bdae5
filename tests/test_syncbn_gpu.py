"""Data-parallel training on the GPU with SYNCHRONISED BatchNorm: 2 ranks (gloo, sharing the one GPU of
the test box), each with half of a global batch, must reproduce the single-process step on the whole batch — the thing
the single-process reference computes: per-sample ELBO terms of each half, and the parameters after clip + Adam."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_RANK, WORLD = 6, 2


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
