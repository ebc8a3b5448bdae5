"""Data-parallel path on CPU (gloo, world_size 2): the optimiser's flat gradient buffer is averaged across ranks by
one all-reduce per flat group and every rank ends with identical gradients.  The HIP kernels
themselves need a GPU; what is covered here is the exchange logic that bench.py --gpus N relies on."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    for p in (REPO, os.path.join(REPO, 'joint-vae_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from module.optimizers import Optimizer
    torch.manual_seed(0)                                  # identical replicas
    params = [torch.nn.Parameter(torch.randn(s)) for s in ((7, 5), (3,), (2, 3, 5, 5), (1,))]
    frozen = torch.nn.Parameter(torch.randn(4))           # never receives a gradient: must stay out of the exchange
    opt = Optimizer(params + [frozen], optim_type='adam', lr=1e-3, weight_decay=3e-5, grad_clipping=100)
    opt.set_distributed(world)
    g = torch.Generator().manual_seed(100 + rank)         # different data per rank
    local = [torch.randn(p.shape, generator=g) for p in params]
    for p, gr in zip(params, local):
        p.grad = gr.clone()
    opt.reduce_gradients()
    assert frozen.grad is None
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert all(torch.equal(gathered[0], t) for t in gathered)
    # expected: mean over ranks of the local gradients
    exp = []
    for i, p in enumerate(params):
        acc = torch.zeros(p.shape)
        for r in range(world):
            gg = torch.Generator().manual_seed(100 + r)
            acc += [torch.randn(q.shape, generator=gg) for q in params][i]
        exp.append(acc / world)
    for p, e in zip(params, exp):
        assert torch.allclose(p.grad, e, atol=1e-6)
    # gradients live in ONE flat, 16-byte aligned buffer (one all-reduce)
    assert len(opt._groups) == 1 and opt._groups[0].g.numel() % 4 == 0
    base = opt._groups[0].g.data_ptr()
    assert all((p.grad.data_ptr() - base) % 16 == 0 for p in params)
    # a second reduce in the same step is a no-op; zero_grad re-arms it and keeps the views
    before = flat.clone()
    opt.reduce_gradients()
    assert torch.equal(torch.cat([p.grad.reshape(-1) for p in params]), before)
    opt.zero_grad()
    assert all(float(p.grad.abs().max()) == 0. for p in params) and params[0].grad.data_ptr() == base
    # second "step": an early bucket (params 1..2) is exchanged asynchronously first, the rest afterwards
    opt.set_early_bucket(params[1:3])
    for p, gr in zip(params, local):
        p.grad.copy_(gr)
    opt.reduce_early_bucket()
    assert opt._early_work is not None and opt._early_range is not None
    opt.reduce_gradients()
    for p, e in zip(params, exp):
        assert torch.allclose(p.grad, e, atol=1e-6)
    open(os.path.join(out_dir, f'ok{rank}'), 'w').close()
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2(tmp_path):
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert sorted(os.listdir(tmp_path)) == ['ok0', 'ok1']
