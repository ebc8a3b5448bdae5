"""GPU box: the data-parallel step over the REAL RCCL backend with a one-rank process group.

A one-GPU box cannot host two RCCL ranks, so this drives every RCCL call of the N>1 path (communicator creation with
device_id, the asynchronous early-bucket all-reduce issued on the side stream from the hook on z, the two remaining
slices in reduce_gradients(), barrier, the MAX reduction of bench.py) on a world of one, where AVG is the identity:
the losses and parameters must then be bit-identical to the non-distributed step.  Optional: --sync-bn.
--graph: the captured form of the step instead (graph_train_step: two HIP graphs with ONE eager RCCL all-reduce between them -
what `bench.py --gpus N` runs by default since round 5), captured and replayed with the RCCL communicator and its watchdog alive.
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402


def run(distributed, sync_bn, steps=6, bs=128, graph=False):
    dev = torch.device('cuda', 0)
    net = bench.build_model(dev, 2)
    if distributed:
        net.optimizer.set_distributed(2)         # flag only: the group below has one rank, AVG over it is the identity
        if sync_bn:
            net.set_sync_batchnorm(2)
    g = torch.Generator(device=dev).manual_seed(1234)
    x = torch.rand(bs, 3, 32, 32, device=dev, generator=g)
    y = torch.randint(0, 10, (bs,), device=dev, generator=g)
    torch.manual_seed(7)
    torch.cuda.manual_seed(7)
    meas, tot = None, []
    replay = net.graph_train_step(x, y) if graph else None
    torch.manual_seed(7)
    torch.cuda.manual_seed(7)
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(steps):
        if graph:
            losses, meas = replay(x, y)
        else:
            losses, meas = net.train_step(x, y, batch=i, current_measures=meas)
        tot.append(losses['total'].detach().clone())
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    dt = (time.time() - t0) / steps
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    return torch.stack(tot), flat, dt


def main():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    sync_bn = '--sync-bn' in sys.argv
    graph = '--graph' in sys.argv
    t_ref, p_ref, dt_ref = run(False, False, graph=graph)
    t_dp, p_dp, dt_dp = run(True, sync_bn, graph=graph)
    t = torch.tensor([dt_dp], device='cuda', dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok_l = torch.equal(t_ref, t_dp)
    ok_p = torch.equal(p_ref, p_dp)
    rel = float((t_ref - t_dp).abs().max() / t_ref.abs().max())
    print(f'rccl one-rank rehearsal: graph={graph} sync_bn={sync_bn} losses bit-identical={ok_l} (max rel diff {rel:.2e}) '
          f'params bit-identical={ok_p}  ms/step plain={dt_ref * 1e3:.2f} dp={float(t) * 1e3:.2f}')
    dist.barrier()
    dist.destroy_process_group()
    # --sync-bn: the layers divide the all-reduced sums by world * N, so with the pretended world of 2 the values differ
    # by construction; that leg only proves that the (C, 2) all-reduces per layer and direction run over RCCL
    sys.exit(0 if (ok_l and ok_p) or (sync_bn and bool(torch.isfinite(t_dp).all())) else 1)


if __name__ == '__main__':
    main()
