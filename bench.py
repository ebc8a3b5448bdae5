#!/usr/bin/env python3
"""Benchmark of the per-batch joint-CVAE training step on MI355X (BASELINE.json metric: training images/s).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = cvae.py:2429-2461 of the reference on one synthetic batch already resident in HBM:
zero_grad -> evaluate(x, y, with_beta=True) -> total.mean().backward() -> clip_grad_norm_(100) -> Adam, with the
reparameterisation noise drawn on the device.  Workload = BASELINE.json configs[1] (CIFAR-10 conv CVAE, bs=512
per GPU, fp32).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, 'joint-vae_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

FLOP_PER_IMAGE = 923.2e6          # conv/linear MACs*2, fwd + bwd, L=1 (decoder runs on 2N latents)
MFMA_F32_PEAK = 157.3e12          # MI355X dense fp32 MFMA, MI355X_MICROARCH.md
MFMA_BF16_PEAK = 2.5e15           # dense bf16 MFMA (same guide)
X3_PRODUCTS = 6                   # bf16 MFMA products per fp32 product in conv_x3.hip (3-way exact operand split)
BATCH_PER_GPU = 512


WORKLOADS = {   # id -> (description, FLOP per image fwd+bwd)   (BASELINE.json configs; the metric is quoted on 2)
    2: ('BASELINE configs[1]: CIFAR-10 3x32x32 conv CVAE (conv32/deconv32, latent_dim=64, C=10, batch_norm=both, '
        'learned sigma, L=1), bs=512 per GPU, fp32, fwd+bwd+clip+Adam', 923.2e6),
    3: ('BASELINE configs[2]: CIFAR-100 3x32x32 conv CVAE, class-conditional gaussian prior C=100, bs=512 per GPU, fp32',
        923.2e6),
    5: ('BASELINE configs[4]: ImageNet-20-shaped 3x64x64 conv CVAE (conv32+/deconv32+, latent_dim=200, C=20), '
        'bs=256 per GPU; --dtype bf16 = bf16 activations + bf16 MFMA from fp32 master weights, f32 = same geometry in fp32',
        5105e6),
}


_ADAM = dict(optim_type='adam', lr=1e-3, weight_decay=3e-5, grad_clipping=100)


def model_kwargs(workload):
    """Constructor arguments of the BASELINE.json configurations (reference `config.ini:137-169`)."""
    C, K, shape, feat, ups = {2: (10, 64, (3, 32, 32), 'conv32', 'deconv32'),
                              3: (100, 64, (3, 32, 32), 'conv32', 'deconv32'),
                              5: (20, 200, (3, 64, 64), 'conv32+', 'deconv32+')}[workload]
    return dict(input_shape=shape, num_labels=C, type='cvae', features=feat, upsampler=ups, encoder=[], decoder=[],
                classifier=[], batch_norm='both', latent_dim=K, latent_sampling=1, test_latent_sampling=1,
                sigma={'value': 1.0, 'learned': True}, gamma=0, beta=1., output_activation='linear',
                prior=dict(distribution='gaussian', init_mean=0., learned_means=True, var_dim='scalar', freeze_means=0),
                optimizer=dict(_ADAM))


def build_model(device, workload=2):
    from cvae import ClassificationVariationalNetwork as Net
    torch.manual_seed(0)
    net = Net(**model_kwargs(workload))
    net.to(device)
    net.train()
    return net


def dominant_kernel_roofline(device, reps=20):
    """Largest layer of the step (imager.15: ConvTranspose2d 32->32 5x5 on 1024x32x32x32, 53.69 GFLOP fwd) launched
    exactly as the step launches it (deferred BatchNorm+ReLU of its input applied while staging, BatchNorm partial sums
    of its output in the epilogue; the 5 us weight re-pack kernel in front of it is inside the timed region), timed with
    HIP events on the launch stream.

    The layer runs on conv5_x3_kernel (conv_x3.hip): fp32 in / fp32 out, every operand split exactly into three bf16
    terms and each fp32 product accumulated from 6 bf16 MFMA products.  `achieved` counts the ALGORITHMIC fp32 FLOPs;
    `peak` is what the matrix pipes allow for that arithmetic: dense bf16 MFMA peak / 6.  `vs_f32_mfma_peak` relates the
    same rate to the fp32-MFMA peak the north-star target is phrased in (the native fp32 MFMA kernel conv5_fwd_kernel
    reaches 0.69-0.81 of it on this layer; JVAE_X3=0 selects it)."""
    from jvae_hip import ops
    N, C, H = 2 * BATCH_PER_GPU, 32, 32
    spec = ops.ConvSpec(C, C, 5, 1, 2, 0, transposed=True)
    x = torch.randn(N, C, H, H, device=device)
    w = torch.randn(C, C, 5, 5, device=device) * 0.03
    b = torch.zeros(C, device=device)
    aff = (torch.rand(C, device=device) + 0.5, torch.randn(C, device=device) * 0.1, True)
    for _ in range(3):
        ops.conv_fwd_aff_raw(x, w, b, spec, aff, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv_fwd_aff_raw(x, w, b, spec, aff, True)
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / reps
    flops = 2.0 * N * H * H * C * C * 25
    x3 = os.environ.get('JVAE_X3', '1') != '0'
    traffic = None
    pmc = os.path.join(REPO, 'profiles', 'r01_x3_kernel_pmc.json' if x3 else 'r01_dominant_kernel_pmc.json')
    if os.path.exists(pmc):           # HBM bytes per launch from the separate rocprofv3 --pmc passes
        traffic = json.load(open(pmc)).get('hbm_bytes_per_launch')
    if not x3:
        return {'bound': 'mfma', 'kernel': 'conv5_fwd_kernel<1,32,4,1,8,aff>: imager.15 forward (ConvT 32->32 5x5 s1, 1024x32x32x32)',
                'achieved': flops / sec / 1e12, 'peak': MFMA_F32_PEAK / 1e12, 'unit': 'TFLOP/s',
                'frac': flops / sec / MFMA_F32_PEAK, 'traffic': traffic, 'launch_ms': sec * 1e3}
    peak = MFMA_BF16_PEAK / X3_PRODUCTS
    return {'bound': 'mfma', 'kernel': 'conv5_x3_kernel<1,32,2,aff>: imager.15 forward (ConvT 32->32 5x5 s1, 1024x32x32x32), fp32 operands '
                                       'split exactly into 3 bf16 terms, 6 v_mfma_f32_32x32x16_bf16 per fp32 product tile',
            'achieved': flops / sec / 1e12, 'peak': peak / 1e12, 'unit': 'TFLOP/s', 'frac': flops / sec / peak,
            'traffic': traffic, 'launch_ms': sec * 1e3,
            'peak_definition': 'dense bf16 MFMA 2500 TFLOP/s / 6 bf16 products per fp32 product',
            'bf16_mfma_achieved': X3_PRODUCTS * flops / sec / 1e12, 'vs_f32_mfma_peak': flops / sec / MFMA_F32_PEAK}


def cpu_baseline(max_seconds=25.0):
    """The CPU oracle (PyTorch-CPU restatement of the reference step, pinned to the reference's goldens) on the
    host cores, bounded sample of the same workload."""
    from oracle import jvae_oracle as O
    from oracle.cases import full_config
    from oracle.det_init import det_inputs
    kw = full_config(2, BATCH_PER_GPU)['net']
    sp = O.make_spec(**kw)
    P = O.init_state(sp, seed=0)
    opt = O.AdamState(sp)
    x, y, eps = det_inputs(BATCH_PER_GPU, kw['input_shape'], 10, 1, 64, seed=1234)
    O.train_step(sp, P, opt, x, y, eps)                       # warm-up
    t0 = time.time()
    n = 0
    while n < 12 and time.time() - t0 < max_seconds:
        O.train_step(sp, P, opt, x, y, eps)
        n += 1
    dt = time.time() - t0
    return {'value': n * BATCH_PER_GPU / dt, 'unit': 'images/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{n} train steps of bs={BATCH_PER_GPU} (config 2) with the PyTorch-CPU oracle, '
                      f'{os.cpu_count()} logical CPUs visible'}


def _watchdog(seconds):
    """Abort (exit code 3, stacks dumped) instead of hanging the box if the process makes no progress for `seconds`."""
    import faulthandler
    faulthandler.dump_traceback_later(seconds, exit=True)
    return faulthandler


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--batch', type=int, default=None, help='diagnostics only: per-GPU batch (the metric is defined at 512)')
    ap.add_argument('--sync-bn', action='store_true', help='BatchNorm statistics over all ranks (default: per rank)')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'], help='bf16: the mixed-precision mode of BASELINE configs[4]')
    ap.add_argument('--workload', type=int, default=2, choices=sorted(WORKLOADS), help='diagnostics: other BASELINE configs')
    a = ap.parse_args()
    fh = _watchdog(900)
    if a.batch is None:
        a.batch = 256 if a.workload == 5 else BATCH_PER_GPU
    side, ncls = (64, 20) if a.workload == 5 else (32, 100 if a.workload == 3 else 10)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        ndev = torch.cuda.device_count()
        backend = os.environ.get('JVAE_BENCH_BACKEND', 'nccl')       # 'gloo': rehearsal of the DP path on one GPU
        local = local % max(ndev, 1)
        torch.cuda.set_device(local)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)

    net = build_model(device, a.workload)
    if a.dtype == 'bf16':
        net.set_compute_dtype('bf16')
    if world > 1:
        net.optimizer.set_distributed(world)
        if a.sync_bn:
            net.set_sync_batchnorm(world)
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    x = torch.rand(a.batch, 3, side, side, device=device, generator=g)
    y = torch.randint(0, ncls, (a.batch,), device=device, generator=g)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    meas = None
    for i in range(a.warmup):
        _, meas = net.train_step(x, y, batch=i, current_measures=meas)
    sync()
    t0 = time.time()
    for i in range(a.steps):
        losses, meas = net.train_step(x, y, batch=i, current_measures=meas)
    sync()
    dt = time.time() - t0
    if dist is not None:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    value = world * a.batch * a.steps / dt

    if rank == 0:
        out = {'metric': 'training_images_per_sec', 'value': value, 'unit': 'images/s', 'n_gpus': world,
               'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': dt / a.steps * 1e3, 'higher_is_better': True,
               'scaling': 'weak', 'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
               'config': {'workload': WORKLOADS[a.workload][0],
                          'global_batch': world * a.batch, 'parallelism': f'dp{world}',
                          'bn_statistics': 'synchronised over ranks' if (a.sync_bn and world > 1) else 'per-rank (local)',
                          'arithmetic': ('fp32 operands, products and accumulation everywhere; the stride-1 and 4-phase 5x5 layers '
                                         'accumulate each fp32 product from 6 bf16 MFMA products of exactly 3-way split operands (dropped terms < 2^-24 of '
                                         'the product: below fp32 rounding) '
                                         '(csrc/conv_x3.hip; JVAE_X3=0: fp32 MFMA in every layer)'
                                         if (a.dtype == 'f32' and os.environ.get('JVAE_X3', '1') != '0') else
                                         ('fp32 MFMA in every layer' if a.dtype == 'f32' else
                                          'bf16 activations / MFMA operands, fp32 accumulation, statistics, losses, optimiser'))},
               'step_mfma_frac': value / world * WORKLOADS[a.workload][1] / (MFMA_F32_PEAK if a.dtype == 'f32' else MFMA_BF16_PEAK),
               'final_loss': float(losses['total'].detach().mean())}
        if a.workload == 2 and a.dtype == 'f32':      # the roofline probe and the CPU baseline belong to the headline config
            out['roofline'] = dominant_kernel_roofline(device)
        if world == 1 and not a.no_cpu_baseline and a.workload == 2 and a.dtype == 'f32':
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    fh.cancel_dump_traceback_later()


if __name__ == '__main__':
    main()
