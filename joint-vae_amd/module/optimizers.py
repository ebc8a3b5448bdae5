"""Optimiser of the joint CVAE: global-norm clipping + Adam on flat fp32 buffers, on HIP kernels.

Public surface of the reference's module/optimizers.py:14-134 (`Optimizer(parameters, optim_type, lr,
lr_decay, weight_decay, grad_clipping, **kw)` with `.params .lr .kind .zero_grad() .clip() .step()
.update_lr() .update_scheduler_from_epoch() .state_dict() .load_state_dict() .to() .__format__`), where the
reference wraps torch.optim.Adam + clip_grad_norm_ + ExponentialLR.

MI355X design: every parameter that receives gradient is re-pointed (`.data`, `.grad`) into ONE flat,
16-byte aligned device buffer, so that
  * zero_grad is one memset,
  * clip_grad_norm_ is one squared-norm reduction whose scalar stays on the device,
  * Adam (+ L2 weight decay + the clip coefficient + the NaN/Inf scan of cvae.py:2454-2457) is ONE kernel,
  * the data-parallel gradient exchange is ONE RCCL all-reduce of that buffer (latency-bound at ~6 MB).
Parameters that never receive gradient (e.g. the classifier when gamma = 0, the scalar prior variance) are
left untouched exactly as torch.optim.Adam skips `grad is None` parameters; they join (as a new group with
its own step count) the first time a gradient shows up.
"""
import logging

import torch

from jvae_hip import ops
from jvae_compat import texify_str

default_lr = {'sgd': 0.01, 'adam': 0.001}
params_by_type = {'sgd': ('momentum', 'nesterov', 'weight_decay'),
                  'adam': ('betas', 'weight_decay', 'amsgrad')}

_ALIGN = 4        # floats: every tensor starts on a 16-byte boundary inside the flat buffer


class _FlatGroup:
    """Parameters that joined at the same time: contiguous p / g / m / v buffers and one step count."""

    def __init__(self, params, device, step=0):
        self.params = list(params)
        self.offsets = []
        n = 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = n
        self.step = step
        self.p = torch.zeros(n, device=device, dtype=torch.float32)
        self.g = torch.zeros(n, device=device, dtype=torch.float32)
        self.m = torch.zeros(n, device=device, dtype=torch.float32)
        self.v = torch.zeros(n, device=device, dtype=torch.float32)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                k = p.numel()
                self.p[o:o + k].copy_(p.data.reshape(-1))
                if p.grad is not None:
                    self.g[o:o + k].copy_(p.grad.reshape(-1))
                p.data = self.p[o:o + k].view(p.shape)
                p.grad = self.g[o:o + k].view(p.shape)

    def view(self, buf, i):
        p, o = self.params[i], self.offsets[i]
        return buf[o:o + p.numel()].view(p.shape)

    def intact(self):
        """Do the parameters still live in this group's buffers and still want gradients?"""
        base_p, base_g = self.p.data_ptr(), self.g.data_ptr()
        for p, o in zip(self.params, self.offsets):
            if not p.requires_grad or p.grad is None:
                return False
            if p.data_ptr() != base_p + 4 * o or p.grad.data_ptr() != base_g + 4 * o:
                return False
        return True


class Optimizer:

    def __init__(self, parameters, optim_type='adam', lr=0, lr_decay=0, weight_decay=0, grad_clipping=None,
                 epoch=0, **kw):
        if optim_type not in ('adam', 'sgd'):
            raise ValueError("optim_type must be 'adam' or 'sgd' (got {!r})".format(optim_type))
        if optim_type == 'sgd' and kw.get('dampening'):
            raise NotImplementedError('SGD dampening is outside the native-kernel contract')
        if kw.get('amsgrad'):
            raise NotImplementedError('amsgrad is outside the native-kernel contract')
        self.kind = optim_type
        lr = lr or default_lr[optim_type]
        self.params = {'optim_type': optim_type, 'lr': lr, 'lr_decay': lr_decay, 'weight_decay': weight_decay,
                       'grad_clipping': grad_clipping}
        self.params.update(kw)
        self.grad_clipping = grad_clipping
        self.init_lr = lr
        self.lr_decay = lr_decay
        self.weight_decay = weight_decay
        self.betas = tuple(kw.get('betas', (0.9, 0.999)))
        self.eps = kw.get('eps', 1e-8)
        self.momentum = float(kw.get('momentum', 0.))          # optim_type='sgd' (module/optimizers.py:39-40 of the reference)
        self.nesterov = bool(kw.get('nesterov', False))
        self._lr = lr
        self._epochs_decayed = 0
        self._all = [p for p in parameters]
        self._groups = []
        self._clip_pending = False
        self._sqnorm = None          # device scalar: squared global gradient norm of the last clip()
        self._flag = None            # device int: set by the Adam kernel when a parameter became NaN/Inf
        self._scan_pending = True    # parameters not yet vouched for (fresh / loaded / moved): check_nonfinite() scans them once
        self._flag_wanted = False    # somebody calls check_nonfinite(): only then is the flag copied to the host per step
        self._world = 1
        self._pg = None
        self._reduced = False
        logging.debug('Creating optimizer with params %s', ' ; '.join(f'{k}:{v}' for k, v in self.params.items()))

    # ---- learning rate -----------------------------------------------------------------------------
    @property
    def lr(self):
        return self._lr

    def update_lr(self):
        if self.lr_decay:
            old = self._lr
            self._epochs_decayed += 1
            # the chainable form of torch's ExponentialLR.step() (lr <- lr * gamma), which is what the reference runs
            # (optimizers.py:50-53,123-127): bit-identical lr sequence, and after load_state_dict() - which restores the
            # decayed lr - update_scheduler_from_epoch(n) multiplies n MORE times, as the reference's resume does (sic:
            # cvae.py:2851; pinned by tests/golden/ckpt_ref_e2)
            self._lr = self._lr * (1 - self.lr_decay)
            self._sync_device_lr()
            logging.debug(f'lr updated from {old:.4e} to {self._lr:.4e}')

    def update_scheduler_from_epoch(self, n):
        if self.lr_decay:
            for _ in range(n):
                self.update_lr()

    # ---- data-parallel hook (no equivalent in the single-process reference) ----------
    def set_distributed(self, world_size, process_group=None, broadcast=True):
        """Average gradients over `world_size` ranks (RCCL all-reduce of the flat gradient buffer) before clip/step.
        broadcast: rank 0's optimiser state (Adam moments, step count, lr) replaces every other rank's, so replicas do not
        depend on identical seeding / loading (the PARAMETERS and BatchNorm buffers are broadcast by the model:
        ClassificationVariationalNetwork.set_distributed)."""
        self._world = int(world_size)
        self._pg = process_group
        if self._world > 1 and broadcast:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                meta = torch.tensor([float(self._lr)] + [float(g.step) for g in self._groups], dtype=torch.float64)
                if dist.get_backend(process_group) == 'nccl':          # RCCL moves device memory only
                    meta = meta.to(torch.device('cuda', torch.cuda.current_device()))
                dist.broadcast(meta, 0, group=process_group)
                self._lr = float(meta[0])
                staged = dist.get_backend(process_group) != 'nccl'
                for i, g in enumerate(self._groups):
                    g.step = int(meta[1 + i])
                    for buf in (g.m, g.v):
                        if staged and buf.is_cuda:                     # gloo rehearsal on a GPU: through host memory
                            h = buf.cpu()
                            dist.broadcast(h, 0, group=process_group)
                            buf.copy_(h)
                        else:
                            dist.broadcast(buf, 0, group=process_group)
                self._sync_device_lr()

    def all_reduce_flat(self):
        """ONE all-reduce (AVG) per flat group, eagerly, on the current stream: the exchange step between the two captured
        halves of a data-parallel graph step (graph_train_step)."""
        if self._world > 1:
            import torch.distributed as dist
            for g in self._groups:
                dist.all_reduce(g.g, op=dist.ReduceOp.AVG, group=self._pg)

    def set_early_bucket(self, params):
        """Parameters whose gradients are complete first in backward (the decoder / imager): their slice of the flat
        buffer is all-reduced as soon as `reduce_early_bucket()` is called, overlapping the rest of backward."""
        self._early = [p for p in params]
        self._early_work = None
        self._early_range = None

    def _early_slice(self):
        """(group, lo, hi) of the contiguous flat range holding the early-bucket parameters, or None."""
        ids = {id(p) for p in getattr(self, '_early', [])}
        if not ids or len(self._groups) != 1:
            # no group yet = the first step of a model (the flat buffer is built from the first gradients): not worth a warning
            if ids and self._world > 1 and len(self._groups) > 1 and not getattr(self, '_early_warned', False):
                self._early_warned = True
                logging.warning('data-parallel: %d flat gradient groups - the early (decoder) bucket is not overlapped',
                                len(self._groups))
            return None
        g = self._groups[0]
        idx = [i for i, p in enumerate(g.params) if id(p) in ids]
        if not idx or idx != list(range(idx[0], idx[-1] + 1)):
            if idx and self._world > 1 and not getattr(self, '_early_warned', False):
                self._early_warned = True
                logging.warning('data-parallel: the early bucket is not contiguous in the flat buffer - not overlapped')
            return None
        lo = g.offsets[idx[0]]
        hi = g.offsets[idx[-1] + 1] if idx[-1] + 1 < len(g.params) else g.numel
        return g, lo, hi

    def reduce_early_bucket(self):
        """Called from a hook once the early bucket's backward kernels are queued (main + side stream)."""
        if self._world <= 1 or self._reduced or getattr(self, '_early_work', None) is not None:
            return
        sl = self._early_slice()
        if sl is None or not all(gr.intact() for gr in self._groups):
            return
        import torch.distributed as dist
        from jvae_hip import lib as _lib
        g, lo, hi = sl
        dev = g.g.device
        if dev.type == 'cuda':
            side = _lib.side_stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))      # dgrad chain so far + wgrads already on `side`
            with torch.cuda.stream(side):
                self._early_work = dist.all_reduce(g.g[lo:hi], op=dist.ReduceOp.AVG, group=self._pg, async_op=True)
        else:
            self._early_work = dist.all_reduce(g.g[lo:hi], op=dist.ReduceOp.AVG, group=self._pg, async_op=True)
        self._early_range = (lo, hi)

    def reduce_gradients(self):
        if getattr(self, '_external_reduce', False):     # graph step: join + exchange happen outside the captured halves
            return
        from jvae_hip import lib as _lib
        if _lib._side_streams:
            _lib.join_side_stream()              # weight gradients written on the side stream are complete
        if self._world > 1 and not self._reduced:
            import torch.distributed as dist
            self._adopt_new()
            early = getattr(self, '_early_work', None)
            for g in self._groups:
                if early is not None and g is self._groups[0]:
                    lo, hi = self._early_range            # the rest of the buffer: two slices around the early bucket
                    if lo > 0:
                        dist.all_reduce(g.g[:lo], op=dist.ReduceOp.AVG, group=self._pg)
                    if hi < g.numel:
                        dist.all_reduce(g.g[hi:], op=dist.ReduceOp.AVG, group=self._pg)
                    early.wait()
                else:
                    dist.all_reduce(g.g, op=dist.ReduceOp.AVG, group=self._pg)
            self._early_work = None
            self._reduced = True

    # ---- flat-buffer management -------------------------------------------------------------------
    def _device(self):
        for p in self._all:
            return p.device
        return torch.device('cpu')

    def _flat_ids(self):
        return {id(p) for g in self._groups for p in g.params}

    def _adopt_new(self):
        """Parameters with a gradient that are not in a flat group yet join a new one."""
        if self._groups and not all(g.intact() for g in self._groups):
            self._rebuild()
        known = self._flat_ids()
        fresh = [p for p in self._all if id(p) not in known and p.requires_grad and p.grad is not None]
        early = {id(p) for p in getattr(self, '_early', [])}
        if early:
            # the early (decoder) bucket goes to the END of the flat buffer: the data-parallel exchange is then exactly
            # two collectives - the bucket, launched from the hook on z, and ONE contiguous remainder
            fresh = [p for p in fresh if id(p) not in early] + [p for p in fresh if id(p) in early]
        if fresh:
            # flattening itself is device-agnostic (the data-parallel exchange is tested on CPU with gloo);
            # clip() / step() launch HIP kernels and raise for tensors that are not on the GPU
            self._groups.append(_FlatGroup(fresh, fresh[0].device))

    def _rebuild(self):
        """Something moved the parameters (.to(), load_state_dict, requires_grad flip): re-flatten, keep state."""
        old = self._groups
        self._groups = []
        for g in old:
            keep = [i for i, p in enumerate(g.params) if p.requires_grad]
            if not keep:
                continue
            ps = [g.params[i] for i in keep]
            dev = ps[0].device
            if dev.type != 'cuda':
                self._cpu_state = getattr(self, '_cpu_state', [])      # state parked until we are back on a GPU
                self._cpu_state.append((ps, [g.view(g.m, i).detach().cpu().clone() for i in keep],
                                        [g.view(g.v, i).detach().cpu().clone() for i in keep], g.step))
                continue
            ng = _FlatGroup(ps, dev, g.step)
            with torch.no_grad():
                for j, i in enumerate(keep):
                    ng.view(ng.m, j).copy_(g.view(g.m, i).to(dev))
                    ng.view(ng.v, j).copy_(g.view(g.v, i).to(dev))
            self._groups.append(ng)

    def _restore_parked(self):
        parked = getattr(self, '_cpu_state', None)
        if not parked:
            return
        rest = []
        for ps, ms, vs, step in parked:
            dev = ps[0].device
            if dev.type != 'cuda':
                rest.append((ps, ms, vs, step))
                continue
            ng = _FlatGroup(ps, dev, step)
            with torch.no_grad():
                for j in range(len(ps)):
                    ng.view(ng.m, j).copy_(ms[j].to(dev))
                    ng.view(ng.v, j).copy_(vs[j].to(dev))
            self._groups.append(ng)
        self._cpu_state = rest

    def to(self, device):
        """Called by the model's .to(): parameters have just been moved, so re-flatten on their new device."""
        logging.debug(f'Sending optimizer to {device}')
        if self._groups:
            self._rebuild()
        self._restore_parked()
        self._sqnorm = None
        self._flag = None
        self._scan_pending = True

    # ---- the step ------------------------------------------------------------------------------------
    def zero_grad(self, *a, **kw):
        from jvae_hip import lib as _lib
        _lib.pack_cache_end()                    # a new step: whatever span of constant weights was left open is over
        flat = self._flat_ids()
        for g in self._groups:
            g.g.zero_()
        for p in self._all:
            if id(p) not in flat:
                p.grad = None
        self._clip_pending = False
        self._reduced = False
        self._early_work = None

    def clip(self, parameters=None):
        """clip_grad_norm_(all parameters, grad_clipping): the norm is reduced here, the coefficient is
        applied inside the Adam kernel of the following step() (and to .grad too if step() does not follow)."""
        self.reduce_gradients()
        if not self.grad_clipping:
            return
        self._adopt_new()
        if not self._groups:
            return
        dev = self._groups[0].g.device
        if self._sqnorm is None or self._sqnorm.device != dev:
            self._sqnorm = torch.zeros(1, device=dev, dtype=torch.float32)
        for i, g in enumerate(self._groups):
            ops.sqnorm_accum(g.g, self._sqnorm, reset=(i == 0))
        self._clip_pending = True

    def grad_norm(self):
        """Global gradient norm measured by the last clip() (device scalar tensor)."""
        return None if self._sqnorm is None else self._sqnorm.sqrt()

    def apply_clip_to_grads(self):
        """Materialise the clipped gradients in .grad (what clip_grad_norm_ leaves behind)."""
        if self._clip_pending:
            for g in self._groups:
                ops.clip_scale(g.g, self._sqnorm, self.grad_clipping)
            self._clip_pending = False

    def step(self):
        from jvae_hip import lib as _lib
        _lib.pack_cache_end()                    # the weights change below: packed operand forms are no longer valid
        self.reduce_gradients()
        self._adopt_new()
        if not self._groups:
            return
        dev = self._groups[0].g.device
        if self._flag is None or self._flag.device != dev:
            self._flag = torch.zeros(1, device=dev, dtype=torch.int32)
        clip = self.grad_clipping if self._clip_pending else 0.
        for g in self._groups:
            g.step += 1
            if self.kind == 'sgd':
                # the momentum buffer lives in g.m; the first step initialises it with the gradient, as torch does
                ops.sgd_step(g.p, g.g, g.m if self.momentum else None, self._lr, self.momentum, self.nesterov,
                             self.weight_decay, g.step == 1, max_norm=clip, sqnorm=self._sqnorm if clip else None,
                             flag=self._flag)
                continue
            if getattr(self, '_device_hyper', False):
                # step count / lr / betas live in a device block (capturable into a HIP graph, see enable_device_hyper)
                if getattr(g, 'hyper', None) is None or g.hyper.device != dev:
                    g.hyper = torch.tensor([self._lr, self.betas[0], self.betas[1], float(g.step - 1), 0., 0.],
                                           device=dev, dtype=torch.float32)
                ops.adam_step_dev(g.p, g.g, g.m, g.v, g.hyper, True, self.eps, self.weight_decay, max_norm=clip,
                                  sqnorm=self._sqnorm if clip else None, flag=self._flag)
            else:
                ops.adam_step(g.p, g.g, g.m, g.v, self._lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                              g.step, max_norm=clip, sqnorm=self._sqnorm if clip else None, flag=self._flag)
        self._clip_pending = False
        self._post_flag(dev)

    # ---- NaN / Inf parameters: the reference scans every parameter before each backward (cvae.py:2454-2457) ---------
    def _post_flag(self, dev):
        """After the update: the kernel's non-finite flag travels to pinned host memory on the side stream (4 bytes, off
        the critical path); check_nonfinite() reads it before the NEXT backward.  Only when somebody reads it: a loop that
        never calls check_nonfinite() pays neither the side-stream wait nor the copy."""
        if not self._flag_wanted or torch.cuda.is_current_stream_capturing():
            return                               # a captured step reports through its measures (graph_train_step)
        from jvae_hip import lib as _lib
        if getattr(self, '_flag_host', None) is None:
            self._flag_host = torch.zeros(1, dtype=torch.int32, pin_memory=True)
        side = _lib.side_stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self._flag_host.copy_(self._flag, non_blocking=True)
            self._flag_event = torch.cuda.Event()
            self._flag_event.record(side)

    def check_nonfinite(self):
        """True if a parameter is NaN / Inf.  Call it between the forward and the backward of a step - where the reference's
        scan sits (cvae.py:2454-2457: every parameter, `print('GRAD NAN'); sys.exit(1)` before backward); train_step() does,
        a hand-written evaluate / backward / clip / step loop should too (this is the public check).

        What is looked at: (i) ONCE after construction, load_state_dict() and .to() - parameters nobody has vouched for yet -
        every parameter is scanned on the device (one host synchronisation), so NaN / Inf present at load time or written by
        anything but the optimiser before the first step are caught on the very first step, as in the reference; (ii) from
        then on the flag the update kernel raised while writing the PREVIOUS step's parameters: the only writer on the
        training path.  Its 4-byte copy finished while this step's forward was being enqueued, so the host does not stall
        on it unless it runs more than a step ahead of the GPU.  Narrower than the reference in one respect: a parameter
        poisoned behind the optimiser's back (`.data` arithmetic) after the first step is seen only when it reaches the
        update kernel (one step later, through its NaN gradient / moment)."""
        self._flag_wanted = True
        if self._scan_pending:
            self._scan_pending = False
            flat = self._flat_ids()
            tensors = [g.p for g in self._groups] + [p.data for p in self._all if id(p) not in flat]
            if any(t.is_cuda and t.is_floating_point() and not bool(torch.isfinite(t).all()) for t in tensors):
                return True
        ev = getattr(self, '_flag_event', None)
        if ev is None:
            return False
        ev.synchronize()
        self._flag_event = None
        return bool(self._flag_host[0] != 0)

    def enable_device_hyper(self, on=True):
        """Keep Adam's step count, learning rate and betas in device memory (updated by a one-thread kernel in front of
        the update) so that a captured training step replays with the right bias correction.  The host-side step count
        stays the reference for state_dict(); a graph replay must call note_replayed_step()."""
        self._device_hyper = bool(on)
        for g in self._groups:
            g.hyper = None
        return self

    def note_replayed_step(self):
        """Host bookkeeping for one optimiser step that ran inside a replayed HIP graph."""
        for g in self._groups:
            g.step += 1

    def _sync_device_lr(self):
        for g in self._groups:
            if getattr(g, 'hyper', None) is not None:
                g.hyper[0] = self._lr

    def nonfinite_flag(self):
        """Device int32 tensor, non-zero once any updated parameter was NaN/Inf (None before the first step)."""
        return self._flag

    # ---- persistence (torch.optim.Adam state_dict layout, so optimizer.pth round-trips) ---------------
    def state_dict(self, *a, **k):
        index = {id(p): i for i, p in enumerate(self._all)}
        state = {}
        if self.kind == 'sgd':                     # torch.optim.SGD layout
            for g in self._groups:
                for j, p in enumerate(g.params):
                    buf = g.view(g.m, j).detach().clone() if self.momentum and g.step > 0 else None
                    state[index[id(p)]] = {'momentum_buffer': buf}
            group = {'lr': self._lr, 'momentum': self.momentum, 'dampening': 0, 'weight_decay': self.weight_decay,
                     'nesterov': self.nesterov, 'maximize': False, 'foreach': None, 'differentiable': False, 'fused': None,
                     'initial_lr': self.init_lr, 'params': list(range(len(self._all)))}
            return {'state': state, 'param_groups': [group]}
        for g in self._groups:
            for j, p in enumerate(g.params):
                state[index[id(p)]] = {'step': torch.tensor(float(g.step)),
                                       'exp_avg': g.view(g.m, j).detach().clone(),
                                       'exp_avg_sq': g.view(g.v, j).detach().clone()}
        group = {'lr': self._lr, 'betas': self.betas, 'eps': self.eps, 'weight_decay': self.weight_decay,
                 'amsgrad': False, 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False,
                 'fused': None, 'decoupled_weight_decay': False, 'initial_lr': self.init_lr,
                 'params': list(range(len(self._all)))}
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd, *a, **k):
        pg = sd['param_groups'][0]
        self._lr = pg['lr']
        self.betas = tuple(pg.get('betas', self.betas))
        self.eps = pg.get('eps', self.eps)
        self.weight_decay = pg.get('weight_decay', self.weight_decay)
        if self.kind == 'sgd':
            self.momentum = float(pg.get('momentum', self.momentum))
            self.nesterov = bool(pg.get('nesterov', self.nesterov))
            idxs = sorted(int(i) for i, st in sd['state'].items() if st.get('momentum_buffer') is not None)
            self._groups = []
            if idxs:
                ps = [self._all[i] for i in idxs]
                if ps[0].device.type != 'cuda':
                    raise NotImplementedError('load the SGD state after the model has been moved to the GPU')
                g = _FlatGroup(ps, ps[0].device, 1)            # step >= 1: the buffers are initialised
                with torch.no_grad():
                    for j, i in enumerate(idxs):
                        g.view(g.m, j).copy_(sd['state'][i]['momentum_buffer'].to(ps[0].device))
                self._groups.append(g)
            return
        by_step = {}
        for idx, st in sd['state'].items():
            by_step.setdefault(int(float(st['step'])), []).append(int(idx))
        self._groups = []
        for step, idxs in sorted(by_step.items(), reverse=True):
            ps = [self._all[i] for i in sorted(idxs)]
            dev = ps[0].device
            if dev.type != 'cuda':
                self._cpu_state = getattr(self, '_cpu_state', [])
                self._cpu_state.append((ps, [sd['state'][i]['exp_avg'].clone() for i in sorted(idxs)],
                                        [sd['state'][i]['exp_avg_sq'].clone() for i in sorted(idxs)], step))
                continue
            g = _FlatGroup(ps, dev, step)
            with torch.no_grad():
                for j, i in enumerate(sorted(idxs)):
                    g.view(g.m, j).copy_(sd['state'][i]['exp_avg'].to(dev))
                    g.view(g.v, j).copy_(sd['state'][i]['exp_avg_sq'].to(dev))
            self._groups.append(g)

    # ---- printing --------------------------------------------------------------------------------------
    def __str__(self):
        return self.__format__('10')

    def __format__(self, format_spec):
        if format_spec.endswith('x'):
            return texify_str(self.__format__(format_spec[:-1]), num=True)
        try:
            level = int(format_spec)
        except ValueError:
            level = 0
        if not level:
            return self.__str__()
        parts = [self.kind, f'lr={self.init_lr}']
        if self.lr_decay:
            parts.append(f'decay={self.lr_decay}')
        else:
            level -= 1
        shown = {'betas': self.betas, 'weight_decay': self.weight_decay, 'amsgrad': False, 'momentum': self.momentum,
                 'nesterov': self.nesterov}
        extra = [f'{k}={shown[k]}' for k in params_by_type[self.kind] if shown[k] and not isinstance(shown[k], bool)]
        if extra:
            parts.append('--'.join(extra))
        return '--'.join(parts[:level])
