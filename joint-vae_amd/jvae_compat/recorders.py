"""Per-sample loss records of a dataset in the reference's on-disk format.

`record-<set>.pth` / `samples-<set>.pth` are `torch.save`d dictionaries (reference utils/save_load/recorders.py:107-174):

    {'batch_size', 'last_batch_size', 'device', '_seed', '_num_batch', '_samples', '_recorded_batches',
     '_tensors': {name: tensor whose SAMPLE axis (the last one for losses - all-class losses are (C, N) - the first one for
                  sample recorders) holds `_samples` = `_num_batch` * `batch_size` slots}, ['_aux']}

`test.py`, `results/` and the OOD tooling of the reference read these files, so jobs trained with this package must
write them the same way; `LossRecorder.load` here also reads files the reference wrote (tests/golden/record_ref).  The
classes are written against that file layout - storage is preallocated and filled with narrow().copy_(), batches are read
back as views - not against the reference's code.
"""
import logging
import os
import re

import numpy as np
import torch


class LossRecorder:
    file_pattern = 'record-{w}.pth'
    _file_pattern = file_pattern          # the reference's spelling
    _sample_dim = -1

    def __init__(self, batch_size, num_batch=1, device=None, **tensors):
        self.batch_size = batch_size
        self.last_batch_size = batch_size
        self.device = device
        self._seed = None
        self._num_batch = 0
        self._samples = 0
        self._recorded_batches = 0
        self._tensors = {}
        self.reset()
        if tensors:
            self._allocate(num_batch, tensors)

    # ---- storage ---------------------------------------------------------------------------------
    def _axis(self, t):
        return self._sample_dim % t.dim()

    def _allocate(self, num_batch, like):
        assert not self._tensors, 'the recorder already has its tensors'
        if self.device is None:
            self.device = next(iter(like.values())).device
        self._num_batch = int(num_batch)
        self._samples = self._num_batch * self.batch_size
        for name, t in like.items():
            shape = list(t.shape)
            shape[self._axis(t)] = self._samples
            self._tensors[name] = torch.zeros(shape, dtype=t.dtype, device=self.device)
        self.last_batch_size = self.batch_size

    @property
    def num_batch(self):
        return self._num_batch

    @num_batch.setter
    def num_batch(self, n):
        """Capacity in batches: grows the storage (zero-filled), never shrinks it; forgets batches beyond n."""
        if not self._tensors:
            return
        want = int(n) * self.batch_size
        for name, t in list(self._tensors.items()):
            ax = self._axis(t)
            if want > t.shape[ax]:
                pad = list(t.shape)
                pad[ax] = want - t.shape[ax]
                self._tensors[name] = torch.cat([t, torch.zeros(pad, dtype=t.dtype, device=t.device)], dim=ax)
        self._num_batch = int(n)
        self._samples = want
        self._recorded_batches = min(int(n), self._recorded_batches)

    def to(self, device):
        self._tensors = {k: t.to(device) for k, t in self._tensors.items()}
        self.device = device

    def reset(self, seed=False):
        self._recorded_batches = 0
        if self._seed is None or seed:
            self._seed = int(np.random.randint(1, int(1e8)))
        self.last_batch_size = self.batch_size

    # the reference re-seeds torch around its DataLoader so that a recorded pass and a recovered one see the same order
    def init_seed_for_dataloader(self):
        self._initial_seed = torch.seed()
        torch.manual_seed(self._seed)

    def restore_seed(self):
        torch.manual_seed(self._initial_seed)

    # ---- container protocol ------------------------------------------------------------------------
    def keys(self):
        return self._tensors.keys()

    def __iter__(self):
        return iter(self._tensors)

    def __len__(self):
        return self._recorded_batches

    def __repr__(self):
        return 'Recorder for ' + ' '.join(str(k) for k in self.keys())

    @property
    def recorded_samples(self):
        return (len(self) - 1) * self.batch_size + self.last_batch_size

    def __getitem__(self, name):
        t = self._tensors[name]
        return t.narrow(self._axis(t), 0, max(self.recorded_samples, 0)).clone()

    def pop(self, name):
        out = self[name]
        del self._tensors[name]
        return out

    def has_batch(self, number, only_full=False):
        if number == len(self) - 1:
            return not only_full or self.last_batch_size == self.batch_size
        return number < self._recorded_batches

    def get_batch(self, i, *which, device=None, force_dict=False):
        if not which:
            if not self._tensors:
                raise KeyError('empty recorder')
            which, force_dict = tuple(self._tensors), True
        if len(which) > 1 or force_dict:
            return {w: self.get_batch(i, w, device=device) for w in which}
        if not self.has_batch(i):
            raise IndexError(f'{i} >= {len(self)}')
        t = self._tensors[which[0]]
        size = self.last_batch_size if i == len(self) - 1 else self.batch_size
        out = t.narrow(self._axis(t), i * self.batch_size, size).clone()
        return out.to(device) if device else out

    def append_batch(self, extend=True, **tensors):
        if not self._tensors:
            self._allocate(1, tensors)
        sizes = {t.shape[self._axis(t)] for t in tensors.values()}
        assert len(sizes) == 1, 'all batches have to be of same size'
        size = sizes.pop()
        assert size <= self.batch_size, 'appended batch to large'
        assert self.last_batch_size == self.batch_size, 'only the last batch of a record may be partial'
        start = self._recorded_batches * self.batch_size
        if start + self.batch_size > self._samples:
            if not extend:
                raise IndexError
            self.num_batch = max(1, self._num_batch) * 2
        for name, t in tensors.items():
            if name not in self._tensors:
                raise KeyError(name)
            store = self._tensors[name]
            store.narrow(self._axis(store), start, size).copy_(t)
        self.last_batch_size = size
        self._recorded_batches += 1

    # ---- files ---------------------------------------------------------------------------------------
    def _state(self):
        return {k: v for k, v in self.__dict__.items() if k != '_initial_seed'}

    def save(self, file_path, cut=True, append=False):
        if append:
            try:
                merged = type(self).load(file_path)
                merged.merge(self)
            except FileNotFoundError:
                merged = self
            merged.save(file_path, cut=cut, append=False)
            return
        if cut:                                       # keep the recorded samples only
            n = self.recorded_samples
            self.num_batch = len(self)
            self._tensors = {k: t.narrow(self._axis(t), 0, n).clone() for k, t in self._tensors.items()}
        torch.save(self._state(), file_path)

    @classmethod
    def load(cls, file_path, device=None, **kw):
        if 'map_location' not in kw and not torch.cuda.is_available():
            kw['map_location'] = torch.device('cpu')
            device = 'cpu'
        kw.setdefault('weights_only', False)          # the file is a plain dict with a torch.device in it
        d = torch.load(file_path, **kw)
        r = cls(d['batch_size'], d['_num_batch'], **d['_tensors'])
        for k, v in d.items():
            if k in ('_seed', '_tensors', '_recorded_batches', '_aux') or not k.startswith('_'):
                setattr(r, k, v)
        if isinstance(r.last_batch_size, dict):       # files of an older reference version
            r.last_batch_size = next(iter(r.last_batch_size.values()))
        if device:
            r.to(device)
        return r

    @classmethod
    def loadall(cls, dir_path, *names, file_name=None, output='recorders', **kw):
        file_name = file_name or cls.file_pattern
        found = {}
        if not names:
            rx = re.compile(re.escape(file_name).replace(re.escape('{w}'), '(?P<name>.+)') + '$')
            for f in sorted(os.listdir(dir_path)):
                m = rx.match(f)
                if m:
                    found[m.group('name')] = os.path.join(dir_path, f)
        for w in names:
            p = os.path.join(dir_path, file_name.format(w=w))
            if os.path.exists(p):
                found[w] = p
            else:
                logging.warning('%s not found', os.path.basename(p))
        if output.startswith('record'):
            return {w: cls.load(p, **kw) for w, p in found.items()}
        return found

    # ---- combining -------------------------------------------------------------------------------------
    def copy(self, device=None):
        twin = type(self)(self.batch_size)
        for i in range(len(self)):
            twin.append_batch(**self.get_batch(i, device=device))
        return twin

    def merge(self, other, axis='samples'):
        assert isinstance(other, type(self))
        assert axis in ('samples', 'keys'), 'axis has to be either samples or keys'
        shared = [k for k in self if k in set(other)]
        if axis == 'keys':
            assert self.recorded_samples == other.recorded_samples
            assert not shared, 'can not merge recorder with common keys ({})'.format(', '.join(shared))
            self._tensors.update(other._tensors)
            return
        total = self.recorded_samples + other.recorded_samples
        joined = {k: torch.cat((self[k], other[k]), dim=self._axis(self._tensors[k])) for k in shared}
        # capacity bookkeeping as the reference leaves it (recorders.py:245-262): room for the added samples + 1 batch
        self._num_batch = len(self) + other.recorded_samples // self.batch_size + 1
        self._samples = self._num_batch * self.batch_size
        self._tensors = joined
        self._recorded_batches = (total - 1) // self.batch_size + 1
        self.last_batch_size = (total - 1) % self.batch_size + 1

    def split(self, *keys, keep=False):
        other = self.copy()
        for k in list(self):
            if k in keys:
                if not keep:
                    self.pop(k)
            else:
                other.pop(k)
        return other


class SampleRecorder(LossRecorder):
    file_pattern = 'samples-{w}.pth'
    _file_pattern = file_pattern
    _sample_dim = 0

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._aux = {}

    def __repr__(self):
        text = 'Sample Recorder for ' + ' '.join(str(k) for k in self.keys())
        if self._aux:
            text += ' with aux data {}'.format(', '.join(self._aux))
        return text

    def add_auxiliary(self, **t):
        self._aux.update(t)
