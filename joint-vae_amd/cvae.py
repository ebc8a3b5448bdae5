"""MI355X-native joint / class-conditional VAE: drop-in for the reference's `cvae.ClassificationVariationalNetwork`.

Scope: the per-batch training step of the reference (cvae.py:2424-2479) —
features -> encoder -> (mu, log sigma^2) -> reparameterised z -> decoder / imager -> ELBO terms -> backward ->
clip -> Adam — executed by hand-written gfx950 HIP kernels (libjvae_hip.so) behind the same Python API:

    ClassificationVariationalNetwork(input_shape, num_labels, type, ..., sigma, optimizer, ...)   cvae.py:135-167
    .forward(x, y=None, x_features=None, z_output=True, sampling_epsilon_norm_out=False, sigma_out=False)  :426-521
    .evaluate(x, y, batch, current_measures, with_beta, kl_var_weighting, gamma_weighting, z_output)       :523-917
    .train_model(...)   (hot loop :2424-2479; test / OOD phases are out of scope)                            :2081-2547
    .train() / .to() / .save() / .load() / .latent_sampling / .device / .nparams

What is NOT rebuilt here (raises NotImplementedError when asked for): (cvae / xvae / vae / vib are complete, jvae: training and labelled evaluation only); resnet
feature stacks (torchvision); a per-dimension sigma (the reference fails on it too); label-free evaluation of a categorical
(256-level) decoder - its training / labelled evaluation is built; the misclassification phases of train_model.  Pooling / up-sampling layer tokens, SGD,
the `y=None` all-class evaluation with its OOD scores and the WIM fine-tuning step are built.
There is no CPU path: calling forward/evaluate with CPU tensors raises.
"""
import contextlib
import json
import logging
import math
import os
import sys
import time

import numpy as np
import torch
from torch import nn

from jvae_hip import ops
from jvae_hip import lib as _lib
from module.optimizers import Optimizer
from module.losses import x_loss, mse_loss, categorical_loss  # noqa: F401  (import surface of the reference)
from module.vae_layers import Encoder, Classifier, Sigma, build_de_conv_layers, find_input_shape
from module.vae_layers import onehot_encoding, activation_layers
from module.vae_layers.layers import HipLinear, HipDropout, DenseStack

DEFAULT_ACTIVATION = 'relu'
DEFAULT_OUTPUT_ACTIVATION = 'linear'
DEFAULT_LATENT_SAMPLING = 100
VERSION = 2.
LOG2PI = math.log(2 * math.pi)


class _LazyDict(dict):
    """A dict whose content is produced by `_fill()` the first time anything reads it."""

    def _fill(self):
        raise NotImplementedError

    def copy(self):
        self._fill()
        return dict(self)


def _lazy(name):
    def method(self, *a, **k):
        self._fill()
        return getattr(dict, name)(self, *a, **k)
    method.__name__ = name
    return method


for _n in ('__getitem__', 'get', '__contains__', '__iter__', '__len__', 'keys', 'values', 'items', '__repr__', '__eq__'):
    setattr(_LazyDict, _n, _lazy(_n))


class Measures(_LazyDict):
    """`total_measures` of evaluate(): a dict of Python floats (rmse, dB, sigma, ...) exactly as in the reference, but
    materialised LAZILY: the 16 scalars sit in one device buffer (running means included, continued on the device from
    batch to batch) that is copied to pinned host memory asynchronously; the first time any entry is read the dict
    waits for that copy.  A training loop that only threads `measures` into the next evaluate() never synchronises."""

    _KEYS = ('sigma', 'xpow', 'mse', 'rmse', 'dB', 'zdist', 'var_kl')

    def __init__(self, dev, has_dictionary, on_nan, from_main=False, only=None):
        super().__init__()
        self._only = only                                # type 'vib': ('sigma', 'zdist', 'var_kl') - nothing is reconstructed
        from jvae_hip import lib as _lib
        self._dev = dev
        self._host = torch.empty(16, dtype=torch.float32, pin_memory=True)
        side = _lib.side_stream(dev.device)          # `dev` was produced on the side stream (logging is off the critical path)
        if from_main:                                # ... or by a graph replay on the current stream
            side.wait_stream(torch.cuda.current_stream(dev.device))
        with torch.cuda.stream(side):
            self._host.copy_(dev, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record(side)
        self._has_dictionary = has_dictionary
        self._on_nan = on_nan
        self._ready = False

    def _fill(self):
        if self._ready:
            return
        self._ready = True
        self._event.synchronize()
        h = self._host.tolist()
        if h[9] != 0:
            self._on_nan()
        dict.update(self, {'sigma': h[0], 'xpow': h[10], 'mse': h[11], 'rmse': h[12], 'dB': h[13], 'zdist': h[14],
                           'var_kl': h[15]})
        if self._has_dictionary:
            dict.update(self, {'ld-norm': h[6], 'imut-zy': h[7], 'd-mind': h[8]})
        if self._only is not None:
            for k in [k for k in dict.keys(self) if k not in self._only]:
                dict.__delitem__(self, k)


class _LazySigmaParams(_LazyDict):
    """training_parameters['sigma'] (Sigma.params with the current rms value) without a device read-back per batch."""

    def __init__(self, sigma, measures):
        super().__init__()
        self._s, self._m, self._done = sigma, measures, False

    def _fill(self):
        if not self._done:
            self._done = True
            dict.update(self, self._s.host_params(self._m['sigma']))


def _mean_over_draws(logits):
    """y_est = logits[1:].mean(0) (cvae.py:910-917): with ONE latent draw (training: L = 1) that mean is the row itself -
    returned as a view, bit-identical, without a reduction kernel on the step's main stream."""
    return logits[1] if logits.shape[0] == 2 else logits[1:].mean(0)


def _grad_nan_exit():
    print('GRAD NAN')              # cvae.py:2454-2457; detected by the Adam kernel: train_step() checks the flag of the previous update
                                   # before every backward (as the reference's scan does), lazily-read measures check it as well
    sys.exit(1)


class ClassificationVariationalNetwork(nn.Module):
    r"""X -- features -- encoder -- Z -- decoder -- imager -- X^   with a class-conditional prior p(z|y)
    (and a classifier head on z when gamma > 0)."""

    # the tables of cvae.py:82-112 for the two types built here
    loss_components_per_type = {'cvae': ('cross_x', 'kl', 'total', 'zdist', 'var_kl', 'dzdist', 'iws',
                                         'sigma', 'wmse', 'z_logdet', 'z_tr_inv_cov'),
                                'vae': ('cross_x', 'kl', 'zdist', 'var_kl', 'total', 'iws'),
                                'jvae': ('cross_x', 'kl', 'cross_y', 'total'),
                                'xvae': ('cross_x', 'kl', 'total', 'zdist', 'iws'),
                                'vib': ('cross_y', 'kl', 'total')}
    predict_methods_per_type = {'cvae': ['iws', 'closest'], 'vae': [], 'jvae': ['loss', 'esty'], 'xvae': ['loss', 'closest'],
                                'vib': ['esty']}
    metrics_per_type = {'cvae': ['rmse', 'dB', 'd-mind', 'ld-norm', 'sigma'], 'vae': ['rmse', 'dB', 'sigma'],
                        'jvae': ['rmse', 'dB', 'sigma'], 'xvae': ['rmse', 'dB', 'zdist', 'd-mind', 'ld-norm', 'sigma'],
                        'vib': ['sigma']}
    ood_methods_per_type = {'cvae': ['iws-2s', 'iws-a-1-1', 'iws-a-4-1', 'iws', 'mse', 'elbo', 'soft',
                                     'elbo-2s', 'elbo-a-1-1', 'elbo-a-4-1', 'zdist'],
                            'vae': ['iws', 'iws-2s', 'iws-a-1-1', 'iws-a-4-1', 'elbo', 'elbo-2s', 'elbo-a-1-1',
                                    'elbo-a-4-1', 'zdist'],
                            'jvae': ['max', 'sum', 'std'], 'xvae': ['max', 'mean', 'std'],
                            'vib': ['odin*', 'baseline', 'logits']}
    misclass_methods_per_type = {'cvae': ['softkl*', 'iws', 'softiws*', 'kl', 'max', 'zdist', 'softzdist*',
                                          'baseline*', 'hyz'],
                                 'vae': [], 'jvae': [], 'xvae': [], 'vib': ['odin*', 'baseline', 'logits', 'hyz']}

    def __init__(self, input_shape, num_labels, type='cvae', y_is_coded=False, output_distribution='gaussian',
                 job_number=0, features=None, pretrained_features=None, batch_norm=False, dropout=False,
                 encoder=[36], latent_dim=32, prior={}, beta=1., gamma=0., decoder=[36], upsampler=None,
                 pretrained_upsampler=None, classifier=[36], name='joint-vae', activation=DEFAULT_ACTIVATION,
                 latent_sampling=DEFAULT_LATENT_SAMPLING, test_latent_sampling=None,
                 encoder_forced_variance=False, output_activation=DEFAULT_OUTPUT_ACTIVATION, sigma={'value': 1},
                 optimizer={}, shadow=False, representation='rgb', version=VERSION, *args, **kw):
        super().__init__(*args, **kw)
        assert type in ('jvae', 'cvae', 'xvae', 'vib', 'vae')
        assert not (y_is_coded and type in ('vib', 'vae'))
        assert output_distribution in ('gaussian', 'categorical')
        assert not upsampler or features, 'no upsampler without features'

        self.name = name
        self.job_number = job_number
        self.type = type
        self.is_cvae, self.is_jvae, self.is_vib, self.is_vae, self.is_xvae = (type == 'cvae', type == 'jvae', type == 'vib',
                                                                              type == 'vae', type == 'xvae')
        self.loss_components = self.loss_components_per_type[type]
        self.metrics = self.metrics_per_type[type]
        self.predict_methods = list(self.predict_methods_per_type[type])
        self.ood_methods = list(self.ood_methods_per_type[type])
        self.misclass_methods = list(self.misclass_methods_per_type[type])
        self.y_is_coded = y_is_coded
        self.y_is_decoded = gamma if (self.is_cvae or self.is_vae) else True      # cvae.py:196-199
        self.x_is_generated = not self.is_vib                               # cvae.py:201: 'vib' has no decoder / imager
        self.output_distribution = output_distribution if self.x_is_generated else None
        self.losses_might_be_computed_for_each_class = not self.is_vae and not self.is_vib      # cvae.py:205
        if not self.x_is_generated:
            decoder, upsampler = [], None                                   # cvae.py:222-224

        if self.y_is_decoded:
            self.classifier_type = 'linear'
            if classifier and isinstance(classifier[0], str):
                assert classifier[0] in ('softmax',)
                self.classifier_type = classifier[0]
            if 'esty' not in self.predict_methods:
                self.predict_methods.append('esty')
            if 'cross_y' not in self.loss_components:
                self.loss_components += ('cross_y',)
        else:
            self.classifier_type = None
            classifier = []

        bn_enc = bool(features) and batch_norm in ('encoder', 'both')
        bn_dec = bool(features) and batch_norm == 'both'
        if not features:
            batch_norm = False

        if features:
            self.features = build_de_conv_layers(input_shape, features, activation=activation, batch_norm=bn_enc,
                                                 pretrained_dict=pretrained_features)
            enc_in = self.features.output_shape
        else:
            self.features = None
            enc_in = input_shape

        self.trained = 0
        if isinstance(sigma, Sigma):
            self.sigma = sigma
        elif isinstance(sigma, dict):
            self.sigma = Sigma(**sigma)
        else:
            self.sigma = Sigma(value=sigma)
        if self.sigma.per_dim:
            # The reference itself cannot run one: cvae.py:649 divides (L,N,C,H,W) by a (C,H,W) sigma but cvae.py:789 adds
            # its (C,H,W) log to the (N,) wmse - `RuntimeError: The size of tensor a (N) must match ...` for a learned and for
            # a coded per-dimension sigma alike (probed on the reference in the build container).  Same outcome here, earlier.
            raise NotImplementedError('per-dimension sigma: the reference fails on it as well (shape mismatch at '
                                      'cvae.py:789); scalar fixed / decayed / learned / rmse / coded sigma are built')

        test_latent_sampling = test_latent_sampling or latent_sampling
        self.beta = beta
        self.gamma = gamma if self.y_is_decoded else None
        prior = dict(prior)
        if self.is_cvae or self.is_xvae:          # cvae.py:274-275: one prior component per class; 'vae' / 'jvae': a single one
            prior['num_priors'] = num_labels
        self.encoder = Encoder(enc_in, num_labels, intermediate_dims=encoder, latent_dim=latent_dim,
                               y_is_coded=self.y_is_coded, dropout=dropout,
                               sigma_output_dim=self.sigma.output_dim if self.sigma.coded else 0,        # cvae.py:283
                               forced_variance=encoder_forced_variance, sampling_size=latent_sampling, prior=prior,
                               activation=activation, sampling=latent_sampling > 1 or beta > 0)

        dense, width = [], latent_dim
        shared_act = activation_layers[activation]()
        for d in decoder:
            dense += [HipLinear(width, d), shared_act]
            if dropout:
                dense.append(HipDropout(p=dropout))
            width = d
        if not self.x_is_generated:
            pass                                  # cvae.py:291: no `decoder` / `imager` modules (nor state_dict keys) at all
        elif upsampler:
            self.decoder = DenseStack(*dense)
            hw = find_input_shape(upsampler, input_shape[1:])
            cells = hw[0] * hw[1]
            assert not width % cells, 'Could not go from {} to *, {} {}'.format(width, *hw)
            self.imager = build_de_conv_layers((width // cells, *hw), upsampler, batch_norm=bn_dec,
                                               activation=activation, output_activation=output_activation,
                                               output_distribution=self.output_distribution,
                                               pretrained_dict=pretrained_upsampler, where='output')
        else:
            self.decoder = DenseStack(*dense)
            upsampler = None
            self.imager = DenseStack(HipLinear(width, int(np.prod(input_shape))),
                                     activation_layers[output_activation]())
            self.imager.input_shape = (width,)

        if self.classifier_type in ('linear', None):
            self.classifier = Classifier(latent_dim, num_labels, classifier, activation=activation)

        self.input_shape = tuple(input_shape)
        self.num_labels = num_labels
        self.input_dim = len(input_shape)
        self.batch_norm = batch_norm
        self.dropout = dropout
        self._sizes_of_layers = [input_shape, num_labels, encoder, latent_dim, decoder, upsampler, classifier]
        self.architecture = {'input_shape': input_shape, 'num_labels': num_labels,
                             'output_distribution': self.output_distribution, 'type': type,
                             'representation': representation, 'encoder': encoder, 'batch_norm': batch_norm,
                             'dropout': dropout, 'activation': activation,
                             'encoder_forced_variance': self.encoder.forced_variance, 'latent_dim': latent_dim,
                             'test_latent_sampling': test_latent_sampling, 'prior': self.encoder.prior.params,
                             'decoder': decoder, 'upsampler': upsampler, 'classifier': classifier,
                             'output_activation': output_activation, 'version': VERSION}
        if features:
            self.architecture['features'] = self.features.name
        linear_clf = self.classifier_type == 'linear'
        self.depth = (len(encoder) + len(decoder) + len(classifier)) if linear_clf else 0
        self.width = (sum(encoder) + sum(decoder) + sum(classifier)) if linear_clf else 0

        self.training_parameters = {'sigma': self.sigma.params, 'beta': self.beta, 'gamma': self.gamma,
                                    'latent_sampling': latent_sampling, 'set': None, 'data_augmentation': [],
                                    'pretrained_features': getattr(pretrained_features, 'name', None),
                                    'pretrained_upsampler': getattr(pretrained_upsampler, 'name', None),
                                    'epochs': 0, 'batch_size': None, 'fine_tuning': []}
        self.testing = {0: {m: {'n': 0, 'epochs': 0, 'accuracy': 0} for m in self.predict_methods}}
        self.ood_results = {}
        self.optimizer = Optimizer(self.parameters(), **optimizer)
        if self.x_is_generated:
            self.optimizer.set_early_bucket(list(self.imager.parameters()) + list(self.decoder.parameters()))
        self.training_parameters['optimizer'] = self.optimizer.params
        self.train_history = {'epochs': 0}

        self.latent_dim = latent_dim
        self._latent_samplings = {'train': latent_sampling, 'eval': test_latent_sampling}
        self.latent_sampling = latent_sampling
        self.encoder_layer_sizes, self.decoder_layer_sizes, self.classifier_layer_sizes = encoder, decoder, classifier
        self.upsampler = upsampler
        self.activation = activation
        self.output_activation = output_activation
        self.z_output = False
        self._host_cache = {}
        self.eval()

    # ------------------------------------------------------------------------------------ module state
    def train(self, *a, **k):
        super().train(*a, **k)
        self.latent_sampling = self._latent_samplings['train' if self.training else 'eval']
        return self

    @property
    def latent_sampling(self):
        return self._latent_sampling

    @latent_sampling.setter
    def latent_sampling(self, v):
        self._latent_sampling = v
        self.encoder.sampling_size = v

    @property
    def device(self):
        return next(self.parameters()).device

    @device.setter
    def device(self, d):
        self.to(d)

    def to(self, d):
        super().to(d)
        self.optimizer.to(d)
        return self

    @property
    def nparams(self):
        return sum(p.nelement() for p in self.parameters())

    def set_distributed(self, world_size, process_group=None, seed_offset=True):
        """Data-parallel replica set-up (no counterpart in the single-process reference): rank 0's
        parameters, BatchNorm buffers and optimiser state are broadcast to every rank (replicas no longer rely on identical
        seeding), the gradient exchange is switched on, and - seed_offset - the device generator that draws the
        reparameterisation noise is re-seeded per rank so that ranks draw DIFFERENT epsilon (and dropout masks)."""
        import torch.distributed as dist
        world_size = int(world_size)
        if world_size > 1 and dist.is_available() and dist.is_initialized():
            with torch.no_grad():
                for t in self.state_dict().values():
                    if t.is_cuda and dist.get_backend(process_group) != 'nccl':
                        h = t.cpu()
                        dist.broadcast(h, 0, group=process_group)
                        t.copy_(h)
                    else:
                        dist.broadcast(t, 0, group=process_group)
            if seed_offset:
                rank = dist.get_rank(process_group)
                base = torch.initial_seed()
                if torch.cuda.is_available():
                    torch.cuda.manual_seed(base + 7919 * (rank + 1))
        self.optimizer.set_distributed(world_size, process_group)
        return self

    def set_sync_batchnorm(self, world_size, process_group=None):
        """Data-parallel option: BatchNorm statistics over ALL ranks, i.e. exactly what the single-process
        reference computes on the global batch (default: per-rank statistics, as DistributedDataParallel does)."""
        from module.vae_layers.conv import HipBatchNorm2d
        for m in self.modules():
            if isinstance(m, HipBatchNorm2d):
                m.sync_world, m.sync_group = int(world_size), process_group
        return self

    def set_compute_dtype(self, dtype):
        """'fp32' (the reference's arithmetic; default) or 'bf16' (config 5 of BASELINE.json): bf16 activations between the
        layers of the conv stacks and bf16 matrix-core convolutions from the fp32 master weights; BatchNorm statistics,
        the dense heads, the latent / loss math, gradients of parameters and Adam stay fp32."""
        from module.vae_layers.conv import HipConvStack
        if dtype not in ('fp32', 'bf16'):
            raise ValueError(dtype)
        if dtype == 'bf16' and self.activation == 'leaky':
            raise NotImplementedError("the bf16 mode has ReLU kernels only: activation='leaky' (config.ini:113) trains in fp32")
        for m in self.modules():
            if isinstance(m, HipConvStack):
                m.compute_dtype = dtype
        self.compute_dtype = dtype
        return self

    @property
    def max_batch_sizes(self):
        """The reference hard-wires {'train': 32, 'test': 32} (cvae.py:1145-1147) after a halving search
        (cvae.py:1087-1143).  Here the bounds follow from the tensors themselves (powers of two, as the search returns):
        train - the decoder batch (L+1)*N times the widest activation stays below 2^31 elements (one launch chain, every
        activation kept for backward: ~3 MB per image of config 2, 50 GB at the bound); test - the label-free evaluation
        decodes in slabs (`_decode`), so only the returned reconstruction (L_test+1, N, ...) and, for a class-conditional prior, the
        (L_test, C, N, K) latents scored under every class must stay below 2^31 elements."""
        def pow2_below(v):
            v = max(int(v), 1)
            return 1 << (v.bit_length() - 1)
        lim = (1 << 31) - 1
        if not self.x_is_generated:
            return {'train': 1 << 16, 'test': 1 << 16}
        wide = self._widest_decoder_activation()
        reco = int(np.prod(self._reco_shape()))
        train = pow2_below(lim // ((self._latent_samplings['train'] + 1) * wide))
        test = pow2_below(lim // ((self._latent_samplings['eval'] + 1) * reco))
        if self.encoder.prior.conditional:
            # the importance weights of the label-free evaluation score every draw under every class: (L, C, N, K) latents
            # go through the prior's Mahalanobis kernel as ONE tensor (cvae.py:793-873)
            test = min(test, pow2_below(lim // (max(self._latent_samplings['eval'], 1) * self.num_labels * self.latent_dim)))
        return {'train': min(train, 1 << 16), 'test': min(test, 1 << 16)}

    # ------------------------------------------------------------------------------------ forward
    def _features_of(self, x):
        if not self.features:
            return x
        lead = (1,) if x.dim() == self.input_dim else x.shape[:-self.input_dim]
        t = self.features(x.reshape(-1, *self.input_shape))
        return t.view(*lead, *self.encoder.input_shape)

    def forward(self, x, y=None, x_features=None, **kw):
        """x (N1..Ng, D1..Dt), y (N1..Ng) -> (x_reco (L+1, N.., D..), logits (L+1, N.., C)[, mu, log_var, z]...)."""
        if y is None and self.y_is_coded:
            raise ValueError('y is supposed to be an input of the net')
        lead = (1,) if x.dim() == self.input_dim else x.shape[:-self.input_dim]
        with self._constant_weights(x):          # a span of its own when called outside evaluate() (see _constant_weights)
            if x_features is None:
                x_features = self._features_of(x)
            return self.forward_from_features(x_features, None if y is None else y.view(*lead), x, **kw)

    def _dump_after_encoder_error(self, err, x, y):
        """cvae.py:476-488: a ValueError out of the encoder saves the model and the offending batch under
        log/dump-<job_number> and is re-raised.  (The reference's own NaN probe that used to raise it is commented out,
        layers.py:381-386; the contract is kept for encoders / priors that do raise.)"""
        where = os.path.join('log', 'dump-{}'.format(self.job_number))
        self.save(where)
        torch.save(x, os.path.join(where, 'x.pt'))
        torch.save(y, os.path.join(where, 'y.pt'))
        logging.error('Error %s, net dumped in %s', str(err), where)

    def _reco_shape(self):
        """Per-sample shape of the decoder output: the image, or (256, C, H, W) level logits for a categorical decoder."""
        return self.input_shape if self.output_distribution != 'categorical' else (256, *self.input_shape)

    # largest activation of one launch chain of the label-free evaluation, in elements: keeps every tensor on that path far
    # below 2^31 elements and the working set of the (L+1)*N decoder batch bounded (the reference bounds the same thing by
    # halving the evaluation batch until it fits, cvae.py:1087-1153)
    EVAL_SLAB_ELEMENTS = 1 << 28

    def _widest_decoder_activation(self):
        """Elements per image of the widest tensor between z and the reconstruction."""
        widths = [int(np.prod(self._reco_shape()))]
        for stack in (self.decoder, self.imager):
            for m in stack.modules():
                if isinstance(m, nn.Linear):
                    widths.append(m.out_features)
            for sh in getattr(stack, 'shapes', None) or []:
                widths.append(int(np.prod(sh)))
        return max(widths)

    def _eval_slab_rows(self):
        env = os.environ.get('JVAE_EVAL_SLAB_ROWS')                 # tests: force many small slabs
        if env:
            return max(int(env), 1)
        return max(self.EVAL_SLAB_ELEMENTS // self._widest_decoder_activation(), 1)

    def _decode_rows(self, z):
        u = self.decoder(z)
        return self.imager(u.reshape(-1, *self.imager.input_shape))

    def _decode(self, z):
        x_ = None
        if self.x_is_generated:
            rows = z.numel() // z.shape[-1]
            slab = self._eval_slab_rows()
            if rows > slab and not torch.is_grad_enabled() and not self.training:
                # evaluation (running-statistics BatchNorm: decoder rows are independent): slabs of rows through the
                # decoder, each written into its place of the one (rows, ...) reconstruction the caller returns
                zf = z.reshape(rows, z.shape[-1])
                x_ = torch.empty((rows, *self._reco_shape()), device=z.device, dtype=torch.float32)
                for r0 in range(0, rows, slab):
                    x_[r0:r0 + slab].copy_(self._decode_rows(zf[r0:r0 + slab]).view(-1, *self._reco_shape()))
            else:
                x_ = self._decode_rows(z)
        if self.classifier_type in ('linear', None):
            logits = self.classifier(z)
        else:                                   # 'softmax': logits from the dictionary itself (cvae.py:498-499)
            m = self.encoder.prior.mean
            logits = ops.linear(z, m, m.pow(2).sum(-1) / 2)
        return x_, logits

    def forward_from_features(self, x_features, y, x, z_output=True, sampling_epsilon_norm_out=False,
                              sigma_out=False, epsilon=None):
        lead = x_features.shape[:x_features.dim() - len(self.encoder.input_shape)]
        flat = x_features.reshape(*lead, -1)
        y1h = None if (y is None or not self.y_is_coded) else onehot_encoding(y, self.num_labels).float()
        try:
            mu, log_var, z, eps, sigma = self.encoder(flat, y1h, epsilon=epsilon)
        except ValueError as err:
            self._dump_after_encoder_error(err, x, y)
            raise
        x_, logits = self._decode(z)
        out = ((x,) if self.is_vib else (x_.view(self.latent_sampling + 1, *lead, *self._reco_shape()),)) + (logits,)   # cvae.py:504-508
        if z_output:
            out += (mu, log_var, z)
        if sampling_epsilon_norm_out:
            out += ((eps ** 2).sum(-1),)
        if sigma_out:
            out += (sigma,)
        return out

    # ------------------------------------------------------------------------------------ evaluate
    def evaluate(self, x, y=None, batch=0, current_measures=None, with_beta=False, kl_var_weighting=1.,
                 gamma_weighting=1, z_output=False, epsilon=None, **kw):
        """Forward + every per-sample loss term of one batch (training branch of cvae.py:523-917).

        Returns (x_reco (L+1,N,..), y_est (N,C), batch_losses {name: (N,) tensor}, total_measures {name: float}
        [, mu, log_var, z]).  `epsilon` (L+1,N,K) optionally injects the reparameterisation noise.
        All Python floats of `total_measures` come from ONE packed device read-back.

        The call opens a span in which the weights are constant (this forward and, in training, the backward that
        follows, until Optimizer.step()): the packed operand forms of all convolution weights are refreshed by ONE launch
        here instead of one launch in front of every convolution (jvae_hip/lib.py::pack_cache_begin)."""
        with self._constant_weights(x):
            return self._evaluate(x, y, batch, current_measures, with_beta, kl_var_weighting, gamma_weighting, z_output,
                                  epsilon, **kw)

    @contextlib.contextmanager
    def _constant_weights(self, x):
        """The span in which the convolution weights are vouched for (packed operand forms served from the step's cache).
        Opened by evaluate() and by a stand-alone forward(); closed when the call returns if no backward can follow (eval
        mode / no_grad), else by the backward pass itself (ops._close_span_after_backward), by Optimizer.zero_grad() /
        step(), and by everything that rewrites the weights wholesale (load_state_dict, load_weights, .to()/_apply).  A
        nested call (forward() inside evaluate()) belongs to the outer span.  A convolution FORWARD called outside any such call
        (a submodule invoked directly) disarms a cache left armed for a backward that has not come yet and packs per call
        (ops._Conv.forward): the weights may have changed through .data in between."""
        if getattr(self, '_span_open', False) or not x.is_cuda:     # CPU tensors: the ops raise their own "no CPU fallback" error
            yield
            return
        self._span_open = True
        _lib.span_depth += 1
        try:
            ws = getattr(self, '_conv_weights', None)
            if ws is None:
                from module.vae_layers.conv import HipConv2d, HipConvTranspose2d
                ws = self._conv_weights = [m.weight for m in self.modules() if isinstance(m, (HipConv2d, HipConvTranspose2d))]
            _lib.pack_cache_begin(ws, x.device)
            if torch.cuda.is_current_stream_capturing():
                _lib.pack_cache_pin()            # the graph bakes this owner's slot addresses in: never recycle its region
            yield
        finally:
            self._span_open = False
            _lib.span_depth -= 1
            if not (self.training and torch.is_grad_enabled()):
                _lib.pack_cache_end()            # no backward will follow: stop vouching for the weights now

    def _apply(self, fn, *a, **kw):
        _lib.pack_cache_end()                    # .to() / .float() / .cuda(): the weights move or change
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        _lib.pack_cache_end()
        if getattr(self, 'optimizer', None) is not None:
            self.optimizer._scan_pending = True          # loaded values: the next check_nonfinite() scans every parameter
        return super().load_state_dict(*a, **kw)

    def _evaluate(self, x, y=None, batch=0, current_measures=None, with_beta=False, kl_var_weighting=1.,
                  gamma_weighting=1, z_output=False, epsilon=None, **kw):
        if y is None or self.is_vib:
            if self.is_vib:
                return self._evaluate_vib(x, y, batch, current_measures, with_beta, kl_var_weighting, gamma_weighting,
                                          z_output, epsilon, kw.get('_raw_measures'))
            return self._evaluate_all_classes(x, batch, current_measures, with_beta, z_output, epsilon)
        if x.dim() != self.input_dim + 1:
            x = x.reshape(-1, *self.input_shape)
            y = y.reshape(-1)
        N = x.shape[0]
        L = self.latent_sampling
        D = int(np.prod(self.input_shape))
        cross_y_weight = False                                               # cvae.py:557-563
        if self.y_is_decoded:
            if self.is_cvae or self.is_vae:
                cross_y_weight = gamma_weighting * self.gamma if self.training else False
            else:
                cross_y_weight = gamma_weighting * self.gamma

        feats = self._features_of(x).reshape(N, -1)
        y1h = onehot_encoding(y, self.num_labels).float() if self.y_is_coded else None
        try:
            mu, log_var, z, eps, sigma_coded, terms = self.encoder.encode(feats, y1h, y, kl_var_weighting, epsilon)
        except ValueError as err:
            self._dump_after_encoder_error(err, x, y)
            raise
        if self.training and self.optimizer._world > 1 and z.requires_grad and not getattr(self, '_graph_capture', False) \
                and not getattr(self.optimizer, '_external_reduce', False):
            # data-parallel: the decoder's gradients are final once d(loss)/dz exists -> start their all-reduce there
            z.register_hook(self._early_reduce_hook)
        x_, logits = self._decode(z)
        x_reco = x_.view(L + 1, N, *self._reco_shape())

        s, s_kind, sigma_rms = self._sigma_operand(sigma_coded, N)
        ce = None
        if self.y_is_decoded:
            ce = x_loss(y, logits, batch_mean=False)                         # all L+1 rows, as cvae.py:738 does
        if self.output_distribution == 'categorical':
            # cvae.py:654-660,752-753: -log p(x|z) is the 256-level cross entropy of every pixel, summed over the image (the
            # only term with a gradient); `wmse` is the plain mean-square error of the arg-max image (reported, sigma not
            # applied: the `catgorical` test at cvae.py:638 never matches), `mse` = wmse * sigma^2 as for a gaussian decoder
            ndim = len(self.input_shape)
            ce_x = categorical_loss(x_reco[1:], x, ndim=ndim, batch_mean=False)                       # (L, N)
            with torch.no_grad():
                levels = x_reco[1:].argmax(-ndim - 1).float() / 255
                wmse = mse_loss(levels, x, ndim=ndim, batch_mean=False).mean(0)
                mse = None
            cross_x = ce_x.mean(0)
            total = cross_x + (self.beta if with_beta else 1.) * terms['kl']
            if cross_y_weight:
                total = total + cross_y_weight * ce
        else:
            wmse_s = ops.recon_wmse(x_reco, x, s, s_kind, snapshot=bool(self.training and self.sigma.decay))     # (L, N)
            wmse, cross_x, total, mse = ops.elbo(wmse_s, terms['kl'], ce if cross_y_weight else None, s, s_kind, D,
                                                 self.beta if with_beta else 1., float(cross_y_weight or 0.), with_mse=True)
        losses = {'kl': terms['kl'], 'zdist': terms['distance'], 'var_kl': terms['var_kl']}
        dictionary = self.encoder.prior.mean if self.encoder.prior.conditional else None
        if dictionary is not None:
            losses['dzdist'] = terms['dzdist']
        losses['wmse'] = wmse
        losses['cross_x'] = cross_x
        if ce is not None:
            losses['cross_y'] = ce
        losses['total'] = total

        prev = current_measures._dev if isinstance(current_measures, Measures) else self._upload_measures(
            current_measures, x.device) if (current_measures and batch) else None
        packed = self._pack_measures(x, wmse, terms, dictionary, prev, batch, mse=mse, sigma_rms=sigma_rms, sigma_t=s)
        if self.training:
            if self.sigma.decay and not self.sigma.learned:                  # the decay rule computes with the rmse now
                from jvae_hip import lib as _lib
                torch.cuda.current_stream(x.device).wait_stream(_lib.side_stream(x.device))
            self.sigma.update(rmse=packed[3])                                # device scalar, as in the reference
        if kw.get('_raw_measures'):
            # graph capture (graph_train_step): the 16-float device buffer itself; the host copy is made after each replay
            measures = (packed, dictionary is not None)
        else:
            measures = Measures(packed, dictionary is not None, _grad_nan_exit)
            if self.training:
                self.training_parameters['sigma'] = _LazySigmaParams(self.sigma, measures)
        out = (x_reco, _mean_over_draws(logits), losses, measures)
        if z_output:
            out += (mu, log_var, z)
        return out

    # ------------------------------------------------------------------------------------ type 'vib'
    def _evaluate_vib(self, x, y, batch, current_measures, with_beta, kl_var_weighting, gamma_weighting, z_output, epsilon,
                      raw_measures=False):
        """evaluate() of a model WITHOUT decoder (type 'vib': cvae.py:189,201,490-505,891-896): the loss is the classifier's
        cross entropy on z plus beta x the KL to the single prior; the returned `reconstruction` is x itself.  With labels:
        every loss (N,); without (the evaluation path): cross_y and total are (C, N) - one row per candidate class
        (module/losses.py:47-86) - and the prediction is the classifier's ('esty')."""
        if x.dim() != self.input_dim + 1:
            x = x.reshape(-1, *self.input_shape)
            y = None if y is None else y.reshape(-1)
        N = x.shape[0]
        cross_y_weight = gamma_weighting * self.gamma                                   # cvae.py:557-563: always in the loss
        with torch.set_grad_enabled(torch.is_grad_enabled() and y is not None):
            feats = self._features_of(x).reshape(N, -1)
            dummy = y if y is not None else torch.zeros(N, dtype=torch.int64, device=x.device)
            try:
                mu, log_var, z, eps, _, terms = self.encoder.encode(feats, None, dummy, kl_var_weighting, epsilon)
            except ValueError as err:
                self._dump_after_encoder_error(err, x, y)
                raise
            _, logits = self._decode(z)
            ce = x_loss(y, logits, batch_mean=False)                                     # (N,) or (C, N); all L+1 rows
            beta = self.beta if with_beta else 1.
            losses = {'kl': terms['kl'], 'zdist': terms['distance'], 'var_kl': terms['var_kl']}
            total = beta * terms['kl']
            if cross_y_weight:
                total = total + cross_y_weight * ce                                     # broadcasts to (C, N) without labels
            elif y is None:
                total = total.unsqueeze(0).expand_as(ce)
            losses['total'] = total
            losses['cross_y'] = ce
        prev = current_measures._dev if isinstance(current_measures, Measures) else self._upload_measures(
            current_measures, x.device) if (current_measures and batch) else None
        with torch.no_grad():
            packed = self._pack_measures_raw(x, torch.zeros(N, device=x.device), terms, None, prev, batch, self.sigma.detach(),
                                             int(self.sigma.is_log))
        keys = ('sigma', 'zdist', 'var_kl')
        measures = (packed, False) if raw_measures else Measures(packed, False, _grad_nan_exit, only=keys)
        out = (x, _mean_over_draws(logits), losses, measures)
        if z_output:
            out += (mu, log_var, z)
        return out

    # ------------------------------------------------------------------------------------ evaluation
    def _evaluate_all_classes(self, x, batch, current_measures, with_beta, z_output, epsilon):
        """evaluate(x) without labels (cvae.py:548-600, 793-873): every class is tried as the prior component.
        Losses kl / zdist / var_kl / total / iws [/ cross_y] are (C, N); wmse / cross_x / dzdist stay (N,).
        The heavy parts (conv stacks on (L+1)N latents, BatchNorm, latent / KL kernel on C*N rows, reconstruction,
        Mahalanobis distances of the L*C*N sampled latents, the importance-weight assembly) run on the HIP kernels."""
        if self.y_is_coded or self.is_jvae:
            # The reference cannot do it either: cvae.py:593-600 builds the (C, N) label grid and forward() (cvae.py:451)
            # then calls y.view(N) on it - "RuntimeError: shape '[N]' is invalid for input of size C*N" for conv and MLP
            # models alike (probed on the reference in the build container).  Same outcome, clearer message.
            raise NotImplementedError('evaluate(x) without labels is not possible for models with coded labels (the '
                                      'reference fails on it: cvae.py:451): pass y')
        if x.dim() != self.input_dim + 1:
            x = x.reshape(-1, *self.input_shape)
        N, C, K = x.shape[0], self.num_labels, self.latent_dim
        L = self.latent_sampling
        D = int(np.prod(self.input_shape))
        pr = self.encoder.prior
        with torch.no_grad():
            feats = self._features_of(x).reshape(N, -1)
            dummy = torch.zeros(N, dtype=torch.int64, device=x.device)
            mu, log_var, z, eps, sigma_coded, _ = self.encoder.encode(feats, None, dummy, 1., epsilon)
            x_, logits = self._decode(z)
            x_reco = x_.view(L + 1, N, *self._reco_shape())
            s, s_kind, sigma_rms = self._sigma_operand(sigma_coded, N)
            categorical = self.output_distribution == 'categorical'
            if categorical:
                # cvae.py:654-660,672-676: log p(x|z_l) = -(256-level pixel cross entropy); the kernels below take it as the
                # equivalent "weighted mse" of a unit-sigma gaussian: -D/2 (w + log 2 pi) = -ce  <=>  w = 2 ce / D - log 2 pi
                ndim = len(self.input_shape)
                ce_x = categorical_loss(x_reco[1:], x, ndim=ndim, batch_mean=False)          # (L, N)
                levels = x_reco[1:].argmax(-ndim - 1).float() / 255
                wmse_cat = mse_loss(levels, x, ndim=ndim, batch_mean=False).mean(0)
                wmse_s = 2. * ce_x / D - LOG2PI
                s_report, s, s_kind = s, torch.ones(1, device=x.device), ops.SIGMA_VALUE
            else:
                wmse_s = ops.recon_wmse(x_reco, x, s, s_kind)                             # (L, N)
            y_all = torch.arange(C, device=x.device).unsqueeze(1).expand(C, N)
            kd = pr.kl(mu, log_var, y=y_all if pr.conditional else None)                  # (C, N) each
            zero_kl = torch.zeros(N, device=x.device)
            wmse, cross_x, _, mse = ops.elbo(wmse_s, zero_kl, None, s, s_kind, D, 1., 0., with_mse=True)
            if categorical:
                wmse, mse = wmse_cat, None             # what the reference reports: the arg-max image's mean-square error
            losses = {'kl': kd['kl'], 'zdist': kd['distance'], 'var_kl': kd['var_kl']}
            dictionary = pr.mean if pr.conditional else None
            terms = {'distance': kd['distance'].reshape(-1), 'var_kl': kd['var_kl'].reshape(-1)}
            if dictionary is not None:
                _, _, _, _, _, dz = ops.latent(mu, log_var, torch.zeros((1, N, K), device=x.device), dummy,
                                               dictionary, pr._var_parameter, var_dim=pr.var_dim, sampled=False)
                losses['dzdist'] = dz
            losses['wmse'], losses['cross_x'] = wmse, cross_x
            if self.y_is_decoded:
                losses['cross_y'] = x_loss(None, logits, batch_mean=False)                # (C, N)
            beta = self.beta if with_beta else 1.
            # one prior component per class: (C, N); a single prior (type 'vae'): (N,)
            losses['total'] = (cross_x.unsqueeze(0) if pr.conditional else cross_x) + beta * kd['kl']
            if self.y_is_decoded and not (self.is_cvae or self.is_vae) and self.gamma:
                losses['total'] = losses['total'] + self.gamma * losses['cross_y']        # cvae.py:557-563,886-889
            # importance-weighted bound: log p(x|z_l) + log p(z_l|y) - log q(z_l|x), cvae.py:672-676,793-873
            z_s = z[1:]
            if pr.conditional:
                z_y = z_s.unsqueeze(1).expand(L, C, N, K)
                y_s = y_all.unsqueeze(0).expand(L, C, N)
                log_pz = pr.log_density(z_y, y_s)                                         # (L, C, N)
            else:
                log_pz = pr.log_density(z_s, None)                                        # (L, N)
            # rows + max / mean-exp fold over the L samples in one kernel pair (jvae_iws_f32)
            losses['iws'] = ops.iws(wmse_s, eps, log_var, log_pz, s, s_kind, D)
            prev = current_measures._dev if isinstance(current_measures, Measures) else None
            packed = self._pack_measures(x, wmse, terms, dictionary, prev, batch, mse=mse, sigma_rms=sigma_rms,
                                         sigma_t=s_report if categorical else s)
        measures = Measures(packed, dictionary is not None, _grad_nan_exit)
        out = (x_reco, _mean_over_draws(logits), losses, measures)
        if z_output:
            out += (mu, log_var, z)
        return out

    def predict_after_evaluate(self, logits, losses, method='default'):
        """Class prediction from the all-class losses (cvae.py:938-970)."""
        if method == 'default':
            method = self.predict_methods[0]
        if method is None:
            return logits.softmax(-1)
        table = {'iws': lambda: losses['iws'].argmax(0), 'closest': lambda: losses['zdist'].argmin(0),
                 'loss': lambda: losses['total'].argmin(0), 'esty': lambda: logits.argmax(-1),
                 'mean': lambda: logits.softmax(-1).mean(0).argmax(-1), 'already': lambda: losses['y_est_already']}
        if method not in table:
            raise ValueError(f'Unknown method {method}')
        return table[method]()

    def predict(self, x, method=None, **kw):
        _, logits, losses, _ = self.evaluate(x)
        return self.predict_after_evaluate(logits, losses, method=method or 'default')

    def batch_dist_measures(self, logits, losses, methods, to_cpu=False):
        """OOD scores per sample (higher = more in-distribution) for the cvae methods (cvae.py:972-1085); the
        '-2s' / '-a-x-y' suffixes only name the thresholding done downstream."""
        C = self.num_labels
        out = {}
        for name in methods:
            m = name[:-3] if name.endswith('-2s') else name
            m = m.split('-')[0] if '-a-' in m else m
            per_class = self.losses_might_be_computed_for_each_class          # cvae.py:996-1016,1036-1039
            if m in ('elbo', 'max'):
                v = (-losses['total']).max(0)[0] if (per_class or m == 'max') else -losses['total']
            elif m == 'iws' and not per_class:
                v = losses['iws']
            elif m == 'iws':
                top = losses['iws'].max(0)[0]
                v = (losses['iws'] - top).exp().sum(0).log() + top + math.log(C)
            elif m in ('soft', 'softkl'):
                v = (-losses['kl']).softmax(0).max(0)[0]
            elif m.startswith('softkl-'):
                v = (-losses['kl'] / float(m[7:])).softmax(0).max(0)[0]
            elif m in ('zdist', 'kl'):
                v = (-losses[m]).max(0)[0] if not self.is_vae else -losses[m]
            elif m == 'mse':
                v = -losses['cross_x']
            elif m == 'wmse':
                v = -losses['wmse']
            elif m == 'logits':
                v = logits.max(-1)[0]
            elif m.startswith('baseline'):
                T = float(m.split('-')[-1]) if '-' in m else 1.
                v = (logits / T).softmax(-1).max(-1)[0]
            else:
                raise NotImplementedError(f'{name}: OOD method outside this build')
            out[name] = v.cpu() if to_cpu else v
        return out

    def accuracy(self, testset=None, batch_size=100, num_batch='all', method='all', print_result=False,
                 update_self_testing=True, outputs=None, sample_dirs=[], recorder=None, epoch='last', from_where='all',
                 epoch_tolerance=0, log=True):
        """Classification accuracy of `testset` (any map-style dataset of (x, label)) per prediction method, with the
        reference's signature and bookkeeping (cvae.py:1187-1452): every batch goes through the label-free evaluation
        (all-class losses, `iws`: SURVEY.md §8f-1); with a `LossRecorder` the per-sample losses, `logits.T` and the labels are
        recorded batch by batch and written as `record-<set>.pth` into `sample_dirs` (the files test.py / results/ of the
        reference read, §8f-3) - or, if the recorder already holds the batches, the losses are RECOVERED from it instead of
        being computed.  Named torchvision datasets and the registry lookup of earlier results (`from_where`) are host-side
        plumbing outside this build: pass the dataset."""
        if testset is None or isinstance(testset, str):
            raise NotImplementedError('named torchvision datasets are outside this build: pass a torch.utils.data.Dataset')
        name = getattr(testset, 'name', 'testset')
        only_one = isinstance(method, str) and method != 'all'
        methods = list(self.predict_methods) if method == 'all' else ([method] if only_one else list(method))
        full = int(np.ceil(len(testset) / batch_size))
        shuffle = not (num_batch == 'all' or num_batch >= full)
        num_batch = full if not shuffle else int(num_batch)
        if epoch == 'last':
            epoch = self.trained
        recorded = recorder is not None and len(recorder) >= num_batch
        recording = recorder is not None and not recorded
        if recorded:
            num_batch, batch_size = len(recorder), recorder.batch_size
        if recording:
            recorder.reset()
            recorder.num_batch = num_batch
        if recorder is not None:
            recorder.init_seed_for_dataloader()
        device = self.device
        was_training = self.training
        self.eval()
        loader = iter(torch.utils.data.DataLoader(testset, batch_size=batch_size, num_workers=0, shuffle=shuffle))
        errors = torch.zeros(len(methods), device=device)
        sums, n, measures, t0 = {}, 0, None, time.time()
        with torch.no_grad():
            for i in range(num_batch):
                if recorded:
                    keys = [k for k in recorder.keys() if k in self.loss_components]
                    losses = recorder.get_batch(i, *keys, force_dict=True)
                    logits = recorder.get_batch(i, 'logits').T
                    y = recorder.get_batch(i, 'y_true')
                else:
                    x, y = next(loader)[:2]
                    x, y = self._device_batch(x.to(device)), y.to(device)      # raw uint8 images: ToTensor on the device
                    _, logits, losses, measures = self.evaluate(x, batch=i, current_measures=measures)
                preds = [self.predict_after_evaluate(logits, losses, method=m) for m in methods]
                if recording:
                    recorder.append_batch(**losses, y_true=y, logits=logits.T)
                errors += torch.stack([(p != y).sum() for p in preds]).float()
                for k, v in losses.items():                     # loss of the TRUE class where a loss is per class (C, N)
                    v = v.gather(0, y.unsqueeze(0))[0] if v.dim() == 2 else v
                    sums[k] = sums.get(k, 0.) + v.float().mean()
                n += y.numel()
                if print_result and outputs is not None and hasattr(outputs, 'results'):
                    acc_now = (1 - errors / n).tolist()
                    outputs.results(i, num_batch, 0, 0, losses={k: float(sums[k]) / (i + 1) for k in self.loss_components if k in sums},
                                    metrics={k: (measures or {}).get(k, np.nan) for k in self.metrics},
                                    accuracy=dict(zip(methods, acc_now)), time_per_i=(time.time() - t0) / (i + 1),
                                    batch_size=batch_size, preambule=print_result)
        acc = dict(zip(methods, (1 - errors / max(n, 1)).tolist()))
        self.test_losses = {k: float(v) / max(num_batch, 1) for k, v in sums.items()}
        if measures:
            self.test_measures = dict(measures)
        if recorder is not None:
            recorder.restore_seed()
        if recording:
            for d in sample_dirs:
                os.makedirs(d, exist_ok=True)
                recorder.save(os.path.join(d, 'record-{}.pth'.format(name)))
        if update_self_testing:
            for m in methods:
                if n > self.testing.get(epoch, {}).get(m, {'n': 0})['n']:
                    self.testing.setdefault(epoch, {})[m] = {'n': n, 'epochs': epoch,
                                                             'sampling': self._latent_samplings['eval'], 'accuracy': acc[m]}
        if was_training:
            self.train()
        return acc[methods[0]] if only_one else acc

    def _early_reduce_hook(self, grad):
        self.optimizer.reduce_early_bucket()
        return grad

    def _upload_measures(self, current, device):
        """Running means handed in as plain floats (first batch of a resumed loop, a caller's own dict)."""
        t = torch.zeros(16)
        for i, k in ((10, 'xpow'), (11, 'mse'), (14, 'zdist'), (15, 'var_kl')):
            t[i] = float(current.get(k, 0.))
        return t.to(device, non_blocking=True)

    def _sigma_operand(self, sigma_coded, N):
        """What the loss kernels get as `sigma` (cvae.py:626-646): (tensor, kind, rms) - rms is the device scalar reported
        as the `sigma` measure for the coded / rmse kinds (the parameter's value BEFORE this batch updates it: cvae.py:624),
        None otherwise (the measures kernel derives it from the parameter).  A coded sigma is also stored into the
        parameter (batch mean of the coded log sigma: Sigma.update(v=...), cvae.py:631-634)."""
        sg = self.sigma
        if not (sg.coded or sg.is_rmse):
            return sg, (ops.SIGMA_LOG if sg.is_log else ops.SIGMA_VALUE), None
        with torch.no_grad():
            d = sg.data
            rms = ((2 * d).exp() if sg.is_log else d * d).mean().sqrt().reshape(1)
        if sg.is_rmse:
            return sg, ops.SIGMA_RMSE, rms
        per_sample = sigma_coded.reshape(-1, *sg.output_dim)
        sg.update(v=per_sample.detach())
        return per_sample.reshape(N), ops.SIGMA_CODED, rms

    def _pack_measures(self, x, wmse, terms, dictionary, prev, batch, mse=None, sigma_rms=None, sigma_t=None):
        """Every scalar evaluate() reports, computed by one kernel into one 16-float device buffer."""
        if sigma_rms is not None:
            if self.sigma.coded:
                # sic (cvae.py:668): wmse (N,) times sigma^2 (N,1,1,1) broadcasts to (N,1,1,N) in the reference; the mean of
                # that - the only use of `mse` with a coded sigma - is mean(wmse) * mean(sigma^2), reproduced here
                with torch.no_grad():
                    mse = (wmse.detach().mean() * (2 * sigma_t.detach()).exp().mean()).expand(wmse.numel()).contiguous()
            return self._pack_measures_raw(x, mse.detach(), terms, dictionary, prev, batch, sigma_rms, 2)
        return self._pack_measures_raw(x, wmse.detach(), terms, dictionary, prev, batch, self.sigma.detach(),
                                       int(self.sigma.is_log))

    def _pack_measures_raw(self, x, wmse, terms, dictionary, prev, batch, sigma_t, sigma_kind):
        from jvae_hip import lib as _lib
        with torch.no_grad():
            if getattr(self, '_scratch', None) is None or self._scratch.device != x.device:
                self._scratch = torch.zeros(1, device=x.device, dtype=torch.float32)
            # logging only: off the critical path -> side stream (the C x C dictionary diagnostics take ~0.2 ms at C = 100)
            main, side = torch.cuda.current_stream(x.device), _lib.side_stream(x.device)
            side.wait_stream(main)
            args = (x, wmse, terms['distance'].detach(), terms['var_kl'].detach(), sigma_t)
            with torch.cuda.stream(side):
                packed = ops.measures(*args, sigma_kind, None if dictionary is None else dictionary.detach(),
                                      self.optimizer.nonfinite_flag(), self._scratch, prev, batch)
            for t in args:
                t.record_stream(side)
            return packed

    # ------------------------------------------------------------------------------------ training loop
    def _mean_backward(self, tot):
        """tot.mean().backward() without the reduction kernels of a mean nobody reads and the expand of its backward:
        d(mean)/d(tot_i) = 1/numel, handed to autograd as a cached constant (the same fp32 value as torch's own)."""
        key = (tot.shape, tot.device)
        if getattr(self, '_mean_grad_key', None) != key:
            self._mean_grad = torch.ones_like(tot.detach()) / tot.numel()
            self._mean_grad_key = key
        torch.autograd.backward(tot, grad_tensors=self._mean_grad)

    def train_step(self, x, y, batch=0, current_measures=None, kl_var_weighting=1., gamma_weighting=1., epsilon=None):
        """One iteration of the reference's hot loop (cvae.py:2429-2461): zero_grad, evaluate, backward, clip, step."""
        self.optimizer.zero_grad()
        _, y_est, losses, measures = self.evaluate(x, y, batch=batch, with_beta=True,
                                                   kl_var_weighting=kl_var_weighting,
                                                   gamma_weighting=gamma_weighting,
                                                   current_measures=current_measures, epsilon=epsilon)
        # total.mean().backward() without the two reduction kernels of the mean nobody reads and the expand of its
        # backward: d(mean)/d(total_i) = 1/N, handed to autograd as a cached constant (same fp32 value as torch's own)
        if self.optimizer.check_nonfinite():     # cvae.py:2454-2457: a NaN / Inf parameter ends the run BEFORE backward
            _grad_nan_exit()
        self._mean_backward(losses['total'])
        self.optimizer.clip(self.parameters())
        self.optimizer.step()
        return losses, measures

    def graph_train_step(self, x, y, kl_var_weighting=1., gamma_weighting=1., warmup=3):
        """Capture one whole training step (zero_grad, evaluate, backward, clip, Adam - ~250 kernel launches on two
        streams) for batches shaped like (x, y) into a HIP graph; returns `step(x, y) -> (losses, measures)` that copies
        the batch into the captured buffers and replays the graph: one host call per step instead of ~4 ms of Python
        enqueue work.  No counterpart in the reference (its loop is eager).  Differences from train_step(): epsilon is
        drawn inside the graph (the generator's offset advances per replay), `measures` are those of the current batch
        only (batch index 0), the `losses` tensors are the graph's own buffers (overwritten by the next replay), the
        warm-up weights are fixed (kl / gamma weights of the call).  Data parallel (optimizer.set_distributed): see below -
        two captured halves with the gradient all-reduce between them.  Drop every reference to the outputs of earlier EAGER
        steps (losses, measures) before calling this: a live eager autograd graph keeps its gradient-accumulation nodes bound
        to the default stream, which a capture must not touch."""
        world = getattr(self.optimizer, '_world', 1)
        dev = x.device
        self.optimizer.enable_device_hyper(True)
        sx, sy = x.clone(), y.clone()
        from jvae_hip import lib as _lib

        def fwd_bwd():
            self.optimizer.zero_grad()
            _, _, losses, raw = self.evaluate(sx, sy, batch=0, with_beta=True, kl_var_weighting=kl_var_weighting,
                                              gamma_weighting=gamma_weighting, _raw_measures=True)
            self._mean_backward(losses['total'])    # (the constant is created by the eager warm-up passes, not captured)
            return losses, raw

        def update():
            self.optimizer.clip(self.parameters())
            self.optimizer.step()

        def body():
            out = fwd_bwd()
            update()
            return out

        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):                      # eager warm-up on a side stream (allocations, lazy initialisations)
            for _ in range(max(int(warmup), 1)):
                body()
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        if world > 1:
            # Data parallel: the step is captured in TWO graphs - [zero_grad, forward, backward] and [clip, Adam] - with the
            # gradient exchange (ONE all-reduce of the flat buffer, any backend) issued eagerly between them: three host
            # calls per step.  The early-bucket overlap of the eager path is given up (the exchange is ~0.1 ms of a ~4 ms step).
            self._graph_capture = True                  # evaluate() must not launch the early-bucket hook while capturing
            self.optimizer._external_reduce = True
            try:
                ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(ga):
                    losses, (packed, has_dict) = fwd_bwd()
                    _lib.join_side_stream()             # weight gradients of the side stream: inside the captured half
                with torch.cuda.graph(gb, pool=ga.pool()):
                    update()
            finally:
                # both flags exist only while the two halves are being captured: a later EAGER train_step() / train_model()
                # on this model must again join the side stream and exchange its gradients inside reduce_gradients()
                self._graph_capture = False
                self.optimizer._external_reduce = False
            for g in self.optimizer._groups:
                g.step -= 1

            def step(xb, yb):
                sx.copy_(xb, non_blocking=True)
                sy.copy_(yb, non_blocking=True)
                ga.replay()
                self.optimizer.all_reduce_flat()
                gb.replay()
                self.optimizer.note_replayed_step()
                return losses, Measures(packed, has_dict, _grad_nan_exit, from_main=True)

            step.graph = (ga, gb)
            step.constants = (self._mean_grad,)         # the captured graphs read it: alive as long as the step is
            return step
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            losses, (packed, has_dict) = body()
        for g in self.optimizer._groups:                # the capture pass advanced the host counter without executing
            g.step -= 1

        def step(xb, yb):
            sx.copy_(xb, non_blocking=True)
            sy.copy_(yb, non_blocking=True)
            graph.replay()
            self.optimizer.note_replayed_step()
            return losses, Measures(packed, has_dict, _grad_nan_exit, from_main=True)

        step.graph = graph
        step.constants = (self._mean_grad,)             # the captured graph reads it: alive as long as the step is
        return step

    def train_model(self, trainset=None, transformer=None, data_augmentation=None, optimizer=None, epochs=50,
                    batch_size=100, test_batch_size=100, validation=4096, device=None, testset=None, oodsets=None,
                    acc_methods=None, fine_tuning=False, warmup=[0, 0], warmup_gamma=[0, 0], latent_sampling=None,
                    validation_sample_size=1024, full_test_every=10, ood_detection_every=10, train_accuracy=False,
                    save_dir=None, outputs=None, signal_handler=None, report_every=10):
        """Training loop with the reference's signature, bookkeeping and phases (cvae.py:2081-2547).

        `trainset` is a map-style dataset of (image, int label) - what utils/torch_load.get_dataset() returns (train.py:236-243
        hands it over with `.name`, `.transformer` and, for the torchvision sets, the raw images in `.data` / `.targets`), or any
        torch Dataset: float tensors in [0,1] shaped like `input_shape`, or RAW uint8 images ((H,W,C) or (C,H,W)), which are
        converted - and augmented - on the device.  Named datasets (a string) are the reference's torchvision plumbing and
        outside this build.

        What the reference does and this does too (same order):
        * `training_parameters` gets `epochs`, and - for an untrained net only - `set` (trainset.name), `transformer`
          (trainset.transformer), `validation`, `full_test_every`, `batch_size`, `latent_sampling`, `data_augmentation`
          (cvae.py:2108-2145: train.py:224-229 reads them back on --resume), then `validation_split_seed`, `warmup`,
          `warmup_gamma` (element-wise max with the recorded ones, cvae.py:2196-2202);
        * a seeded `random_split` holds `validation` samples out; training runs on the remainder (cvae.py:2164-2167);
        * every epoch (and once more at epoch == epochs) starts with the test phase: `accuracy(testset)` every
          `full_test_every` epochs -> history `test_accuracy / test_measures / test_loss`, `accuracy(validationset)` every
          epoch -> `validation_accuracy / _measures / _loss`, with `record-<set>.pth` written under `save_dir/samples/{last,
          <epoch>}` (cvae.py:2293-2382); `train_accuracy=True` adds `accuracy(trainset)` -> `train_accuracy`;
        * the hot loop cvae.py:2424-2501 = train_step(); the final test pass of cvae.py:2528-2545.
        OOD detection rates (`oodsets`, cvae.py:2328-2336,2513-2526: ROC / FPR tooling over the recorders) are out of scope
        (SURVEY.md §2a): a warning says so once and the phase is skipped.

        data_augmentation: the reference hands the list to its dataset factory, which prepends RandomHorizontalFlip ('flip')
        and RandomCrop(size, padding=size//8 [0 for imagenet sets], padding_mode='edge') ('crop') to the training transforms
        (utils/torch_load.py:405-426) - it re-opens the dataset BY NAME for that (cvae.py:2160-2162), so train.py passes the
        un-augmented float (ToTensor) dataset together with the list.  Here the two transforms run on the device, one launch
        per batch (ops.augment_batch: flip, edge-pad crop, /255, bit-exact against the torchvision chain), on the raw uint8
        images: those the dataset yields, or - for a float dataset - those it CARRIES (`.data` uint8 + `.targets` / `.labels`,
        as torchvision's CIFAR / SVHN objects do), once two samples have shown that `dataset[i]` is exactly `data[i] / 255`
        (i.e. its own transform is ToTensor and nothing else).  A float dataset without raw images, or a token other than
        'flip' / 'crop', raises instead of training silently un-augmented.  `outputs.results` gets the running batch-mean
        losses the reference prints (cvae.py:2463-2479), refreshed from the device every `report_every` batches (extra
        keyword) so that the loop does not synchronise per batch.  The batch size is clamped to `max_batch_sizes['train']`."""
        if isinstance(trainset, str):
            raise NotImplementedError('named torchvision datasets are host-side plumbing outside this build: '
                                      'pass a torch.utils.data.Dataset')
        if epochs:
            self.training_parameters['epochs'] = epochs
        set_name = None
        if trainset is not None:                 # cvae.py:2111-2118
            set_name = getattr(trainset, 'name', None)
            if not set_name:
                set_name = str(trainset).splitlines()[0].split()[-1].lower()
            transformer = getattr(trainset, 'transformer', transformer)
        if self.trained:                         # cvae.py:2120-2122: a partially trained net keeps its recorded parameters
            logging.info('Network partially trained (%d epochs)', self.trained)
        else:
            if trainset is not None:
                self.training_parameters.update({'set': set_name, 'transformer': transformer, 'validation': validation,
                                                 'full_test_every': full_test_every})
            if batch_size:
                self.training_parameters['batch_size'] = batch_size
            if latent_sampling:
                self._latent_samplings['train'] = latent_sampling
                self.training_parameters['latent_sampling'] = latent_sampling
            if data_augmentation:
                self.training_parameters['data_augmentation'] = list(data_augmentation)
        if not self.training_parameters.get('set'):
            if trainset is None:
                raise AssertionError("training_parameters['set'] is empty and no trainset was given (cvae.py:2147)")
            self.training_parameters['set'] = set_name        # a resumed job whose record lacks the key (written before round 5)
        set_name = str(self.training_parameters['set'])
        if trainset is None:
            raise NotImplementedError('re-opening the recorded set {!r} by name is torchvision plumbing outside this build: '
                                      'pass the dataset (train.py:224-226 does)'.format(set_name))
        data_augmentation = list(self.training_parameters.get('data_augmentation') or [])
        full_test_every = self.training_parameters.get('full_test_every', 10)
        unknown = [t for t in data_augmentation if t not in ('flip', 'crop')]
        if unknown:
            raise ValueError('data_augmentation: only flip and crop exist (utils/torch_load.py:405-413), got {}'.format(unknown))
        # cvae.py:2155-2158 asks for the key 'validation_split_seed ' (trailing blank), which never exists: the reference draws
        # a new seed in EVERY call, a resumed one included.  Same here.
        np.random.seed()
        seed = int(np.random.randint(0, 2 ** 12))
        self.training_parameters['validation_split_seed'] = seed
        if validation >= len(trainset):          # the reference's random_split([v, n - v]) would leave an EMPTY training set
            raise ValueError('validation={} leaves nothing of the {} training samples to train on'.format(validation, len(trainset)))
        validationset, trainset = torch.utils.data.random_split(trainset, [validation, len(trainset) - validation],
                                                                generator=torch.Generator().manual_seed(seed))
        validationset.name = 'validation'
        validation_sample_size = min(validation, validation_sample_size)
        device = device or self.device
        optimizer = optimizer or self.optimizer
        max_batch_sizes = self.max_batch_sizes
        test_batch_size = min(max_batch_sizes['test'], test_batch_size)
        if batch_size:                           # cvae.py:2180-2194
            train_batch_size = min(batch_size, max_batch_sizes['train'])
        else:
            train_batch_size = max_batch_sizes['train']
        logging.info('Train batch size is {}'.format(train_batch_size))
        batch_size = train_batch_size
        self.training_parameters['batch_size'] = batch_size
        warmup, warmup_gamma = list(warmup), list(warmup_gamma)
        warmup_ = self.training_parameters.get('warmup', [0, 0])
        warmup_gamma_ = self.training_parameters.get('warmup_gamma', [0, 0])
        for _ in (0, 1):                         # cvae.py:2196-2202
            warmup[_] = max(warmup[_], warmup_[_])
            warmup_gamma[_] = max(warmup_gamma[_], warmup_gamma_[_])
        self.training_parameters['warmup'] = warmup
        self.training_parameters['warmup_gamma'] = warmup_gamma

        from jvae_compat.recorders import LossRecorder
        sets = [set_name] + (['validation'] if validation else [])
        recorders = {s: LossRecorder(test_batch_size) for s in sets}      # tensors allocated by the first recorded batch
        if oodsets:
            logging.warning('OOD detection rates (%s) are outside this build (SURVEY.md 2a): phase skipped; the recorders of '
                            'accuracy() hold the losses the reference computes them from',
                            ','.join(str(getattr(s, 'name', s)) for s in oodsets))
        batches = self._train_batches(trainset, batch_size, data_augmentation, set_name, device)
        per_epoch = len(batches)
        done_epochs = self.train_history['epochs']
        if done_epochs == 0:
            self.train_history = {'epochs': 0}
        if not acc_methods:
            acc_methods = self.predict_methods
        if fine_tuning:
            for p in self.parameters():
                p.requires_grad_(True)
        sig = signal_handler

        def signalled(level):
            return sig is not None and getattr(sig, 'sig', 0) > level
        epoch = done_epochs
        for epoch in range(done_epochs, epochs + 1):
            self.train_history[epoch] = {}
            history_checkpoint = self.train_history[epoch]
            for s in recorders:
                recorders[s].reset()
            # ---- test phase (cvae.py:2302-2382)
            full_test = bool((epoch - done_epochs) and epoch % full_test_every == 0) or epoch == epochs
            if (full_test or not epoch) and save_dir:
                sample_dirs = [os.path.join(save_dir, 'samples', d) for d in ('last', '{:04d}'.format(epoch))]
                for d in sample_dirs:
                    os.makedirs(d, exist_ok=True)
            else:
                sample_dirs = []
            with torch.no_grad():
                self.test_losses, self.test_measures = {}, {}
                if full_test and testset is not None:
                    test_accuracy = self.accuracy(testset, batch_size=test_batch_size, num_batch='all', method=acc_methods,
                                                  outputs=outputs, sample_dirs=sample_dirs, update_self_testing=full_test,
                                                  recorder=recorders[set_name], print_result='TEST' if full_test else 'test')
                    history_checkpoint['test_accuracy'] = test_accuracy
                    history_checkpoint['test_measures'] = dict(self.test_measures)
                    history_checkpoint['test_loss'] = dict(self.test_losses)
                if validation:
                    validation_accuracy = self.accuracy(validationset, batch_size=test_batch_size, num_batch='all',
                                                        method=acc_methods, outputs=outputs, sample_dirs=sample_dirs,
                                                        update_self_testing=False, recorder=recorders['validation'],
                                                        print_result='VALID' if full_test else 'valid')
                    history_checkpoint['validation_accuracy'] = validation_accuracy
                    history_checkpoint['validation_measures'] = dict(self.test_measures)
                    history_checkpoint['validation_loss'] = dict(self.test_losses)
                if signalled(3):
                    logging.warning('Abruptly breaking training loop bc of %s', sig)
                    break
                if save_dir:
                    self.save(save_dir)
            if epoch == epochs:
                break
            # ---- train phase (cvae.py:2388-2501)
            if train_accuracy:
                with torch.no_grad():
                    train_accuracy = self.accuracy(trainset, batch_size=test_batch_size, num_batch='all', method=acc_methods,
                                                   update_self_testing=False, log=False, outputs=outputs, print_result='acc')
            if signalled(3):
                logging.warning('Abruptly breaking training loop bc of %s', sig)
                break
            if save_dir:
                self.save(save_dir)
            if signalled(2) or (full_test and signalled(1)):
                logging.warning('Breaking training loop bc of signal %s after %d epochs', sig, epoch)
                break
            self.encoder.prior.thaw_means(epoch)
            self.train()
            w_kl = max(0., min(1., (epoch + 1 - warmup[0]) / (warmup[1] + 1)))
            w_gamma = max(0., min(1., (epoch + 1 - warmup_gamma[0]) / (warmup_gamma[1] + 1)))
            t0 = time.time()
            measures, nb = None, 0
            keys, acc = None, None          # running sums of the batch means of every loss: ONE device vector
            shown = {}
            for i, (x, y) in enumerate(batches):
                losses, measures = self.train_step(x, y, batch=i, current_measures=measures,
                                                   kl_var_weighting=w_kl, gamma_weighting=w_gamma)
                if keys is None:
                    keys = list(losses)
                    acc = torch.zeros(len(keys), device=x.device)
                # one small kernel per batch (stack of means) instead of the reference's .item() per loss (cvae.py:2463-2469)
                acc += torch.stack([losses[k].detach().mean() for k in keys])
                nb = i + 1
                if outputs is not None and hasattr(outputs, 'results'):
                    if i % report_every == 0 or i + 1 == per_epoch:
                        host = (acc / nb).tolist()                    # the only read-back of the loop: every k batches
                        shown = dict(zip(keys, host))
                    outputs.results(i, per_epoch, epoch + 1, epochs, preambule='train',
                                    losses={k: shown.get(k, float('nan')) for k in self.loss_components},
                                    metrics={k: measures[k] for k in self.metrics} if (i % report_every == 0 or i + 1 == per_epoch)
                                    else {k: float('nan') for k in self.metrics},
                                    accuracy={k: np.nan for k in self.predict_methods},
                                    time_per_i=(time.time() - t0) / (i + 1), batch_size=batch_size, end_of_epoch='\n')
            self.eval()
            mean_loss = dict(zip(keys or [], (acc / max(nb, 1)).tolist() if acc is not None else []))
            if train_accuracy:
                history_checkpoint['train_accuracy'] = train_accuracy
            history_checkpoint['train_loss'] = mean_loss
            history_checkpoint['train_measures'] = dict(measures or {})
            self.train_history['epochs'] += 1
            history_checkpoint['lr'] = self.optimizer.lr
            self.trained += 1
            if fine_tuning:
                self.training_parameters['fine_tuning'].append(epoch)
            optimizer.update_lr()
            if signalled(3):
                logging.warning('Abruptly breaking training loop bc of %s', sig)
                break
            if save_dir:
                self.save(save_dir)
        # ---- after the loop (cvae.py:2503-2545): a last test pass over the whole test set, then save
        for s in recorders:
            recorders[s].reset()
        sample_dirs = []
        if save_dir:
            sample_dirs = [os.path.join(save_dir, 'samples', d) for d in ('last', '{:04d}'.format(epoch + 1))]
            for d in sample_dirs:
                os.makedirs(d, exist_ok=True)
        if testset is not None and not signalled(1):
            with torch.no_grad():
                self.accuracy(testset, batch_size=test_batch_size, method=acc_methods, recorder=recorders[set_name],
                              sample_dirs=sample_dirs, outputs=outputs, print_result='TEST')
        if signalled(3):
            logging.warning('Skipping saving because of %s', sig)
        elif save_dir:
            self.save(save_dir)
        return self.train_history

    def _train_batches(self, trainset, batch_size, data_augmentation, set_name, device):
        """The training loader of cvae.py:2245-2249 (batch_size, shuffle=True, num_workers=0, no drop_last) as a re-iterable of
        DEVICE batches (x float32 (n, *input_shape), y int64).  Three sources:
        * a dataset that yields uint8 images: collated by the DataLoader, converted / augmented by the input-pipeline kernel;
        * a float dataset, no augmentation asked: collated by the DataLoader, passed through;
        * a float dataset + `data_augmentation`: the raw uint8 images the dataset object carries (`.data`, `.targets` or
          `.labels`; behind a `random_split` Subset: its `.dataset` and `.indices`) are gathered per batch by the SAME
          DataLoader machinery run over the sample indices - the shuffling consumes the global generator exactly as the
          reference's loader does - and augmented on the device.  Accepted only if the dataset's own transform is ToTensor and
          nothing else: its first and last samples must equal `data[i] / 255` bit for bit, and its target_transform (held-out
          classes: utils/torch_load.py:338-343) is applied to the gathered labels."""
        model = self

        class Batches:
            def __init__(self, loader, convert):
                self.loader, self.convert = loader, convert

            def __len__(self):
                return len(self.loader)

            def __iter__(self):
                for item in self.loader:
                    yield self.convert(item)
        probe = trainset[0][0] if len(trainset) else None
        is_raw = torch.is_tensor(probe) and probe.dtype == torch.uint8
        if is_raw or not data_augmentation or probe is None:
            loader = torch.utils.data.DataLoader(trainset, batch_size=batch_size, shuffle=True, num_workers=0)
            return Batches(loader, lambda b: (model._device_batch(b[0].to(device), data_augmentation, set_name), b[1].to(device)))
        base, indices = trainset, None
        while isinstance(base, torch.utils.data.Subset):
            idx = torch.as_tensor(base.indices, dtype=torch.int64)
            indices = idx if indices is None else idx[indices]
            base = base.dataset
        data = getattr(base, 'data', None)
        labels = getattr(base, 'targets', None)
        if labels is None:
            labels = getattr(base, 'labels', None)
        why = None
        if data is None or labels is None:
            why = 'the dataset carries no raw images (.data and .targets / .labels)'
        else:
            data = torch.as_tensor(np.asarray(data) if not torch.is_tensor(data) else data)
            labels = torch.as_tensor(np.asarray(labels) if not torch.is_tensor(labels) else labels).to(torch.int64)
            C, H, W = self.input_shape
            if data.dtype != torch.uint8:
                why = 'its .data is {} (uint8 expected)'.format(data.dtype)
            elif data.dim() == 3 and C == 1 and tuple(data.shape[1:]) == (H, W):
                data = data.unsqueeze(1)
            elif data.dim() != 4 or tuple(data.shape[1:]) not in ((H, W, C), (C, H, W)):
                why = 'its .data is shaped {} (input_shape {})'.format(tuple(data.shape), tuple(self.input_shape))
        if why is None:
            n_base = len(base)
            if len(data) != n_base or len(labels) != n_base:
                why = '.data / .targets hold {} / {} entries for {} samples'.format(len(data), len(labels), n_base)
        if why is None:
            nhwc = tuple(data.shape[1:]) == (H, W, C) and not (C == H == W)
            for i in {0, n_base - 1}:
                xi = base[i][0]
                raw = data[i].permute(2, 0, 1) if nhwc else data[i]
                if not torch.is_tensor(xi) or tuple(xi.shape) != (C, H, W) or not torch.equal(
                        xi.to(torch.float32), raw.to(torch.float32).div(255)):
                    why = 'sample {} of the dataset is not its raw image / 255: its own transform is more than ToTensor'.format(i)
                    break
        if why is not None:
            raise ValueError('data_augmentation={} needs the raw uint8 images (the reference augments PIL images before '
                             'ToTensor, utils/torch_load.py:405-426); this dataset yields {} and {}'.format(
                                 list(data_augmentation), getattr(probe, 'dtype', type(probe)), why))
        target_transform = getattr(base, 'target_transform', None)

        class Indices(torch.utils.data.Dataset):
            def __len__(self):
                return len(trainset)

            def __getitem__(self, i):
                return i

        def gather(idx):
            if indices is not None:
                idx = indices[idx]
            y = labels[idx]
            if target_transform is not None:
                y = torch.tensor([int(target_transform(int(v))) for v in y.tolist()], dtype=torch.int64)
            return (model._device_batch(data[idx].to(device), data_augmentation, set_name), y.to(device))
        loader = torch.utils.data.DataLoader(Indices(), batch_size=batch_size, shuffle=True, num_workers=0)
        return Batches(loader, gather)

    def _device_batch(self, x, data_augmentation=(), set_name=''):
        """Collated training batch -> the float32 (N, *input_shape) tensor the step takes.  Raw uint8 images go through the
        input-pipeline kernel (flip / edge-pad crop decided per image on the device, then /255: the ToTensor of
        utils/torch_load.py:415-426); float batches pass through and cannot be augmented."""
        if x.dtype != torch.uint8:
            if data_augmentation:
                raise ValueError('data_augmentation={} needs the raw uint8 images (the reference augments PIL images before '
                                 'ToTensor, utils/torch_load.py:405-426); this dataset yields {}'.format(list(data_augmentation), x.dtype))
            return x
        C, H, W = self.input_shape
        if tuple(x.shape[1:]) == (H, W, C):
            nhwc = True
        elif tuple(x.shape[1:]) == (C, H, W):
            nhwc = False
        else:
            raise ValueError('uint8 batch of shape {} matches neither (N,{},{},{}) nor (N,{},{},{})'.format(
                tuple(x.shape), H, W, C, C, H, W))
        pad = 0
        if 'crop' in data_augmentation:          # utils/torch_load.py:409-411
            pad = 0 if 'imagenet' in set_name else H // 8
        flip, dy, dx = ops.draw_augmentation(x.shape[0], pad, x.device, generator=getattr(self, 'augmentation_generator', None),
                                             flip='flip' in data_augmentation, crop=pad > 0)
        return ops.augment_batch(x, flip, dy, dx, pad=pad, nhwc=nhwc)

    # ------------------------------------------------------------------------------------ persistence
    def save(self, dir_name=None, except_optimizer=False, except_state=False):
        """params.json / train_params.json / test.json / ood.json / history.json, and - once the model is trained and
        unless `except_state` - state.pth plus, unless `except_optimizer`, optimizer.pth (cvae.py:2650-2675; the
        fine-tuning jobs save with except_optimizer=True and except_state=<bool>: ft/job.py:154-158)."""
        if dir_name is None:
            dir_name = getattr(self, 'saved_dir', None) or os.path.join('jobs', self.print_architecture(),
                                                                        str(self.job_number))
        os.makedirs(dir_name, exist_ok=True)
        tp = dict(self.training_parameters)
        tp['sigma'] = self.sigma.params

        def dump(obj, name):
            with open(os.path.join(dir_name, name), 'w') as f:
                json.dump(obj, f, default=str)
        dump(self.architecture, 'params.json')
        dump(tp, 'train_params.json')
        dump(self.testing, 'test.json')
        dump(self.ood_results, 'ood.json')
        dump(self.train_history, 'history.json')
        if self.trained and not except_state:
            torch.save(self.state_dict(), os.path.join(dir_name, 'state.pth'))
            if not except_optimizer:
                torch.save(self.optimizer.state_dict(), os.path.join(dir_name, 'optimizer.pth'))
        self.saved_dir = dir_name
        return dir_name

    @classmethod
    def load(cls, dir_name, build_module=True, load_state=True, load_train=True, load_test=True, strict=True,
             device=None, **kw):
        """Rebuild a model from a job directory written by save() of this class OR of the reference
        (cvae.py:2677-2857): params.json + train_params.json give the constructor arguments, state.pth /
        optimizer.pth the tensors; history.json / test.json / ood.json the training record (`trained` =
        history['epochs'], and the learning-rate schedule is fast-forwarded by that many epochs: cvae.py:2815-2851)."""
        def read(name, int_keys=False, default=None):
            path = os.path.join(dir_name, name)
            if default is not None and not os.path.exists(path):
                return default
            with open(path) as f:
                d = json.load(f)
            if int_keys:                         # utils/save_load/misc.py:58-66 (presumed_type=int)
                d = {(int(k) if k.lstrip('-').isdigit() else k): v for k, v in d.items()}
            return d
        arch = read('params.json')
        tp = read('train_params.json')
        ctor = {k: arch[k] for k in ('input_shape', 'num_labels', 'type', 'output_distribution', 'representation',
                                     'encoder', 'batch_norm', 'dropout', 'activation', 'encoder_forced_variance',
                                     'latent_dim', 'test_latent_sampling', 'decoder', 'upsampler', 'classifier',
                                     'output_activation') if k in arch}
        ctor['features'] = arch.get('features')
        prior = dict(arch.get('prior', {}))
        for drop in ('dim', 'num_priors'):
            prior.pop(drop, None)
        ctor['prior'] = prior
        sig = {k: v for k, v in dict(tp.get('sigma', {'value': 1})).items()
               if k in ('value', 'learned', 'is_rmse', 'sdim', 'input_dim', 'reach', 'decay', 'max_step', 'sigma0', 'is_log')}
        if sig.get('learned'):          # `value` in the json is the CURRENT rms; the state_dict restores the tensor anyway
            sig['value'] = sig.get('sigma0') or sig.get('value')
        if not sig.get('reach'):
            sig.pop('reach', None)
        ctor['sigma'] = sig
        ctor['beta'] = tp.get('beta', 1.)
        ctor['gamma'] = tp.get('gamma') or 0.
        ctor['latent_sampling'] = tp.get('latent_sampling', 1)
        ctor['optimizer'] = dict(tp.get('optimizer', {}))
        net = cls(**ctor)
        net.training_parameters.update({k: v for k, v in tp.items() if k not in ('sigma', 'optimizer')})
        net.saved_dir = dir_name
        history = read('history.json', int_keys=True, default={'epochs': 0})
        net.train_history = history
        net.trained = int(history.get('epochs', 0))
        if load_test:
            tested = read('test.json', int_keys=True, default={})
            if tested:
                net.testing.update(tested)
            net.ood_results = read('ood.json', int_keys=True, default={})
        if device is not None:
            net.to(device)
        if load_state and build_module and os.path.exists(os.path.join(dir_name, 'state.pth')):
            net.load_weights(dir_name, strict=strict)
            net.optimizer.update_scheduler_from_epoch(net.trained)
        return net

    def load_weights(self, dir_name, strict=True, with_optimizer=True):
        """Load state.pth (+ optimizer.pth) written by save() of this class or of the reference."""
        state = torch.load(os.path.join(dir_name, 'state.pth'), map_location=self.device)
        _lib.pack_cache_end()                    # the weights change below
        self.optimizer._scan_pending = True      # ... to values nobody has vouched for: scanned before the next backward
        with torch.no_grad():
            mine = self.state_dict()
            for k, v in state.items():
                if k in mine:
                    mine[k].copy_(v)
                elif strict:
                    raise KeyError(k)
        opt = os.path.join(dir_name, 'optimizer.pth')
        if with_optimizer and os.path.exists(opt):
            self.optimizer.load_state_dict(torch.load(opt, map_location=self.device))
        return self

    def print_architecture(self, *a, **kw):
        arch = self.architecture
        return '{type} K={latent_dim} features={f} upsampler={upsampler}'.format(f=arch.get('features'), **arch)

    def print_training(self, *a, **kw):
        return 'sigma={} optimizer={}'.format(self.sigma, self.optimizer)
