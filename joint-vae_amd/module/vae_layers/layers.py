"""Dense blocks of the joint CVAE on HIP kernels: Sigma, Sampling, Encoder, Classifier.

Same public surface as the reference's module/vae_layers/layers.py — `Sigma` (:73-213), `Sampling`
(:216-250), `Encoder` (:253-403), `Classifier` (:456-483) — and the same `state_dict` keys
(`encoder.dense_projs.{0,2,..}`, `encoder.dense_mean`, `encoder.dense_log_var`, `encoder.prior.*`,
`classifier.{0,2,..}`).  The unused colour-space / Decoder helpers of that file are not rebuilt.
"""
import logging
import math

import numpy as np
import torch
from torch import nn
from torch.nn import Parameter

from jvae_hip import ops
from jvae_compat import texify_str
from module.priors import build_prior
from .misc import activation_layers, ACT_OF_MODULE


def _sigma_kind(is_rmse, coded, learned, decay):
    """Which of the five behaviours a Sigma has (first match wins, the reference's precedence: layers.py:194-215)."""
    for kind, on in (('rmse', is_rmse), ('coded', coded), ('learned', learned), ('decayed', decay)):
        if on:
            return kind
    return 'fixed'


class Sigma(Parameter):
    """Standard deviation of p(x|z) as an nn.Parameter (reference: module/vae_layers/layers.py:73-213; used at
    cvae.py:626-670,769).  Five kinds:

      fixed    value given, never changes                          str: '0.5'
      decayed  fixed start, pulled towards reach * rmse            str: '1->2*rmse[-0.1*<0.05]'
      learned  stored as log sigma, receives gradient              str: '1->rmse[l] (0.98)'
      rmse     follows the running rmse of the batch exactly       str: 'rmse (0.29)'
      coded    log sigma is an output of the encoder               str: 'coded scalar' / 'coded mask'

    The public attributes (`params` lists every one without a leading underscore) and the printed forms are what the
    reference writes into train_params.json and into job-directory names, so they are kept verbatim; the constructor
    keywords are the reference's: Sigma(value, sdim, input_dim, reach, decay, max_step, learned, is_rmse, sigma0, is_log).
    """

    @staticmethod
    def __new__(cls, value=None, sdim=1, input_dim=False, learned=False, is_rmse=False, is_log=False, **_):
        if value is None:
            assert is_rmse or input_dim, 'a Sigma needs a value unless it follows the rmse or is coded'
        stored_as_log = bool(is_log or learned or input_dim)       # coded implies learned implies log storage
        if is_rmse or (input_dim and value is None):
            start = 0.
        else:
            start = float(value)
        shape = (sdim,) if isinstance(sdim, int) else tuple(sdim)
        with np.errstate(divide='ignore'):
            init = float(np.log(start)) if stored_as_log else start
        return super().__new__(cls, torch.full(shape, init), requires_grad=bool(learned or input_dim))

    def __init__(self, value=None, learned=False, is_rmse=False, sdim=1, input_dim=False, reach=1, decay=0,
                 max_step=None, sigma0=None, is_log=False):
        if learned:
            assert not is_rmse and not decay, 'a learned sigma neither follows the rmse nor decays'
        follows = bool(decay) or is_rmse
        public = dict(is_rmse=is_rmse,
                      sigma0=sigma0 if (sigma0 is not None or is_rmse) else value,
                      learned=learned,
                      input_dim=input_dim,
                      is_log=learned or is_log or input_dim,
                      decay=1 if is_rmse else decay,
                      reach=reach if follows else None,
                      max_step=max_step,
                      sdim=sdim)
        self._rmse = np.nan
        self.__dict__.update(public)                               # insertion order = key order of `params`
        self._output_dim = None
        if input_dim:
            self._output_dim = input_dim if sdim != 1 else (1,) * len(input_dim)

    def __deepcopy__(self, memo):
        twin = Sigma.__new__(Sigma, value=1.0, sdim=self.sdim, learned=self.requires_grad)
        twin.__dict__.update(self.__dict__)
        twin.data = self.data.clone()
        memo[id(self)] = twin
        return twin

    # -- read-only views --------------------------------------------------------------------------
    coded = property(lambda self: bool(self.input_dim))
    per_dim = property(lambda self: self.sdim != 1)
    output_dim = property(lambda self: self._output_dim)
    kind = property(lambda self: _sigma_kind(self.is_rmse, self.coded, self.learned, self.decay))

    @property
    def value(self):
        """Root mean square of sigma over its dimensions, as a Python float (a device read-back: the training step
        reads it from the packed measures instead)."""
        with torch.no_grad():
            squares = (2 * self.data).exp() if self.is_log else self.data * self.data
            return float(squares.mean().sqrt())

    def host_params(self, value):
        """`params` with an already known rms value (no device read-back)."""
        listed = {k: v for k, v in vars(self).items() if k[:1] != '_'}
        listed['value'] = value
        return listed

    @property
    def params(self):
        return self.host_params(self.value)

    # -- updates ----------------------------------------------------------------------------------
    def update(self, rmse=None, v=None):
        """`v`: new (coded) values, averaged over the leading batch dimensions.  `rmse`: the batch rmse; a decayed /
        rmse sigma moves by decay * (reach * rmse - sigma), at most `max_step` in magnitude."""
        assert rmse is None or v is None
        if v is not None:
            extra = v.dim() - self.dim()
            assert extra >= 0
            self.data = v.mean(tuple(range(extra))) if extra else v
        elif rmse is not None:
            self._rmse = rmse
            if self.kind in ('rmse', 'decayed'):
                step = self.decay * (self.reach * rmse - self.data)
                if self.max_step:
                    step = torch.clamp(torch.as_tensor(step), -self.max_step, self.max_step)
                self.data += step

    # -- printed forms (job-directory names, train_params.json: kept as the reference prints them) ----------------
    def _describe(self):
        kind = self.kind
        if kind == 'rmse':
            return 'rmse' if self._rmse is np.nan else 'rmse ({:g})'.format(self._rmse)
        if kind == 'coded':
            return 'coded ' + ('mask' if self.per_dim else 'scalar')
        if kind == 'learned':
            return '{:g}->rmse[l] ({:g})'.format(self.sigma0, self.value)
        if kind == 'fixed':
            return '{:g}'.format(self.value if self.dim() and self.numel() > 1 else float(self.data))
        target = ('' if self.reach == 1 else '{:g}*'.format(self.reach)) + 'rmse'
        limit = '<{:g}'.format(self.max_step) if self.max_step else ''
        return '{:g}->{}[-{:g}*{}]'.format(self.sigma0, target, self.decay, limit)

    def __str__(self):
        return self._describe()

    def __format__(self, spec):
        tail = spec[-1:] if spec else ''
        if tail in ('f', 'g', 'e'):
            return format(self.value, spec)
        if tail == 'x':
            return texify_str(self._describe(), num=True)
        if tail == 'i':
            letter = {'rmse': 'e', 'learned': 'l', 'coded': 'C' if self.per_dim else 'c'}.get(self.kind)
            if letter:
                return letter
        return self._describe()

    def __repr__(self):
        if self.kind == 'rmse':
            return 'Sigma will be RMSE'
        text = super().__repr__()
        if self.decay:
            text = '{}, decaying to {}*mse with rate {})'.format(text[:-1], self.reach, self.decay)
        return text


def draw_epsilon(sampling_size, shape, device, distribution='gaussian'):
    """(L+1, *shape) noise from the device's default generator, row 0 zeroed (layers.py:233-238)."""
    size = (sampling_size + 1,) + tuple(shape)
    if distribution == 'gaussian':
        eps = torch.randn(size, device=device)
    else:
        eps = (torch.rand(size, device=device) - 0.5) * math.sqrt(12)
    eps[0] = 0
    return eps


class Sampling(nn.Module):
    """z[l] = mu + exp(log_var / 2) * eps[l] * is_sampled for l = 0..L with eps[0] = 0.

    forward(z_mean, z_log_var[, epsilon]) -> (z (L+1, ..., K), eps[1:]).  Runs the fused latent kernel with a
    standard normal prior (its KL outputs are discarded); Encoder uses the kernel directly to get both.
    """

    def __init__(self, latent_dim, sampling_size=1, sampling=True, distribution='gaussian', **kwargs):
        assert distribution in ('gaussian', 'uniform'), '{} for sampling unknown'.format(distribution)
        super().__init__(**kwargs)
        self.distribution = distribution
        self.sampling_size = sampling_size
        self.is_sampled = sampling

    def forward(self, z_mean, z_log_var, epsilon=None):
        K = z_mean.shape[-1]
        if epsilon is None:
            epsilon = draw_epsilon(self.sampling_size, z_log_var.shape, z_mean.device, self.distribution)
        mu2 = z_mean.reshape(-1, K)
        n = mu2.shape[0]
        labels = torch.zeros(n, dtype=torch.int64, device=z_mean.device)
        means = torch.zeros((1, K), device=z_mean.device)
        T = torch.ones(1, device=z_mean.device)
        _, z, _, _, _, _ = ops.latent(mu2, z_log_var.reshape(-1, K), epsilon.reshape(epsilon.shape[0], n, K), labels,
                                      means, T, sampled=bool(self.is_sampled))
        return z.reshape(epsilon.shape), epsilon[1:]

    def __repr__(self):
        if not self.is_sampled:
            return 'Deactivated, returns mean'
        return 'Sampling({}, L={})'.format(self.distribution, self.sampling_size)


class HipLinear(nn.Linear):
    """nn.Linear parameters; forward = MFMA GEMM with the bias (and a following activation) in the epilogue."""

    def forward(self, x, act=ops.IDENT):
        return ops.linear(x, self.weight, self.bias, act)


class DenseStack(nn.Sequential):
    """Linear / activation chain; fuses each activation into the producing GEMM's epilogue."""

    def forward(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(m, HipLinear) and type(nxt) in ACT_OF_MODULE:
                x = m(x, ACT_OF_MODULE[type(nxt)])
                i += 2
            else:
                x = m(x)
                i += 1
        return x


class HipDropout(nn.Module):
    """nn.Dropout(p) (reference layers.py:287-288, cvae.py:297-298): identity in eval mode; in train mode a HIP kernel
    with a counter-based mask.  The seed lives in DEVICE memory: it is drawn once from torch's default CPU generator (so
    torch.manual_seed makes runs repeatable; data-parallel ranks are offset by their rank so that replicas draw different
    masks) and advanced on the stream after every call - no host value enters a launch, so a captured HIP graph
    (graph_train_step) draws a fresh mask at every replay.  The mask itself is not torch's."""

    _STRIDE = 0x9E3779B97F4A7C15 & (2 ** 62 - 1)

    def __init__(self, p=0.5):
        super().__init__()
        if not 0. <= p < 1.:
            raise ValueError('dropout probability has to be in [0, 1), got {}'.format(p))
        self.p = float(p)
        self._seed = None

    def _next_seed(self, device):
        if self._seed is None or self._seed.device != device:
            first = int(torch.randint(0, 2 ** 61, (1,)))
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                first += torch.distributed.get_rank() * 1_000_003
            self._seed = torch.tensor([first], dtype=torch.int64, device=device)
        now = self._seed.clone()                      # this call's seed (kept for the backward pass)
        self._seed += self._STRIDE % (2 ** 31)        # advance on the stream
        return now

    def forward(self, x):
        if not self.training or self.p == 0.:
            if not x.is_cuda:
                raise ops.L.JvaeHipError('jvae_hip ops need tensors resident on the GPU (no CPU fallback)')
            return x
        return ops.dropout(x, self.p, self._next_seed(x.device))

    def extra_repr(self):
        return f'p={self.p}'


class Encoder(nn.Module):
    """Dense trunk -> (mu, log sigma^2) heads -> reparameterised samples; owns the latent prior."""

    def __init__(self, input_shape, num_labels, representation='rgb', y_is_coded=False, latent_dim=32,
                 intermediate_dims=[64], name='encoder', dropout=False, activation='relu', sampling_size=10,
                 sampling=True, sigma_output_dim=0, forced_variance=False, prior={}, **kwargs):
        super().__init__(**kwargs)
        self.name = name
        self.y_is_coded = y_is_coded
        self.input_shape = input_shape
        self.num_labels = num_labels
        self.forced_variance = forced_variance
        self._sampling_size = sampling_size
        width = int(np.prod(input_shape)) + num_labels * bool(y_is_coded)
        trunk = []
        for d in intermediate_dims:
            trunk += [HipLinear(width, d), activation_layers[activation]()]
            if dropout:
                trunk.append(HipDropout(p=dropout))
            width = d
        self.dense_projs = DenseStack(*trunk)
        self.dense_mean = HipLinear(width, latent_dim)
        self.dense_log_var = HipLinear(width, latent_dim)
        self.sigma_output_dim = sigma_output_dim
        if sigma_output_dim:
            self.sigma = HipLinear(width, int(np.prod(sigma_output_dim)))

        noise = 'uniform' if prior.get('distribution', 'gaussian') == 'uniform' else 'gaussian'
        self.sampling = Sampling(latent_dim, sampling_size, sampling, distribution=noise)
        prior['dim'] = latent_dim
        self.prior = build_prior(**prior)
        logging.debug('Built %s', self.prior)

    @property
    def sampling_size(self):
        return self._sampling_size

    @sampling_size.setter
    def sampling_size(self, v):
        self._sampling_size = v
        self.sampling.sampling_size = v

    # -- dictionary diagnostics (evaluation / logging only) -----------------------------------------
    def capacity(self):
        """Upper bound of I(Z;Y) from the pairwise distances of the class means (layers.py:323-336)."""
        m = self.prior.mean
        C = self.num_labels
        d2 = torch.cdist(m, m).pow(2)
        return math.log(C) - torch.exp(-d2 / 4).sum(0).log().sum() / C

    def dict_min_distance(self):
        m = self.prior.mean
        C = self.num_labels
        guard = 2 * m.norm(dim=1).max() * torch.eye(C, device=m.device)
        return (torch.cdist(m, m) + guard).min()

    # -- forward -------------------------------------------------------------------------------------
    def heads(self, x, y=None):
        """Trunk + the two heads: (u, mu, raw log-variance) — raw = before the +-20 clip."""
        u = x if y is None else torch.cat((x, y), dim=-1)
        u = self.dense_projs(u)
        return u, self.dense_mean(u), (None if self.forced_variance else self.dense_log_var(u))

    def encode(self, x, y_onehot, labels, kl_var_weighting=1., epsilon=None):
        """Fused path used by the training step: heads, then ONE latent kernel for clip + samples + KL.

        labels: int64 class of every row (prior component), or None for a non-conditional prior.
        Returns (mu, log_var, z, eps[1:], sigma_coded, kl_terms dict with kl / distance / var_kl / dzdist).
        """
        u, mu, lv_raw = self.heads(x, y_onehot)
        K = mu.shape[-1]
        batch = mu.shape[:-1]
        mu2 = mu.reshape(-1, K)
        n = mu2.shape[0]
        pr = self.prior
        if epsilon is None:
            epsilon = draw_epsilon(self._sampling_size, mu.shape, mu.device, self.sampling.distribution)
        lab, means, T = pr._kernel_operands(labels if pr.conditional else None, n, mu.device)
        forced = math.log(self.forced_variance) if self.forced_variance else None
        raw2 = mu2 if lv_raw is None else lv_raw.reshape(-1, K)
        lv, z, kl, dist, vkl, dzd = ops.latent(mu2, raw2, epsilon.reshape(epsilon.shape[0], n, K), lab, means, T,
                                               prior=pr.distribution, var_dim=pr.var_dim, tau=pr._tau,
                                               alpha=pr._alpha_k, w=kl_var_weighting,
                                               sampled=bool(self.sampling.is_sampled), forced_lv=forced)
        sigma = self.sigma(u) if self.sigma_output_dim else None
        terms = {'kl': kl.reshape(batch), 'distance': dist.reshape(batch), 'var_kl': vkl.reshape(batch),
                 'dzdist': dzd.reshape(batch)}
        return mu, lv.reshape(mu.shape), z.reshape(epsilon.shape), epsilon[1:], sigma, terms

    def forward(self, x, y=None, epsilon=None):
        """x (..., D) [, y one-hot (..., C)] -> (z_mean, z_log_var, z (L+1, ..., K), eps[1:], sigma)."""
        labels = None
        if self.prior.conditional:      # KL terms are discarded here: any component will do
            labels = torch.zeros(x.shape[:-1], dtype=torch.int64, device=x.device)
        mu, log_var, z, e, sigma, _ = self.encode(x, y, labels, epsilon=epsilon)
        return mu, log_var, z, e, sigma


class Classifier(DenseStack):
    """Latent (..., K) -> logits (..., C): Linear(+activation) chain ending in Linear(num_labels)."""

    def __init__(self, latent_dim, num_labels, intermediate_dims=[], name='classifier', activation='relu', **kwargs):
        layers = []
        width = latent_dim
        act = activation_layers[activation]()          # one shared activation module, as in the reference
        for d in intermediate_dims:
            layers += [HipLinear(width, d), act]
            width = d
        layers.append(HipLinear(width, num_labels))
        super().__init__(*layers, **kwargs)
        self.name = name
