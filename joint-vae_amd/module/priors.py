"""Latent priors of the joint CVAE, evaluated by the fused HIP latent kernel (csrc/latent.hip).

Public surface of the reference's module/priors.py: `build_prior` (:35-52), `GaussianPrior` (:55-353),
`TiltedGaussianPrior` (:356-408), `UniformWithGaussianTailPrior` (:411-499) with the same constructor
arguments, parameters (`mean` (C,K), `_var_parameter` (C,) | (C,K) | (C,K,K)), `params` dict and methods.

Reminder: `_var_parameter` / `inv_trans` is the WHITENING factor T of each component
(whiten = T.(x - m), precision = T^T T, log|Sigma| = -2 sum log|T_kk|), not a variance.

`kl()` / `mahala()` run the same kernel as the training step; `log_density`, `log_det_per_class`,
`whiten`, `trace_prod_by_var` are evaluation-time helpers written as tiny per-class
tensor expressions.
"""
import logging
import math

import torch
from torch import nn
from torch.nn import Parameter

from jvae_hip import ops
from jvae_compat import texify_str  # noqa: F401  (kept importable from here like the reference)


def build_prior(dim, distribution='gaussian', **kw):
    """Factory used by Encoder (layers.py:305-306 of the reference)."""
    known = ('gaussian', 'tilted', 'uniform')
    assert distribution in known, '{} unknown (try one of: {})'.format(distribution, ', '.join(known))
    if kw.get('num_priors', 1) == 1:
        kw.pop('learned_means', False)
    if distribution == 'gaussian':
        if kw.pop('tau', None) is not None:
            logging.debug('discarded value of tau for gaussian prior')
        return GaussianPrior(dim, **kw)
    if kw.pop('var_dim', 'scalar') != 'scalar':
        logging.info('discarded variance type for %s prior', distribution)
    cls = TiltedGaussianPrior if distribution == 'tilted' else UniformWithGaussianTailPrior
    return cls(dim, **kw)


class _KLTerms(dict):
    """The dictionary GaussianPrior.kl returns (priors.py:287-324): keys trace, log_det_prior, log_det, distance, var_kl,
    kl in the reference's order.  distance / var_kl / kl come out of the fused kernel; the three terms var_kl is made of
    are diagnostics nobody on the training / evaluation path reads, so they are derived on first access (from the prior's
    public helpers, as the reference does) instead of costing three more passes per call."""

    _LAZY = ('trace', 'log_det_prior', 'log_det')

    def __init__(self, prior, log_var, y, distance, var_kl, kl):
        super().__init__()
        self._src = (prior, log_var, y)
        for k in self._LAZY:
            dict.__setitem__(self, k, None)
        dict.__setitem__(self, 'distance', distance)
        dict.__setitem__(self, 'var_kl', var_kl)
        dict.__setitem__(self, 'kl', kl)

    def _fill(self):
        if self._src is None:
            return
        prior, log_var, y = self._src
        self._src = None
        with torch.no_grad():
            dict.__setitem__(self, 'trace', prior.trace_prod_by_var(log_var.exp(), y))
            ld = prior.log_det_per_class()
            if prior.conditional:
                ld = ld.index_select(0, y.reshape(-1)).view(y.shape)
            dict.__setitem__(self, 'log_det_prior', ld)
            dict.__setitem__(self, 'log_det', log_var.sum(-1))

    def __getitem__(self, k):
        if k in self._LAZY:
            self._fill()
        return dict.__getitem__(self, k)

    def get(self, k, default=None):
        return self[k] if k in self else default

    def values(self):
        self._fill()
        return dict.values(self)

    def items(self):
        self._fill()
        return dict.items(self)


class GaussianPrior(nn.Module):
    """C class-conditional Gaussians N(m_c, (T_c^T T_c)^-1); in training `y` selects ONE component."""

    distribution = 'gaussian'

    def __init__(self, dim, var_dim='scalar', num_priors=1, init_mean=0, mean_shift=0, learned_means=False,
                 freeze_means=0, force_conditional=False, seed=None):
        assert not learned_means or num_priors > 1
        if var_dim not in ('scalar', 'diag', 'full'):
            raise ValueError('var_dim {} unknown'.format(var_dim))
        super().__init__()
        gen = torch.Generator()
        if seed is None:
            gen.seed()
        else:
            gen.manual_seed(seed)
            logging.info('Seed for prior: {}'.format(seed))

        self.dim = dim
        self.num_priors = num_priors
        self.var_dim = var_dim
        self.learned_var = var_dim != 'scalar'
        self.learned_means = learned_means
        self.freeze_means = freeze_means
        self.conditional = num_priors > 1 or bool(force_conditional)

        if num_priors == 1:
            means = init_mean * torch.randn(1, dim, generator=gen) + mean_shift
        elif isinstance(init_mean, str) and init_mean == 'onehot':
            assert dim >= num_priors, 'K={}<C={}'.format(dim, num_priors)
            means = torch.eye(num_priors, dim)
        elif torch.is_tensor(init_mean):
            means = init_mean.squeeze().clone()
        else:
            means = float(init_mean) * torch.randn(num_priors, dim, generator=gen).squeeze() + mean_shift
        self._frozen_means = (not learned_means) or freeze_means > 0
        self.mean = Parameter(means, requires_grad=not self._frozen_means)

        unit = {'scalar': torch.tensor(1.), 'diag': torch.ones(dim), 'full': torch.eye(dim)}[var_dim]
        factor = torch.stack([unit] * num_priors) if self.conditional else unit
        self._var_parameter = Parameter(factor.clone(), requires_grad=self.learned_var)

        self.params = {'distribution': 'gaussian', 'dim': dim, 'init_mean': init_mean,
                       'var_dim': self.var_dim, 'num_priors': self.num_priors}
        if self.conditional:
            self.params.update({'learned_means': self.learned_means, 'freeze_means': freeze_means})

    # ---- bookkeeping -------------------------------------------------------------------------------
    def thaw_means(self, epoch=None):
        """Start training the dictionary once `epoch >= freeze_means` (called every epoch, cvae.py:2420)."""
        if not self.learned_means or not self._frozen_means:
            return
        if epoch is None or epoch >= self.freeze_means:
            logging.debug('Defreezing prior means')
            self.mean.requires_grad_()
            self._frozen_means = True          # sic: the reference leaves the flag set (harmless re-run)

    @property
    def inv_trans(self):
        return self._var_parameter.tril() if self.var_dim == 'full' else self._var_parameter

    @property
    def inv_var(self):
        if self.var_dim == 'full':
            t = self.inv_trans
            return torch.matmul(t.transpose(-1, -2), t)
        return self._var_parameter ** 2

    # ---- kernel plumbing ---------------------------------------------------------------------------
    def _kernel_operands(self, y, n, device):
        """(labels (n,), means (C,K), T) in the layout the latent kernel expects."""
        if self.conditional:
            assert y is not None
            labels = y.reshape(-1)
            means, T = self.mean, self._var_parameter
        else:
            assert y is None
            labels = torch.zeros(n, dtype=torch.int64, device=device)
            means, T = self.mean.reshape(1, self.dim), self._var_parameter.unsqueeze(0)
        return labels, means, T

    _tau = 0.
    _alpha_k = 0.

    def _run(self, mu, log_var, y, var_weighting):
        """Flatten leading dims, run the fused kernel without sampling, restore the batch shape."""
        batch = mu.shape[:-1]
        mu2 = mu.reshape(-1, self.dim)
        lv2 = log_var.reshape(-1, self.dim)
        n = mu2.shape[0]
        labels, means, T = self._kernel_operands(y, n, mu.device)
        eps = torch.zeros((1, n, self.dim), device=mu.device, dtype=torch.float32)
        _, _, kl, dist, vkl, _ = ops.latent(mu2, lv2, eps, labels, means, T, prior=self.distribution,
                                            var_dim=self.var_dim, tau=self._tau, alpha=self._alpha_k,
                                            w=var_weighting, sampled=False)
        return kl.reshape(batch), dist.reshape(batch), vkl.reshape(batch)

    def _broadcast_over_classes(self, mu, log_var, y):
        """All-class evaluation (y has one more leading dim than the batch): repeat mu / log_var over it."""
        shape = (y.shape[0],) + tuple(mu.shape)
        return mu.unsqueeze(0).expand(shape), log_var.unsqueeze(0).expand(shape)

    # ---- public API ----------------------------------------------------------------------------------
    def kl(self, mu, log_var, y=None, output_dict=True, var_weighting=1.):
        """KL(q(z|x) || p(z|y)) per sample.  mu, log_var: (..., K) (log_var within +-20); y: (...) or None.

        kl = 1/2 (distance + w * var_kl), var_kl = tr(T^T T diag(var)) - sum(log_var) + log|Sigma_y| - K.
        """
        if y is not None and y.ndim == mu.ndim:
            mu, log_var = self._broadcast_over_classes(mu, log_var, y)
        kl, dist, vkl = self._run(mu, log_var, y, var_weighting)
        if not output_dict:
            return kl
        return _KLTerms(self, log_var, y, dist, vkl, kl)

    def mahala(self, x, y=None):
        """Squared Mahalanobis distance |T_y (x - m_y)|^2, shape x.shape[:-1]."""
        _, dist, _ = GaussianPrior._run_gauss(self, x, y)
        return dist

    def _run_gauss(self, x, y):
        batch = x.shape[:-1]
        x2 = x.reshape(-1, self.dim)
        n = x2.shape[0]
        labels, means, T = self._kernel_operands(y, n, x.device)
        eps = torch.zeros((1, n, self.dim), device=x.device, dtype=torch.float32)
        lv = torch.zeros_like(x2)
        _, _, kl, dist, vkl, _ = ops.latent(x2, lv, eps, labels, means, T, prior='gaussian', var_dim=self.var_dim,
                                            sampled=False)
        return kl.reshape(batch), dist.reshape(batch), vkl.reshape(batch)

    def log_det_per_class(self):
        """log|Sigma_c| for every component (eval helper)."""
        t = self.inv_trans
        if self.var_dim == 'full':
            return -2 * torch.diagonal(t, dim1=-2, dim2=-1).abs().log().sum(-1)
        if self.var_dim == 'diag':
            return -2 * t.abs().log().sum(-1)
        return -2 * self.dim * t.log()

    def whiten(self, x, y=None):
        assert self.conditional ^ (y is None)
        t = self.inv_trans.index_select(0, y.reshape(-1)) if self.conditional else self.inv_trans
        if self.var_dim == 'full':
            return torch.matmul(t, x.unsqueeze(-1)).squeeze(-1)
        return x * (t if self.var_dim == 'diag' else t.unsqueeze(-1))

    def trace_prod_by_var(self, var, y=None):
        assert self.conditional ^ (y is None)
        t = self.inv_trans
        diag = t.pow(2).sum(-2) if self.var_dim == 'full' else t.pow(2)
        if self.conditional:
            diag = diag.index_select(0, y.reshape(-1))
        if self.var_dim == 'scalar':
            diag = diag.unsqueeze(-1)
        return (var.reshape(-1, self.dim) * diag).sum(-1).reshape(var.shape[:-1])

    def log_density(self, z, y=None):
        """log p(z | y) (importance-weighting at evaluation time, cvae.py:806)."""
        assert self.conditional ^ (y is None)
        u = GaussianPrior.mahala(self, z, y)
        log_det = self.log_det_per_class()
        if self.conditional:
            log_det = log_det.index_select(0, y.reshape(-1)).view(u.shape)
        return -math.log(2 * math.pi) * self.dim / 2 - u / 2 - log_det / 2

    def __repr__(self):
        pre = 'conditional ' if self.conditional else ''
        var = ('learned ' if self.learned_var else '') + self.var_dim + ' variance'
        if self.conditional:
            mean = '{} {}means and '.format(self.num_priors, 'learned ' if self.learned_means else '')
        elif self.params['init_mean']:
            mean = 'mean centered on {} '.format(self.params['init_mean'])
        else:
            mean = ''
        return 'gaussian {p}prior of dim {K} with {m}{v}'.format(p=pre, m=mean, v=var, K=self.dim)


class TiltedGaussianPrior(GaussianPrior):
    """kl = 1/2 (|T_y (mu - m_y)| - tau)^2, var_kl = 0."""

    distribution = 'tilted'

    def __init__(self, dim, num_priors=1, init_mean=0, learned_means=False, tau=25, **kw):
        super().__init__(dim, num_priors=num_priors, init_mean=init_mean, learned_means=learned_means,
                         var_dim='scalar', **kw)
        self.tau = tau
        self._tau = float(tau)
        self._mu_star = tau
        self.params['distribution'] = 'tilted'
        self.params['tau'] = tau

    @property
    def mu_star(self):
        return self._mu_star

    def kl(self, mu, log_var, y=None, output_dict=True, var_weighting=1.):
        if var_weighting != 1.:
            logging.debug('var weighting != 1 but tilted gaussian does not care')
        if y is not None and y.ndim == mu.ndim:
            mu, log_var = self._broadcast_over_classes(mu, log_var, y)
        kl, dist, vkl = self._run(mu, log_var, y, 1.)
        if not output_dict:
            return kl
        return {'distance': dist, 'mu_norm': dist.sqrt(), 'var_kl': vkl, 'kl': kl}

    def log_density(self, z, y=None):
        return super().log_density(z, y) - z.norm(dim=-1)

    def __repr__(self):
        m = ' with {} {}means'.format(self.num_priors, 'learned ' if self.learned_means else '') \
            if self.num_priors > 1 else ''
        return 'tilted gaussian {c}prior{m}, tau={tau}'.format(c='conditional ' if self.conditional else '',
                                                               m=m, tau=self.tau)


class UniformWithGaussianTailPrior(GaussianPrior):
    """Uniform on [-tau, tau]^K with Gaussian tails; q(z|x) is uniform of the same variance as N(mu, var)."""

    distribution = 'uniform'

    def __init__(self, dim, num_priors=1, init_mean=0, learned_means=False, tau=5, **kw):
        super().__init__(dim, num_priors=num_priors, init_mean=init_mean, learned_means=learned_means,
                         var_dim='scalar')
        self.tau = tau
        self._tau = float(tau)
        phi_tau = 0.5 * (1 + math.erf(tau / math.sqrt(2)))
        self._alpha = math.log(2 * tau) - math.log(2 * phi_tau - 1)     # -log rho(z) inside [-tau, tau]
        self._alpha_k = float(self._alpha)
        self.params['distribution'] = 'uniform'
        self.params['tau'] = tau

    def kl(self, mu, log_var, y=None, output_dict=True, var_weighting=1.0):
        if y is not None and y.ndim == mu.ndim:
            mu, log_var = self._broadcast_over_classes(mu, log_var, y)
        kl, dist, vkl = self._run(mu, log_var, y, var_weighting)
        if not output_dict:
            return kl
        return {'distance': dist, 'var_kl': vkl, 'kl': kl}

    def log_density(self, z, y=None):
        assert self.conditional ^ (y is None)
        if self.conditional:
            z = z - self.mean.index_select(0, y.reshape(-1)).view(*y.shape, -1)
        c = math.log(2 * math.pi)
        inside = -self._alpha * torch.ones_like(z)
        tail = -c / 2 - z.square() / 2
        return torch.where(z.abs() > self.tau, tail, inside).sum(-1)

    def __repr__(self):
        m = ' with {} {}means'.format(self.num_priors, 'learned ' if self.learned_means else '') \
            if self.num_priors > 1 else ''
        return 'uniform {c}prior{m}, tau={tau}'.format(c='conditional ' if self.conditional else '',
                                                       m=m, tau=self.tau)
