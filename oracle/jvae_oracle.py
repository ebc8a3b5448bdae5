"""TEST INFRASTRUCTURE — CPU oracle: a from-scratch restatement of the reference's per-batch joint-CVAE
training step in plain PyTorch-CPU fp32 (the reference's own arithmetic IS PyTorch ops).

NOT product code.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Pinned by tests/test_oracle_golden.py against tests/golden/*.npz, which oracle/gen_golden.py produced by
running the reference itself (moxime/joint-vae @ /root/reference) in the build container.

Design: purely functional.  A model is (spec, P) where `spec` is a small description derived from the
constructor kwargs and `P` is a flat dict {state_dict key -> tensor} that uses the reference's key names
(features.0.weight, encoder.dense_mean.bias, encoder.prior.mean, imager.18.weight, sigma, ...).

Reference anchors (file:line under /root/reference):
  layer DSL ............ module/vae_layers/conv.py:20-86,128-244 + conv-models.ini:11-30
  encoder / sampling ... module/vae_layers/layers.py:230-244,350-403
  forward .............. cvae.py:426-521
  evaluate (train) ..... cvae.py:523-917
  losses ............... module/losses.py:8-27,52-86
  priors ............... module/priors.py:142-148,173-250,252-326,389-408,429-476
  step ................. cvae.py:2424-2461, module/optimizers.py:79-121 (clip_grad_norm_ + Adam, L2 in grad)
"""
import math
import re

import numpy as np
import torch
import torch.nn.functional as F

NAMED_FEATURES = {            # conv-models.ini:13-21
    'conv32': '[x5+2]32-32:2-64-64:2-200x7+0',
    'conv32-': '[x3+1]32-32-32-32:2-64-64-64-64:2-200x7+0',
    'conv32+': '[x5+2]32-32:2-64-64:2-128-128:2-200x3+0',
}
NAMED_UPSAMPLERS = {          # conv-models.ini:25-27
    'deconv32': '[x5+2]64x8+0-64-64:2++1-32-32:2++1-32-!3x5+2',
    'deconv32-': '[x3+1]64x8+0-64-64-64-64:2++1-32-32-32-32:2++1-32-!3x5+2',
    'deconv32+': '[x5+2]128x4+0-128-128:2++1-64-64:2++1-32-32:2++1-32-!3x5+2',
}


# ----------------------------------------------------------------------------------------------- DSL
def _tok(token, upsampler, dflt):
    """One '-' separated token -> dict(kind, c, k, p, s, op).  conv.py:20-86 (conv/deconv tokens only)."""
    d = dict(dflt)
    kind = 'deconv' if upsampler else 'conv'
    t = token
    if upsampler and t.startswith('!'):
        kind = 'conv'
        t = t[1:]
    m = re.match(r'^(\d+)', t)
    if m:
        d['c'] = int(m.group(1))
    m = re.search(r'x(\d+)', t)
    if m:
        d['k'] = int(m.group(1))
    m = re.search(r'\+\+(\d+)', t)
    if m and upsampler:
        d['op'] = int(m.group(1))
    m = re.search(r'(?<!\+)\+(\d+)', t)      # a single '+': padding
    if m:
        d['p'] = int(m.group(1))
    m = re.search(r':(\d+)', t)
    if m:
        d['s'] = int(m.group(1))
    d['kind'] = kind
    return d


def parse_stack(layers_name, in_shape, upsampler):
    """-> list of dict(kind,cin,c,k,p,s,op, h_in,w_in,h,w).  conv.py:128-230 restricted to (de)conv tokens."""
    table = NAMED_UPSAMPLERS if upsampler else NAMED_FEATURES
    s = table.get(layers_name, layers_name)
    dflt = dict(c=32, k=5, p=None, s=1, op=0)
    if s[0] == '[':
        end = s.index(']')
        for t in s[1:end].split('-'):
            if t[0] in 'MAUmau!':
                continue
            dflt = {**dflt, **{k: v for k, v in _tok(t, upsampler, dflt).items() if k != 'kind'}}
        s = s[end + 1:]
    cin, h, w = in_shape
    out = []
    for t in s.split('-'):
        d = _tok(t, upsampler, dflt)
        if d['p'] is None:
            d['p'] = d['k'] // 2 if not upsampler else 0
        if d['kind'] == 'conv':
            d['op'] = 0
            ho = (h + 2 * d['p'] - d['k']) // d['s'] + 1
            wo = (w + 2 * d['p'] - d['k']) // d['s'] + 1
        else:
            ho = (h - 1) * d['s'] - 2 * d['p'] + d['k'] + d['op']
            wo = (w - 1) * d['s'] - 2 * d['p'] + d['k'] + d['op']
        d.update(cin=cin, h_in=h, w_in=w, h=ho, w=wo)
        out.append(d)
        cin, h, w = d['c'], ho, wo
    return out


def find_imager_hw(upsampler, target_hw):
    """Smallest (h, w) input that the upsampler maps to target_hw.  conv.py:108-125."""
    h, w = 1, 1
    while True:
        st = parse_stack(upsampler, (1, h, w), True)
        oh, ow = st[-1]['h'], st[-1]['w']
        if (oh, ow) == tuple(target_hw):
            return h, w
        if oh > target_hw[0] or ow > target_hw[1]:
            raise ValueError('no input shape for ' + upsampler)
        h += int(oh < target_hw[0])
        w += int(ow < target_hw[1])


# ----------------------------------------------------------------------------------------------- spec
def make_spec(input_shape, num_labels, type='cvae', features=None, upsampler=None, encoder=(), decoder=(),
              classifier=(), batch_norm=False, latent_dim=32, latent_sampling=1, test_latent_sampling=None,
              sigma=None, gamma=0., beta=1., output_activation='linear', activation='relu', prior=None,
              optimizer=None, **_):
    assert type == 'cvae' and activation in ('relu', 'leaky')         # cvae.py:46-49: 'leaky' = nn.LeakyReLU() (slope 0.01)
    sp = dict(act=activation, input_shape=tuple(input_shape), C=num_labels, K=latent_dim, L=latent_sampling,
              beta=beta, gamma=gamma, out_act=output_activation,
              enc=list(encoder), dec=list(decoder), clf=list(classifier) if gamma else [],
              prior=dict(prior or {}), opt=dict(optimizer or {}), sigma=dict(sigma or {'value': 1}))
    bn_e = bool(features) and batch_norm in ('encoder', 'both')     # cvae.py:235-240
    bn_d = bool(features) and batch_norm == 'both'
    sp['bn_e'], sp['bn_d'] = bn_e, bn_d
    if features:
        sp['features'] = parse_stack(features, input_shape, False)
        last = sp['features'][-1]
        enc_in = last['c'] * last['h'] * last['w']
    else:
        sp['features'] = None
        enc_in = int(np.prod(input_shape))
    sp['enc_in'] = enc_in
    dec_out = sp['dec'][-1] if sp['dec'] else latent_dim
    if upsampler:
        hw = find_imager_hw(upsampler, input_shape[1:])
        f = hw[0] * hw[1]
        assert dec_out % f == 0
        sp['imager_in'] = (dec_out // f, hw[0], hw[1])          # cvae.py:305-310
        sp['imager'] = parse_stack(upsampler, sp['imager_in'], True)
    else:
        sp['imager'] = None
        sp['imager_in'] = (dec_out,)
    sp['sampled'] = latent_sampling > 1 or beta > 0            # cvae.py:276
    return sp


def param_keys(sp):
    """(key, shape) for every state_dict entry the reference model of this spec has, in its order."""
    out = [('sigma', (1,))]

    def stack(prefix, layers, bn):
        i = 0
        for d in layers:
            if d['kind'] == 'conv':
                out.append((f'{prefix}.{i}.weight', (d['c'], d['cin'], d['k'], d['k'])))
            else:
                out.append((f'{prefix}.{i}.weight', (d['cin'], d['c'], d['k'], d['k'])))
            out.append((f'{prefix}.{i}.bias', (d['c'],)))
            i += 1
            if bn:
                for leaf in ('weight', 'bias', 'running_mean', 'running_var'):
                    out.append((f'{prefix}.{i}.{leaf}', (d['c'],)))
                out.append((f'{prefix}.{i}.num_batches_tracked', ()))
                i += 1
            i += 1                                            # the activation slot
    if sp['features']:
        stack('features', sp['features'], sp['bn_e'])
    d_in = sp['enc_in']
    for j, d in enumerate(sp['enc']):
        out += [(f'encoder.dense_projs.{2 * j}.weight', (d, d_in)), (f'encoder.dense_projs.{2 * j}.bias', (d,))]
        d_in = d
    K, C = sp['K'], sp['C']
    out += [('encoder.dense_mean.weight', (K, d_in)), ('encoder.dense_mean.bias', (K,)),
            ('encoder.dense_log_var.weight', (K, d_in)), ('encoder.dense_log_var.bias', (K,))]
    if sp['sigma'].get('input_dim'):                       # coded sigma: one more head on the trunk (layers.py:296-298)
        out += [('encoder.sigma.weight', (1, d_in)), ('encoder.sigma.bias', (1,))]
    out.append(('encoder.prior.mean', (C, K)))
    vd = sp['prior'].get('var_dim', 'scalar') if sp['prior'].get('distribution', 'gaussian') == 'gaussian' else 'scalar'
    out.append(('encoder.prior._var_parameter', {'scalar': (C,), 'diag': (C, K), 'full': (C, K, K)}[vd]))
    d_in = K
    for j, d in enumerate(sp['dec']):
        out += [(f'decoder.{2 * j}.weight', (d, d_in)), (f'decoder.{2 * j}.bias', (d,))]
        d_in = d
    if sp['imager']:
        stack('imager', sp['imager'], sp['bn_d'])
    else:
        D = int(np.prod(sp['input_shape']))
        out += [('imager.0.weight', (D, d_in)), ('imager.0.bias', (D,))]
    d_in = K
    for j, d in enumerate(sp['clf']):
        out += [(f'classifier.{2 * j}.weight', (d, d_in)), (f'classifier.{2 * j}.bias', (d,))]
        d_in = d
    j = len(sp['clf'])
    out += [(f'classifier.{2 * j}.weight', (C, d_in)), (f'classifier.{2 * j}.bias', (C,))]
    return out


BUFFER_LEAVES = ('running_mean', 'running_var', 'num_batches_tracked')


def trainable(sp, key):
    """Which state entries are nn.Parameters with requires_grad (priors.py:79,105-106,122; layers.py:84-89)."""
    leaf = key.rsplit('.', 1)[-1]
    if leaf in BUFFER_LEAVES:
        return False
    if key == 'sigma':                                   # coded implies learned (layers.py:82-83), but it never gets a gradient
        return bool(sp['sigma'].get('learned', False) or sp['sigma'].get('input_dim'))
    if key == 'encoder.prior.mean':
        return bool(sp['prior'].get('learned_means', False)) and not sp['prior'].get('freeze_means', 0) > 0
    if key == 'encoder.prior._var_parameter':
        return sp['prior'].get('distribution', 'gaussian') == 'gaussian' and sp['prior'].get('var_dim', 'scalar') != 'scalar'
    return True


def init_state(sp, seed=0):
    from .det_init import det_tensor
    P = {}
    for key, shape in param_keys(sp):
        if key == 'sigma':
            sg = sp['sigma']
            if sg.get('is_rmse'):                           # starts at 0, then follows the batch rmse (layers.py:79-80)
                P[key] = torch.zeros(1)
            elif sg.get('input_dim'):                       # coded: log(0) until the first batch sets it
                P[key] = torch.full((1,), -math.inf)
            else:
                v = float(sg.get('value', 1))
                P[key] = torch.full((1,), math.log(v) if sg.get('learned') else v)   # layers.py:84-89
        elif key.endswith('num_batches_tracked'):
            P[key] = torch.zeros((), dtype=torch.int64)
        else:
            P[key] = det_tensor(key, shape, seed)
    for key in P:
        if trainable(sp, key):
            P[key].requires_grad_(True)
    return P


# ----------------------------------------------------------------------------------------------- forward
TAPE = None        # debugging aid: set to a list to record (name, tensor) of every stack intermediate


def _tape(name, t):
    if TAPE is not None:
        if t.requires_grad:
            t.retain_grad()
        TAPE.append((name, t))
    return t


# ---- bf16 emulation of the product's mixed-precision mode (BASELINE configs[4]; the reference has no such mode) ----------------
# BF16_CONV = True makes run_stack() round where the drop-in's bf16 path stores or multiplies bf16 (joint-vae_amd/module/
# vae_layers/conv.py::_forward_b8, jvae_hip/ops_b8.py): a 5x5 (de)convolution - the geometries with bf16 kernels - multiplies
# bf16(input) by bf16(weight) with fp32 accumulation and fp32 bias and stores its output in bf16 (the last convolution of a stack
# that feeds the loss stores fp32); the BatchNorm behind it takes its batch statistics from the fp32 accumulators, normalises
# the STORED bf16 tensor in fp32 and stores bf16 again; the gradients that travel between those layers are bf16 tensors too
# (dgrad output, BatchNorm-backward output; the gradient arriving at an fp32 output is rounded when it becomes an MFMA
# operand).  Everything else - other kernel sizes, dense layers, latent and loss arithmetic, parameter gradients, clip, Adam -
# is fp32, as in the product.  This pins the bf16 model to a MODEL of its arithmetic instead of to a tolerance read off a run.
BF16_CONV = False


class bf16_convs:
    """`with bf16_convs(): ...` - the oracle's conv stacks emulate the product's bf16 mode inside the block."""

    def __enter__(self):
        global BF16_CONV
        self.old, BF16_CONV = BF16_CONV, True

    def __exit__(self, *a):
        global BF16_CONV
        BF16_CONV = self.old


def _rbf(t):
    return t.to(torch.bfloat16).to(torch.float32)


class _StoredBF16(torch.autograd.Function):
    """A tensor kept in bf16: the value is rounded on the way forward, its gradient on the way back."""

    @staticmethod
    def forward(ctx, t):
        return _rbf(t)

    @staticmethod
    def backward(ctx, g):
        return _rbf(g)


class _OperandBF16(torch.autograd.Function):
    """fp32 master weights rounded when they become an MFMA operand; their gradient stays fp32."""

    @staticmethod
    def forward(ctx, t):
        return _rbf(t)

    @staticmethod
    def backward(ctx, g):
        return g


class _GradBF16(torch.autograd.Function):
    """An fp32 output whose incoming gradient is rounded when the backward kernels take it as an operand."""

    @staticmethod
    def forward(ctx, t):
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return _rbf(g)


def _bn_from(stat_src, x, rm, rv, gamma, beta, training, momentum, eps):
    """BatchNorm2d of x with the batch statistics of `stat_src` (the fp32 accumulators the stored tensor x was rounded from);
    the backward is the ordinary BatchNorm backward of x (what the product's kernel computes from the stored tensor)."""
    if not training:
        return F.batch_norm(x, rm, rv, gamma, beta, False, momentum, eps)
    dims = (0, 2, 3)
    n = x.numel() // x.shape[1]
    mean_x, var_x = x.mean(dims), x.var(dims, unbiased=False)
    with torch.no_grad():
        m_s, v_s = stat_src.mean(dims), stat_src.var(dims, unbiased=False)
        rm.mul_(1 - momentum).add_(momentum * m_s)
        rv.mul_(1 - momentum).add_(momentum * v_s * n / max(n - 1, 1))
    mean = mean_x + (m_s - mean_x).detach()          # value of the accumulators' statistics, gradient of x's own
    var = var_x + (v_s - var_x).detach()
    xh = (x - mean.view(1, -1, 1, 1)) * torch.rsqrt(var.view(1, -1, 1, 1) + eps)
    return xh * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)


def run_stack(P, prefix, layers, bn, x, last_act, training=True, momentum=0.1, eps=1e-5, hidden_act='relu'):
    """conv.py:186-230: (de)conv -> [BatchNorm2d] -> ReLU; the LAST activation is `last_act` for upsamplers."""
    i = 0
    b8 = False                       # bf16 emulation: x is a tensor the product holds in bf16
    for li, d in enumerate(layers):
        w, b = P[f'{prefix}.{i}.weight'], P[f'{prefix}.{i}.bias']
        native = BF16_CONV and d['k'] == 5
        acc = None
        if native:
            if not b8:
                x = _StoredBF16.apply(x)                 # to_b8(): the fp32 tensor enters the bf16 layout
            w = _OperandBF16.apply(w)
        if d['kind'] == 'conv':
            x = F.conv2d(x, w, b, stride=d['s'], padding=d['p'])
        else:
            x = F.conv_transpose2d(x, w, b, stride=d['s'], padding=d['p'], output_padding=d['op'])
        if native:
            last_f32 = li == len(layers) - 1 and not bn   # the stack's last convolution writes fp32 for the loss
            acc = x
            x = _GradBF16.apply(x) if last_f32 else _StoredBF16.apply(x)
            b8 = not last_f32
        else:
            b8 = False
        _tape(f'{prefix}.{i}', x)
        i += 1
        if bn:
            if b8:
                x = _bn_from(acc, x, P[f'{prefix}.{i}.running_mean'], P[f'{prefix}.{i}.running_var'],
                             P[f'{prefix}.{i}.weight'], P[f'{prefix}.{i}.bias'], training, momentum, eps)
            else:
                x = F.batch_norm(x, P[f'{prefix}.{i}.running_mean'], P[f'{prefix}.{i}.running_var'],
                                 P[f'{prefix}.{i}.weight'], P[f'{prefix}.{i}.bias'], training, momentum, eps)
            if training:
                P[f'{prefix}.{i}.num_batches_tracked'] += 1
            i += 1
        act = hidden_act if (last_act is None or li < len(layers) - 1) else last_act
        x = _act(x, act)
        if b8:
            x = _StoredBF16.apply(x)                     # the normalised activation is stored (or becomes an operand) in bf16
        x = _tape(f'{prefix}.{i}', x)
        i += 1
    return x


def _act(x, name):
    return {'relu': torch.relu, 'linear': lambda t: t, 'sigmoid': torch.sigmoid, 'leaky': F.leaky_relu}[name](x)


def prior_T(sp, P):
    vp = P['encoder.prior._var_parameter']
    return vp.tril() if vp.ndim == 3 else vp                     # priors.py:142-148


def prior_kl(sp, P, mu, log_var, y, w):
    """priors.py:252-326 (gaussian), 389-408 (tilted), 429-476 (uniform).  y: (N,) int64."""
    dist = sp['prior'].get('distribution', 'gaussian')
    K = sp['K']
    m = P['encoder.prior.mean'].index_select(0, y)
    if dist == 'uniform':
        tau = float(sp['prior'].get('tau', 5))
        phi = 0.5 * (1 + math.erf(tau / math.sqrt(2)))
        alpha = math.log(2 * tau) - math.log(2 * phi - 1)
        c = math.log(2 * math.pi)
        span = 2 * math.sqrt(3) * (0.5 * log_var).exp()
        d = mu - m
        dist2 = d.square()
        a_ = tau * F.hardtanh((d - 0.5 * span) / tau)
        b_ = tau * F.hardtanh((d + 0.5 * span) / tau)
        elogq = -0.5 * log_var - 0.5 * math.log(12)
        nel = (c + dist2 + span.square() / 12) / 2
        nel = nel + (alpha - c / 2) * (b_ - a_) / span
        nel = nel - (b_.pow(3) - a_.pow(3)) / span / 6
        vkl = (elogq + alpha).sum(-1)
        kl = torch.max(elogq.sum(-1) + nel.sum(-1), vkl)
        if w != 1.0:
            kl = kl + (w - 1) * vkl
        return dict(distance=dist2.sum(-1), var_kl=2 * vkl, kl=kl)
    T = prior_T(sp, P).index_select(0, y)
    d = mu - m
    if T.ndim == 3:
        wd = torch.matmul(T, d.unsqueeze(-1)).squeeze(-1)       # priors.py:202-203
        pdiag = T.pow(2).sum(-2)                                 # priors.py:231-233
        logdet_p = -2 * torch.diagonal(T, dim1=-2, dim2=-1).abs().log().sum(-1)
    elif T.ndim == 2:
        wd = d * T
        pdiag = T.pow(2)
        logdet_p = -2 * T.abs().log().sum(-1)
    else:
        wd = d * T.unsqueeze(-1)
        pdiag = T.pow(2).unsqueeze(-1)
        logdet_p = -2 * K * T.log()
    distance = wd.pow(2).sum(-1)
    if dist == 'tilted':
        tau = float(sp['prior'].get('tau', 25))
        return dict(distance=distance, var_kl=torch.zeros_like(distance), kl=0.5 * (distance.sqrt() - tau) ** 2)
    trace = (log_var.exp() * pdiag).sum(-1)
    var_kl = trace - log_var.sum(-1) + logdet_p - K
    return dict(distance=distance, var_kl=var_kl, kl=0.5 * (distance + w * var_kl))


def recon_terms(sp, P, u, x, x_reco, L, N, training):
    """Reconstruction term for every kind of sigma (cvae.py:626-670): -> (wmse_s (L,N), wmse (N,), mse (N,), log_sigma,
    sigma_value_reported).  Kinds: scalar value / learned log (broadcast), `is_rmse` (sigma_n^2 = the sample's own mse,
    averaged over the L draws) and coded (log sigma_n = one more linear head on the encoder trunk)."""
    sg = sp['sigma']
    s = P['sigma']
    with torch.no_grad():
        reported = float(((2 * s).exp() if (sg.get('learned') or sg.get('input_dim') or sg.get('is_log')) else s.pow(2)).mean().sqrt())
    diff2 = lambda sig: ((x_reco[1:] / sig - (x / sig).unsqueeze(0)) ** 2).reshape(L, N, -1).mean(-1)   # losses.py:8-27
    if sg.get('is_rmse'):
        raw = diff2(1.)
        sigma2 = raw.mean(0)                                   # (N,) cvae.py:662-666
        wmse_s = raw / sigma2.unsqueeze(0)
        log_sigma = sigma2.sqrt().log()
        new = None
    elif sg.get('input_dim'):
        s_ = F.linear(u, P['encoder.sigma.weight'], P['encoder.sigma.bias']).view(-1, *([1] * len(sp['input_shape'])))
        sigma2 = (2 * s_).exp().reshape(N)
        wmse_s = diff2(s_.exp())
        log_sigma = s_.reshape(N)
        new = s_.detach().mean(0).reshape(1) if False else s_.detach().mean(tuple(range(s_.dim() - 1)))   # Sigma.update(v=...)
    else:
        sigma_ = s.exp() if sg.get('learned') else s
        log_sigma = s.squeeze() if sg.get('learned') else s.log().squeeze()
        sigma2 = sigma_ ** 2
        wmse_s = diff2(sigma_)
        new = None
    wmse = wmse_s.mean(0)
    mse = wmse * sigma2
    if sg.get('input_dim'):
        # sic (cvae.py:668): wmse (N,) times sigma2_ (N,1,1,1) broadcasts to (N,1,1,N): its mean - the only use of `mse`
        # with a coded sigma - is mean(wmse) * mean(sigma^2), not the mean of the products
        mse = (wmse.mean() * sigma2.mean()).expand(N)
    if training:
        with torch.no_grad():
            rmse = mse.mean().sqrt()
            if new is not None:
                P['sigma'] = new.reshape(P['sigma'].shape).clone().requires_grad_(P['sigma'].requires_grad)
            elif sg.get('is_rmse') or (sg.get('decay') and not sg.get('learned')):
                decay = 1. if sg.get('is_rmse') else sg['decay']
                delta = decay * (sg.get('reach', 1) * rmse - P['sigma'])
                if sg.get('max_step'):
                    delta = delta.clamp(-sg['max_step'], sg['max_step'])
                # IN PLACE, as Sigma.update does (`self.data += delta`, layers.py:168): the division x_reco / sigma saved this
                # very tensor for backward, so the reference's backward divides by the UPDATED sigma (gradient of the
                # reconstruction term = 2 (x_reco - x) / (D sigma_old sigma_new)); golden c2_n8_decay pins it
                P['sigma'].data += delta
    return wmse_s, wmse, mse, log_sigma, reported


def evaluate(sp, P, x, y, eps, kl_var_weighting=1.0, gamma_weighting=1.0, with_beta=True, training=True, alt_prior=None):
    """Training-branch evaluate (cvae.py:523-917) -> (x_reco, y_est, losses, measures, mu, log_var, z).
    alt_prior: {'mean': (1,K), 'T': (1,)} = a NON-conditional prior swapped in for this call (WIM, ft/wim.py:54-70):
    no dictionary terms (dzdist, ld-norm, imut-zy, d-mind), cvae.py:701,746."""
    N = x.shape[0]
    K, L = sp['K'], sp['L']
    D = int(np.prod(sp['input_shape']))
    if sp['features']:
        t = run_stack(P, 'features', sp['features'], sp['bn_e'], x, None, training, hidden_act=sp.get('act', 'relu'))
    else:
        t = x
    u = t.reshape(N, -1)
    for j in range(len(sp['enc'])):
        u = _act(F.linear(u, P[f'encoder.dense_projs.{2 * j}.weight'], P[f'encoder.dense_projs.{2 * j}.bias']), sp.get('act', 'relu'))
    mu = F.linear(u, P['encoder.dense_mean.weight'], P['encoder.dense_mean.bias'])
    log_var = torch.clip(F.linear(u, P['encoder.dense_log_var.weight'], P['encoder.dense_log_var.bias']), -20, 20)
    z = mu + torch.exp(0.5 * log_var) * eps * float(sp['sampled'])       # layers.py:243, eps[0] == 0
    h = z
    for j in range(len(sp['dec'])):
        h = _act(F.linear(h, P[f'decoder.{2 * j}.weight'], P[f'decoder.{2 * j}.bias']), sp.get('act', 'relu'))
    if sp['imager']:
        xr = run_stack(P, 'imager', sp['imager'], sp['bn_d'], h.reshape(-1, *sp['imager_in']), sp['out_act'], training, hidden_act=sp.get('act', 'relu'))
    else:
        xr = _act(F.linear(h, P['imager.0.weight'], P['imager.0.bias']), sp['out_act'])
    x_reco = xr.reshape(L + 1, N, *sp['input_shape'])
    c = z
    nclf = len(sp['clf'])
    for j in range(nclf):
        c = _act(F.linear(c, P[f'classifier.{2 * j}.weight'], P[f'classifier.{2 * j}.bias']), sp.get('act', 'relu'))
    logits = F.linear(c, P[f'classifier.{2 * nclf}.weight'], P[f'classifier.{2 * nclf}.bias'])

    wmse_s, wmse, mse, log_sigma, sigma_reported = recon_terms(sp, P, u, x, x_reco, L, N, training)
    if alt_prior is not None:
        Pa = {'encoder.prior.mean': alt_prior['mean'], 'encoder.prior._var_parameter': alt_prior['T']}
        spa = dict(sp, prior=dict(distribution='gaussian', var_dim='scalar'))
        kd = prior_kl(spa, Pa, mu, log_var, torch.zeros_like(y), kl_var_weighting)
    else:
        kd = prior_kl(sp, P, mu, log_var, y, kl_var_weighting)
    losses = dict(kl=kd['kl'], zdist=kd['distance'], var_kl=kd['var_kl'])
    dic = P['encoder.prior.mean']
    if alt_prior is None:
        dmean = dic.mean(0)
        losses['dzdist'] = (mu - dmean).pow(2).sum(1) + (dic.pow(2).sum(1).mean(0) - dmean.pow(2).sum())
    losses['wmse'] = wmse
    losses['cross_x'] = D * (2 * log_sigma + wmse + math.log(2 * math.pi)) / 2        # cvae.py:773-789, sdim=1
    total = losses['cross_x']
    if sp['gamma']:
        # NB cvae.py:738 hands x_loss the full (L+1, N, C) logits: the eps=0 row is part of the average
        ce = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), y.repeat(L + 1),
                             reduction='none').reshape(L + 1, N).mean(0)                   # losses.py:76-86
        losses['cross_y'] = ce
        cw = gamma_weighting * sp['gamma'] if training else 0
        if cw:
            total = total + cw * ce
    total = total + (sp['beta'] if with_beta else 1.) * losses['kl']
    losses['total'] = total

    with torch.no_grad():
        xpow = x.pow(2).mean().item()
        mse_m = mse.mean().item()
        cd = torch.cdist(dic, dic)
        C = sp['C']
        meas = {'sigma': sigma_reported, 'xpow': xpow, 'mse': mse_m,
                'rmse': math.sqrt(mse_m), 'dB': 10 * math.log10(xpow / mse_m),
                'zdist': kd['distance'].mean().item(), 'var_kl': kd['var_kl'].mean().item(),
                'ld-norm': dic.pow(2).mean().item(),
                'imut-zy': (math.log(C) - 1 / C * torch.exp(-cd.pow(2) / 4).sum(0).log().sum()).item(),
                'd-mind': (cd + 2 * dic.norm(dim=1).max() * torch.eye(C)).min().item()}
        if alt_prior is not None:
            for k in ('ld-norm', 'imut-zy', 'd-mind'):
                meas.pop(k)
    return x_reco, logits[1:].mean(0), losses, meas, mu, log_var, z


def evaluate_all_classes(sp, P, x, eps):
    """evaluate(x) WITHOUT labels in eval mode (cvae.py:548-600,619-762,793-873): every class is tried as the prior
    component; returns (x_reco, y_est, losses with (C,N) entries incl. `iws`, measures).  eps: (Ltest+1, N, K)."""
    N = x.shape[0]
    K, C = sp['K'], sp['C']
    L = eps.shape[0] - 1
    D = int(np.prod(sp['input_shape']))
    t = run_stack(P, 'features', sp['features'], sp['bn_e'], x, None, False, hidden_act=sp.get('act', 'relu')) if sp['features'] else x
    u = t.reshape(N, -1)
    for j in range(len(sp['enc'])):
        u = _act(F.linear(u, P[f'encoder.dense_projs.{2 * j}.weight'], P[f'encoder.dense_projs.{2 * j}.bias']), sp.get('act', 'relu'))
    mu = F.linear(u, P['encoder.dense_mean.weight'], P['encoder.dense_mean.bias'])
    log_var = torch.clip(F.linear(u, P['encoder.dense_log_var.weight'], P['encoder.dense_log_var.bias']), -20, 20)
    z = mu + torch.exp(0.5 * log_var) * eps * float(sp['sampled'])
    h = z
    for j in range(len(sp['dec'])):
        h = _act(F.linear(h, P[f'decoder.{2 * j}.weight'], P[f'decoder.{2 * j}.bias']), sp.get('act', 'relu'))
    if sp['imager']:
        xr = run_stack(P, 'imager', sp['imager'], sp['bn_d'], h.reshape(-1, *sp['imager_in']), sp['out_act'], False, hidden_act=sp.get('act', 'relu'))
    else:
        xr = _act(F.linear(h, P['imager.0.weight'], P['imager.0.bias']), sp['out_act'])
    x_reco = xr.reshape(L + 1, N, *sp['input_shape'])
    c = z
    nclf = len(sp['clf'])
    for j in range(nclf):
        c = _act(F.linear(c, P[f'classifier.{2 * j}.weight'], P[f'classifier.{2 * j}.bias']), sp.get('act', 'relu'))
    logits = F.linear(c, P[f'classifier.{2 * nclf}.weight'], P[f'classifier.{2 * nclf}.bias'])

    wmse_s, wmse, mse, log_sigma, sigma_reported = recon_terms(sp, P, u, x, x_reco, L, N, False)
    log_iws = -D / 2 * (wmse_s + 2 * log_sigma + math.log(2 * math.pi))                 # (L, N)   cvae.py:672-676
    y_all = torch.arange(C).repeat_interleave(N)                                        # class-major (C*N,)
    kd = prior_kl(sp, P, mu.repeat(C, 1), log_var.repeat(C, 1), y_all, 1.0)
    losses = {k2: kd[k1].reshape(C, N) for k1, k2 in (('kl', 'kl'), ('distance', 'zdist'), ('var_kl', 'var_kl'))}
    dic = P['encoder.prior.mean']
    dmean = dic.mean(0)
    losses['dzdist'] = (mu - dmean).pow(2).sum(1) + (dic.pow(2).sum(1).mean(0) - dmean.pow(2).sum())
    losses['wmse'] = wmse
    losses['cross_x'] = D * (2 * log_sigma + wmse + math.log(2 * math.pi)) / 2
    if sp['gamma']:
        log_p = (logits.softmax(-1) + 1e-6).log()                                       # losses.py:60-68
        losses['cross_y'] = -(log_p[1:].mean(0) if L + 1 > 1 else log_p[0]).t()
    losses['total'] = losses['cross_x'].unsqueeze(0) + losses['kl']                     # beta = 1 (with_beta False)
    # importance-weighted bound (cvae.py:793-873); gaussian prior: log p(z|y) = -K/2 log 2pi - mahala/2 - log|S_y|/2
    T = prior_T(sp, P)
    d = z[1:].unsqueeze(1) - dic.unsqueeze(0).unsqueeze(2)                              # (L, C, N, K)
    if T.ndim == 3:
        wd = torch.einsum('cij,lcnj->lcni', T, d)
        logdet = -2 * torch.diagonal(T, dim1=-2, dim2=-1).abs().log().sum(-1)
    elif T.ndim == 2:
        wd = d * T.view(1, C, 1, K)
        logdet = -2 * T.abs().log().sum(-1)
    else:
        wd = d * T.view(1, C, 1, 1)
        logdet = -2 * K * T.log()
    log_p_z_y = -math.log(2 * math.pi) * K / 2 - wd.pow(2).sum(-1) / 2 - logdet.view(1, C, 1) / 2
    if sp['prior'].get('distribution') == 'tilted':
        log_p_z_y = log_p_z_y - z[1:].norm(dim=-1).unsqueeze(1)
    log_inv_q = ((eps[1:] ** 2).sum(-1) + log_var.sum(-1)) / 2 + K / 2 * math.log(2 * math.pi)   # (L, N)
    li = log_iws.unsqueeze(1) + log_p_z_y + log_inv_q.unsqueeze(1)
    rem = li.max(0)[0]
    losses['iws'] = (li - rem).exp().mean(0) + rem                                      # sic (cvae.py:868)
    with torch.no_grad():
        xpow, mse_m = x.pow(2).mean().item(), mse.mean().item()
        cd = torch.cdist(dic, dic)
        meas = {'sigma': sigma_reported, 'xpow': xpow, 'mse': mse_m, 'rmse': math.sqrt(mse_m),
                'dB': 10 * math.log10(xpow / mse_m), 'zdist': losses['zdist'].mean().item(),
                'var_kl': losses['var_kl'].mean().item(), 'ld-norm': dic.pow(2).mean().item(),
                'imut-zy': (math.log(C) - 1 / C * torch.exp(-cd.pow(2) / 4).sum(0).log().sum()).item(),
                'd-mind': (cd + 2 * dic.norm(dim=1).max() * torch.eye(C)).min().item()}
    return x_reco, logits[1:].mean(0), losses, meas


def predict(losses, y_est, method):
    """predict_after_evaluate, cvae.py:938-970 (methods of the cvae type)."""
    if method == 'iws':
        return losses['iws'].argmax(0)
    if method == 'closest':
        return losses['zdist'].argmin(0)
    if method == 'esty':
        return y_est.argmax(-1)
    if method == 'loss':
        return losses['total'].argmin(0)
    raise ValueError(method)


def ood_scores(losses, C, methods):
    """batch_dist_measures, cvae.py:972-1085, for the cvae type's OOD methods (suffixes -2s / -a-x-y name the
    thresholding applied later, not the score)."""
    out = {}
    for m_ in methods:
        m = m_[:-3] if m_.endswith('-2s') else m_
        m = m.split('-')[0] if '-a-' in m else m
        if m == 'elbo':
            v = (-losses['total']).max(0)[0]
        elif m == 'iws':
            mx = losses['iws'].max(0)[0]
            v = (losses['iws'] - mx).exp().sum(0).log() + mx + math.log(C)
        elif m in ('soft', 'softkl'):
            v = (-losses['kl']).softmax(0).max(0)[0]
        elif m in ('zdist', 'kl'):
            v = (-losses[m]).max(0)[0]
        elif m == 'mse':
            v = -losses['cross_x']
        else:
            raise ValueError(m_)
        out[m_] = v
    return out


# ----------------------------------------------------------------------------------------------- step
class AdamState:
    """torch.optim.Adam(lr, betas=(.9,.999), eps=1e-8, weight_decay=L2-in-grad) + clip_grad_norm_
    restated by hand (optimizers.py:39-47,79-81,120-121)."""

    def __init__(self, sp):
        o = sp['opt']
        self.lr = o.get('lr') or 1e-3
        self.wd = o.get('weight_decay', 0.)
        self.clip = o.get('grad_clipping')
        self.t = 0
        self.m, self.v = {}, {}

    def step(self, P, grads):
        gn = torch.sqrt(sum(g.double().pow(2).sum() for g in grads.values())).float()
        coef = 1.0
        if self.clip:
            coef = torch.clamp(self.clip / (gn + 1e-6), max=1.0)
        self.t += 1
        b1, b2 = 0.9, 0.999
        with torch.no_grad():
            for k, g in grads.items():
                g = g * coef + self.wd * P[k]
                m = self.m.setdefault(k, torch.zeros_like(g))
                v = self.v.setdefault(k, torch.zeros_like(g))
                m.mul_(b1).add_(g, alpha=1 - b1)
                v.mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (v.sqrt() / math.sqrt(1 - b2 ** self.t)).add_(1e-8)
                P[k].addcdiv_(m, denom, value=-self.lr / (1 - b1 ** self.t))
        return float(gn)


def wim_step(sp, P, opt, x_in, y_in, eps_in, x_mix, eps_mix, alt_prior, alpha):
    """WIM fine-tuning step (ft/wim.py:215-255 + ft/job.py:380-399): original prior on the in-distribution batch,
    alternate prior on the mixture batch, L = total_in.mean() + alpha * total_mix.mean(), backward, Adam step
    (the reference clips AFTER stepping, i.e. without effect)."""
    for p in P.values():
        if p.requires_grad:
            p.grad = None
    o_in = evaluate(sp, P, x_in, y_in, eps_in, with_beta=True, training=True)
    o_mix = evaluate(sp, P, x_mix, torch.zeros_like(y_in), eps_mix, with_beta=True, training=True, alt_prior=alt_prior)
    L = o_in[2]['total'].mean() + alpha * o_mix[2]['total'].mean()
    L.backward()
    grads = {k: p.grad for k, p in P.items() if p.requires_grad and p.grad is not None}
    clip, opt.clip = opt.clip, None
    gn = opt.step(P, grads)
    opt.clip = clip
    return o_in, o_mix, float(L.detach()), grads, gn


def train_step(sp, P, opt, x, y, eps, kl_var_weighting=1.0, gamma_weighting=1.0):
    """cvae.py:2424-2461: evaluate -> total.mean().backward() -> clip -> Adam.  Returns (losses, measures, grads)."""
    for p in P.values():
        if p.requires_grad:
            p.grad = None
    out = evaluate(sp, P, x, y, eps, kl_var_weighting, gamma_weighting, with_beta=True, training=True)
    losses = out[2]
    losses['total'].mean().backward()
    grads = {k: p.grad for k, p in P.items() if p.requires_grad and p.grad is not None}
    gn = opt.step(P, grads)
    return out, grads, gn
