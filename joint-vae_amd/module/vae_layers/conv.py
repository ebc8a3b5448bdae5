"""Layer-string DSL -> (de)convolution stacks that run on the gfx950 HIP kernels.

Same public surface as the reference's module/vae_layers/conv.py (`build_de_conv_layers`,
`find_input_shape`, `parse_conv_layer_name`, `conv_layer_name`, `features_dict`, `upsampler_dict`; reference
lines 20-86, 89-105, 108-125, 128-244) and the same `state_dict` layout (conv at index 3i, BatchNorm2d at
3i+1, activation at 3i+2), but the returned container executes conv -> BatchNorm(batch stats) -> ReLU through
libjvae_hip.so instead of ATen/MIOpen.

DSL recap (conv-models.ini): `[defaults]tok-tok-...`; a token is `C x K + P : S` (channels, kernel,
padding, stride), upsamplers add `++OP` (output padding) and `!C...` (a plain convolution inside a
transposed stack).  Pooling (`M`/`A`) and nearest up-sampling (`U`) tokens build HipPool2d / HipUpsamplingNearest2d
(pool.hip); an `output_distribution='categorical'` decoder ends in 256 x C channels + Reshape (reference lines 180-185,
228-230).  torchvision resnet* / densenet* feature extractors (pretrained downloads) raise.
"""
import configparser
import logging
import os
import re

from torch import nn

from jvae_hip import ops, ops_b8
from .misc import activation_layers, Reshape, ACT_OF_MODULE

_ini = configparser.ConfigParser()
_ini.read(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'conv-models.ini'))
features_dict = dict(_ini['features'])
upsampler_dict = dict(_ini['upsampler'])

_FIELD_RE = {'out_channels': r'^(\d+)', 'kernel_size': r'x(\d+)', 'output_padding': r'\+\+(\d+)',
             'padding': r'(?<!\+)\+(\d+)(?!\d*\+)', 'stride': r':(\d+)'}


def parse_conv_layer_name(s, ltype='conv', out_channels=32, kernel_size=5, padding='*', stride=None,
                          output_padding=0, activation='relu', output_activation='linear', where='input'):
    """One DSL token -> dict(ltype, out_channels, kernel_size, padding, stride[, output_padding]).

    Keyword arguments are the defaults for fields the token leaves out (the `[...]` prefix of a layer
    string is parsed with this same function and fed back in).  padding '*' = k // 2 for input-side
    convolutions, 0 otherwise.
    """
    upsampling_side = where == 'output'
    if upsampling_side:
        ltype = 'deconv'
    head = s[:1].lower()
    body = s
    if head in ('a', 'm'):
        ltype, body = head + 'pooling', s[1:]
    elif head == 'u':
        ltype, body = 'upsampler', s[1:]
    conv_inside_deconv = upsampling_side and body.startswith('!')
    if conv_inside_deconv:
        body = body[1:]

    found = {}
    for field, pattern in _FIELD_RE.items():
        m = re.search(pattern, body)
        if m:
            found[field] = int(m.group(1))

    p = {'ltype': ltype, 'kernel_size': found.get('kernel_size', kernel_size),
         'padding': found.get('padding', padding), 'stride': found.get('stride', stride)}
    if ltype in ('conv', 'deconv'):
        p['out_channels'] = found.get('out_channels', out_channels)
    if ltype == 'deconv' and not conv_inside_deconv:
        p['output_padding'] = found.get('output_padding', output_padding)
    if conv_inside_deconv:
        p['ltype'] = 'conv'
    if p['padding'] == '*':
        # the reference resolves '*' against the side-wide layer type, so a `!` conv on the
        # upsampling side gets 0 as well
        p['padding'] = p['kernel_size'] // 2 if (ltype == 'conv') else 0
    if p['stride'] is None and p['ltype'].endswith('conv'):
        p['stride'] = 1
    return p


def conv_layer_name(conv_layer):
    """Canonical token of a built layer (used for the stack's `.name` when it is not a named net)."""
    if isinstance(conv_layer, (nn.Conv2d, nn.ConvTranspose2d)):
        k, p, s = conv_layer.kernel_size[0], conv_layer.padding[0], conv_layer.stride[0]
        tok = f'{conv_layer.out_channels}x{k}'
        if p != k // 2:
            tok += f'+{p}'
        if s != 1:
            tok += f':{s}'
        return tok
    if isinstance(conv_layer, HipPool2d):
        tok = '{}x{}'.format(conv_layer.letter, conv_layer.kernel_size)
        if conv_layer.stride != conv_layer.kernel_size:
            tok += ':{}'.format(conv_layer.stride)
        return tok
    if isinstance(conv_layer, HipUpsamplingNearest2d):
        return 'u:{}'.format(conv_layer.scale_factor)
    raise NotImplementedError(type(conv_layer).__name__)


class HipPool2d(nn.Module):
    """nn.MaxPool2d / nn.AvgPool2d of the layer DSL (tokens `M` / `A`, reference conv.py:201-206) on the HIP kernels."""

    def __init__(self, letter, kernel_size, stride=None, padding=0):
        super().__init__()
        self.letter = letter.upper()
        self.kernel_size, self.stride, self.padding = int(kernel_size), int(stride or kernel_size), int(padding)

    def forward(self, x):
        return ops.pool2d(x, self.kernel_size, self.stride, self.padding, ops.POOL_MAX if self.letter == 'M' else ops.POOL_AVG)

    def extra_repr(self):
        return f'{self.letter}: kernel_size={self.kernel_size}, stride={self.stride}, padding={self.padding}'


class HipUpsamplingNearest2d(nn.Module):
    """nn.UpsamplingNearest2d(scale_factor) of the layer DSL (token `U`, reference conv.py:208-212)."""

    def __init__(self, scale_factor):
        super().__init__()
        self.scale_factor = scale_factor

    def forward(self, x):
        return ops.upsample_nearest(x, self.scale_factor)

    def extra_repr(self):
        return f'scale_factor={self.scale_factor}'


class HipConv2d(nn.Conv2d):
    """nn.Conv2d parameters (same init / state_dict), forward on the HIP implicit-GEMM kernels."""

    def _spec(self):
        sp = self.__dict__.get('_conv_spec')
        if sp is None:
            sp = ops.ConvSpec(self.in_channels, self.out_channels, self.kernel_size[0], self.stride[0], self.padding[0])
            self.__dict__['_conv_spec'] = sp
        return sp

    def forward(self, x, dead_bias=False, stats_out=None, aff=None):
        return ops.conv2d(x, self.weight, self.bias, self._spec(), dead_bias, stats_out, aff)


class HipConvTranspose2d(nn.ConvTranspose2d):
    def _spec(self):
        sp = self.__dict__.get('_conv_spec')
        if sp is None:
            sp = ops.ConvSpec(self.in_channels, self.out_channels, self.kernel_size[0], self.stride[0], self.padding[0],
                              self.output_padding[0], transposed=True)
            self.__dict__['_conv_spec'] = sp
        return sp

    def forward(self, x, output_size=None, dead_bias=False, stats_out=None, aff=None):
        return ops.conv2d(x, self.weight, self.bias, self._spec(), dead_bias, stats_out, aff)


class HipBatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d state; forward = batch-statistics kernel (+ the following ReLU when fused by the stack)."""

    sync_world = 1          # > 1: statistics over all data-parallel ranks (ClassificationVariationalNetwork.set_sync_batchnorm)
    sync_group = None

    def defer(self, x, relu=False, ext=None):
        """Statistics + coefficients only; the consumer convolution applies the normalisation (ops.batchnorm_defer)."""
        return ops.batchnorm_defer(x, self.weight, self.bias, self.running_mean, self.running_var,
                                   self.num_batches_tracked, self.training, relu, self.momentum, self.eps, ext)

    def forward(self, x, relu=False, ext=None):
        if self.training and self.sync_world > 1:
            return ops.sync_batchnorm_act(x, self.weight, self.bias, self.running_mean, self.running_var,
                                          self.num_batches_tracked, relu, self.momentum, self.eps, self.sync_world,
                                          self.sync_group)
        return ops.batchnorm_act(x, self.weight, self.bias, self.running_mean, self.running_var,
                                 self.num_batches_tracked, self.training, relu, self.momentum, self.eps, ext)


class HipConvStack(nn.Sequential):
    """nn.Sequential whose forward fuses BatchNorm2d with the activation that follows it and lets the producing
    convolution's epilogue compute the batch statistics.

    compute_dtype 'bf16' (config 5 of BASELINE.json; no counterpart in the fp32 reference): activations between the
    layers are bf16 in the B8 layout (jvae_hip/ops_b8.py), convolutions run on the bf16 matrix cores with fp32
    accumulation from the fp32 master weights; the stack still takes and returns fp32 NCHW tensors.  Layers whose
    geometry has no bf16 kernel at all (the 3x3 / 4x4 / 7x7 / 8x8 heads) stay on the fp32 kernels."""

    compute_dtype = 'fp32'
    # BatchNorm+ReLU applied by the consuming convolution where its kernels can (fp32 path); JVAE_DEFER_BN=0: A/B switch
    defer_batchnorm = os.environ.get('JVAE_DEFER_BN', '1') != '0'

    def forward(self, x):
        if self.compute_dtype == 'bf16':
            return self._forward_b8(x)
        mods = list(self)
        i = 0
        ext = None
        aff = None           # (scale, shift, relu) of a BatchNorm deferred into the next convolution
        while i < len(mods):
            m = mods[i]
            if isinstance(m, HipBatchNorm2d) and i + 1 < len(mods) and type(mods[i + 1]) in ACT_OF_MODULE \
                    and ACT_OF_MODULE[type(mods[i + 1])] in ops.BN_ACT:
                relu = ops.BN_ACT[ACT_OF_MODULE[type(mods[i + 1])]]      # 0 none, 1 ReLU, 2 leaky ReLU: fused into the BatchNorm kernels
                nxt = mods[i + 2] if i + 2 < len(mods) else None
                if self.defer_batchnorm and isinstance(nxt, (HipConv2d, HipConvTranspose2d)) \
                        and not (m.training and m.sync_world > 1) \
                        and ops.conv_affine_ok(nxt._spec(), x.shape[0], x.shape[2], x.shape[3]):
                    # the normalised activation is never materialised: the next convolution (forward and weight
                    # gradient) applies fmaf(x, scale, shift) + ReLU while it stages its input
                    x, aff = m.defer(x, relu=relu, ext=ext)
                else:
                    x = m(x, relu=relu, ext=ext)
                ext = None
                i += 2
                continue
            if isinstance(m, (HipConv2d, HipConvTranspose2d)) and i + 1 < len(mods) \
                    and isinstance(mods[i + 1], HipBatchNorm2d) and mods[i + 1].training:
                ext = {} if mods[i + 1].sync_world <= 1 else None      # synchronised BN reduces its own sums
                # BatchNorm removes the channel mean: d(loss)/d(bias) == 0 exactly
                x = m(x, dead_bias=True, stats_out=ext, aff=aff)
            elif isinstance(m, (HipConv2d, HipConvTranspose2d)):
                ext = None
                x = m(x, aff=aff)
            else:
                ext = None
                x = m(x)
            aff = None
            i += 1
        return x

    def _forward_b8(self, x):
        mods = list(self)
        if any(type(m).__name__ == 'HipLeakyReLU' for m in mods):
            raise NotImplementedError("the bf16 mode (no counterpart in the reference) has ReLU kernels only: activation='leaky' needs fp32")
        convs = [k for k, m in enumerate(mods) if isinstance(m, (HipConv2d, HipConvTranspose2d))]
        i = 0
        ext = None
        aff = None           # (scale, shift, relu) of a BatchNorm deferred into the next bf16 convolution
        channels = x.shape[1]
        while i < len(mods):
            m = mods[i]
            b8 = ops_b8.is_b8(x)
            if isinstance(m, (HipConv2d, HipConvTranspose2d)):
                spec = m._spec()
                N = x.shape[0]
                H, W = (x.shape[2], x.shape[3])
                native = ops_b8.native_mask(spec, N, H, W) != 0
                bn_next = i + 1 < len(mods) and isinstance(mods[i + 1], HipBatchNorm2d)
                fused = bn_next and mods[i + 1].training and mods[i + 1].sync_world <= 1
                ext = {} if fused else None
                dead = bn_next and mods[i + 1].training
                if native:
                    if not b8:
                        x = ops_b8.to_b8(x)
                    # the last convolution of the stack writes fp32 NCHW directly (it feeds the loss / the dense heads)
                    last = i == convs[-1] and not bn_next and not (spec.transposed and spec.s == 2)
                    x = ops_b8.conv2d(x, m.weight, m.bias, spec, dead, ext, out_f32=last, aff=aff)
                else:
                    if b8:
                        x = ops_b8.from_b8(x, channels)
                    x = ops.conv2d(x, m.weight, m.bias, spec, dead, ext)
                aff = None
                channels = spec.cout
                i += 1
                continue
            if isinstance(m, HipBatchNorm2d):
                relu = False
                step = 1
                if i + 1 < len(mods) and type(mods[i + 1]) in ACT_OF_MODULE \
                        and ACT_OF_MODULE[type(mods[i + 1])] in (ops.RELU, ops.IDENT):
                    relu = ACT_OF_MODULE[type(mods[i + 1])] == ops.RELU
                    step = 2
                nxt = mods[i + step] if i + step < len(mods) else None
                if b8 and not (m.training and m.sync_world > 1) and self.defer_batchnorm \
                        and isinstance(nxt, (HipConv2d, HipConvTranspose2d)) \
                        and ops_b8.conv_affine_ok(nxt._spec(), x.shape[0], x.shape[2], x.shape[3]):
                    # normalisation + ReLU applied by the next bf16 convolution while it stages this tensor
                    x, aff = ops_b8.batchnorm_defer(x, channels, m.weight, m.bias, m.running_mean, m.running_var,
                                                    m.num_batches_tracked, m.training, relu, m.momentum, m.eps, ext)
                elif b8 and not (m.training and m.sync_world > 1):
                    x = ops_b8.batchnorm_act(x, channels, m.weight, m.bias, m.running_mean, m.running_var,
                                             m.num_batches_tracked, m.training, relu, m.momentum, m.eps, ext)
                elif b8:                  # train mode, statistics over all data-parallel ranks: bf16 kernels of their own
                    x = ops_b8.sync_batchnorm_act(x, channels, m.weight, m.bias, m.running_mean, m.running_var,
                                                  m.num_batches_tracked, relu, m.momentum, m.eps, m.sync_world, m.sync_group)
                else:
                    x = m(x, relu=relu, ext=ext)
                ext = None
                i += step
                continue
            ext = None
            if b8 and type(m) in ACT_OF_MODULE and ACT_OF_MODULE[type(m)] == ops.RELU:
                x = ops_b8.relu(x)
            elif b8 and type(m) in ACT_OF_MODULE and ACT_OF_MODULE[type(m)] == ops.IDENT:
                pass
            else:
                if b8:
                    x = ops_b8.from_b8(x, channels)
                x = m(x)
            i += 1
        if ops_b8.is_b8(x):
            x = ops_b8.from_b8(x, channels)
        return x


def build_de_conv_layers(input_shape, layers_name, batch_norm=False, where='input', activation='relu',
                         output_activation='linear', output_distribution='gaussian', pretrained_dict=None):
    """Build the encoder-side (`where='input'`) or decoder-side (`where='output'`) convolution stack.

    Returns an nn.Sequential-compatible module with `.name .output_shape .input_shape .shapes`.
    """
    if where == 'input' and layers_name.startswith('resnet'):
        raise NotImplementedError('torchvision resnet/densenet features are outside the native-kernel contract')

    table = features_dict if where == 'input' else upsampler_dict
    net_name = layers_name if layers_name in table else None
    spec = table.get(layers_name, layers_name)
    if isinstance(input_shape, int):
        input_shape = (input_shape, 1, 1)

    defaults = {}
    if spec.startswith('['):
        close = spec.index(']')
        for tok in spec[1:close].split('-'):
            d = parse_conv_layer_name(tok, where=where)
            defaults[d.pop('ltype')] = d
        spec = spec[close + 1:]

    channels, h, w = input_shape
    modules, tokens, shapes = [], [], [tuple(input_shape)]
    last_act = None
    toks = spec.split('-')
    for n, tok in enumerate(toks):
        kind = parse_conv_layer_name(tok, where=where)['ltype']
        p = parse_conv_layer_name(tok, **defaults.get(kind, {}), where=where)
        kind = p.pop('ltype')
        if kind.endswith('pooling') or kind == 'upsampler':
            # no parameters, no BatchNorm / activation behind them (reference conv.py:201-222)
            if kind == 'upsampler':
                layer = HipUpsamplingNearest2d(p['stride'])
                h, w = int(h * p['stride']), int(w * p['stride'])
            else:
                k, pad = p['kernel_size'], p['padding']
                s = p['stride'] or k
                layer = HipPool2d(kind[0], k, s, pad)
                h, w = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
            modules.append(layer)
            tokens.append(conv_layer_name(layer))
            shapes.append((channels, h, w))
            continue
        if kind not in ('conv', 'deconv'):
            raise NotImplementedError(f'layer token {tok!r} ({kind}) is outside the native-kernel contract')
        if where == 'output' and n == len(toks) - 1 and output_distribution == 'categorical':
            p['out_channels'] *= 256
        k, pad, s = p['kernel_size'], p['padding'], p['stride']
        if kind == 'conv':
            layer = HipConv2d(channels, **p)
            h, w = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
        else:
            layer = HipConvTranspose2d(channels, **p)
            op = p['output_padding']
            h, w = (h - 1) * s - 2 * pad + k + op, (w - 1) * s - 2 * pad + k + op
        channels = p['out_channels']
        modules.append(layer)
        if batch_norm:
            modules.append(HipBatchNorm2d(channels))
        modules.append(activation_layers[activation]())
        last_act = len(modules) - 1
        tokens.append(conv_layer_name(layer))
        shapes.append((channels, h, w))

    out_channels = (channels,)
    if where == 'output':
        modules[last_act] = activation_layers[output_activation]()
        if output_distribution == 'categorical':
            modules.append(Reshape((256, channels // 256, h, w)))
            out_channels = (256, channels // 256)
    stack = HipConvStack(*modules)
    stack.name = net_name or '-'.join(tokens)
    stack.output_shape = (*out_channels, h, w)
    stack.input_shape = input_shape
    stack.shapes = shapes

    if pretrained_dict:
        stack.load_state_dict(pretrained_dict)
        logging.debug('Pretrained conv layers for %s', where)
        for prm in stack.parameters():
            prm.requires_grad_(False)
    return stack


def find_input_shape(layers_name, wanted_output_shape, input_shape=(1, 1)):
    """Smallest spatial input (h, w) an upsampler maps onto `wanted_output_shape` (grown one step at a time)."""
    h, w = input_shape
    wanted = tuple(wanted_output_shape)
    while True:
        got = tuple(_upsampler_out_hw(layers_name, h, w))
        if got == wanted:
            logging.debug('Found input_shape for %s: %s, %s', layers_name, h, w)
            return (h, w)
        if got[0] > wanted[0] or got[1] > wanted[1]:
            raise ValueError('Did not find an input shape yielding output size ({}, {}) for {}'.format(*wanted, layers_name))
        h += int(got[0] < wanted[0])
        w += int(got[1] < wanted[1])


def _upsampler_out_hw(layers_name, h, w):
    """Shape inference only (no parameters are allocated)."""
    spec = upsampler_dict.get(layers_name, layers_name)
    defaults = {}
    if spec.startswith('['):
        close = spec.index(']')
        for tok in spec[1:close].split('-'):
            d = parse_conv_layer_name(tok, where='output')
            defaults[d.pop('ltype')] = d
        spec = spec[close + 1:]
    for tok in spec.split('-'):
        kind = parse_conv_layer_name(tok, where='output')['ltype']
        p = parse_conv_layer_name(tok, **defaults.get(kind, {}), where='output')
        k, pad, s = p['kernel_size'], p['padding'], p['stride']
        if p['ltype'] == 'conv':
            h, w = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
        elif p['ltype'] == 'deconv':
            op = p['output_padding']
            h, w = (h - 1) * s - 2 * pad + k + op, (w - 1) * s - 2 * pad + k + op
        elif p['ltype'] == 'upsampler':
            h, w = int(h * s), int(w * s)
        else:
            raise NotImplementedError(tok)
    return h, w
