"""Small layer helpers (API of the reference's module/vae_layers/misc.py:6-39), executing on HIP kernels."""
import torch
from torch import nn

from jvae_hip import ops


def onehot_encoding(y, C):
    """Integer labels (...,) -> float one-hot (..., C).  Pure indexing (no arithmetic): a scatter of ones."""
    out = torch.zeros(*y.shape, C, device=y.device)
    return out.scatter_(-1, y.unsqueeze(-1), 1)


class HipReLU(nn.ReLU):
    def forward(self, x):
        return ops.act(x, ops.RELU)


class HipSigmoid(nn.Sigmoid):
    def forward(self, x):
        return ops.act(x, ops.SIGMOID)


class HipIdentity(nn.Identity):
    def forward(self, x):
        return ops.act(x, ops.IDENT)


class HipLeakyReLU(nn.LeakyReLU):
    """nn.LeakyReLU() as the reference builds it for activation='leaky' (misc.py:24-27, conv.py:218-219: no arguments, i.e. negative
    slope 0.01).  The kernels carry that slope as a constant; another one is refused."""

    def __init__(self, negative_slope=0.01, inplace=False):
        if negative_slope != 0.01:
            raise NotImplementedError('leaky ReLU kernels are built for negative_slope = 0.01 (nn.LeakyReLU default)')
        super().__init__(negative_slope, inplace)

    def forward(self, x):
        return ops.act(x, ops.LEAKY)


activation_layers = {'linear': HipIdentity, 'sigmoid': HipSigmoid, 'relu': HipReLU, 'leaky': HipLeakyReLU}

ACT_OF_MODULE = {HipReLU: ops.RELU, HipSigmoid: ops.SIGMOID, HipIdentity: ops.IDENT, HipLeakyReLU: ops.LEAKY}


def _no_activation(a):
    return a


class Reshape(nn.Module):
    def __init__(self, output_shape):
        super().__init__()
        self.shape = output_shape

    def forward(self, x):
        return x.view(-1, *self.shape)
