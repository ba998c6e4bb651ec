"""Building blocks of the hot path, same names / constructor signatures / state_dict keys as the
reference's featuresynth/util/modules.py (ResidualAtom :350-388, ResidualStack :391-405,
zero_grad :38-40, freeze/unfreeze :43-56), backed by the gfx950 kernels.

Every conv-bearing module's class name contains "Conv" and exposes .weight / .bias
nn.Parameters, because the reference's weights_init (experiment/init.py:3-9) dispatches on
`m.__class__.__name__`.
"""
import math

import torch
from torch import nn

from .._ops import functional as F_
from .._ops import lib as L


class HipConv1d(nn.Module):
    """nn.Conv1d drop-in (same ctor args, same parameter shapes and default init) with an
    optionally fused reflection pad in front and activation behind."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, activation=None, reflect_pad=False):
        super().__init__()
        if in_channels % groups or out_channels % groups:
            raise ValueError("in_channels and out_channels must be divisible by groups")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = (kernel_size,), (stride,), (padding,)
        self.dilation, self.groups = (dilation,), groups
        self.activation = activation            # None | 'lrelu' | 'tanh'
        self.reflect_pad = reflect_pad
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):  # nn.Conv1d's default initialisation
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.weight.shape[1] * self.weight.shape[2]
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def _act(self):
        return {None: L.ACT_NONE, "lrelu": L.ACT_LRELU, "tanh": L.ACT_TANH}[self.activation]

    def forward(self, x):
        return F_.Conv1dFn.apply(x, self.weight, self.bias, self.stride[0], self.padding[0],
                                 self.dilation[0], self.groups,
                                 L.PAD_REFLECT if self.reflect_pad else L.PAD_ZERO, self._act())

    def extra_repr(self):
        return "%d, %d, kernel_size=%d, stride=%d, padding=%d%s, dilation=%d, groups=%d, act=%s" % (
            self.in_channels, self.out_channels, self.kernel_size[0], self.stride[0], self.padding[0],
            " (reflect)" if self.reflect_pad else "", self.dilation[0], self.groups, self.activation)


class HipConvTranspose1d(nn.Module):
    """nn.ConvTranspose1d drop-in (weight layout (in, out, k)) with a fused activation."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True,
                 activation=None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = (kernel_size,), (stride,), (padding,)
        self.activation = activation
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.weight.shape[1] * self.weight.shape[2]
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        act = {None: L.ACT_NONE, "lrelu": L.ACT_LRELU, "tanh": L.ACT_TANH}[self.activation]
        return F_.ConvTranspose1dFn.apply(x, self.weight, self.bias, self.stride[0], self.padding[0], act)

    def extra_repr(self):
        return "%d, %d, kernel_size=%d, stride=%d, padding=%d, act=%s" % (
            self.in_channels, self.out_channels, self.kernel_size[0], self.stride[0], self.padding[0],
            self.activation)


class Fused(nn.Module):
    """Parameter-free placeholder that keeps the reference's nn.Sequential indices (and therefore
    its state_dict keys) for a pad / activation layer that is fused into the neighbouring conv."""

    def __init__(self, what):
        super().__init__()
        self.what = what

    def forward(self, x):
        return x

    def extra_repr(self):
        return self.what


class ResidualAtom(nn.Module):
    """Reference util/modules.py:350-388: x + lrelu(conv(lrelu(conv_dilated(x)))), zero padding =
    dilation, bias on both convs; `add_weight_norm` is accepted (the reference shadows
    weight_norm with the identity at generator/full.py:12-13 for this model)."""

    def __init__(self, channels, dilation, add_weight_norm=False):
        super().__init__()
        if add_weight_norm:
            raise NotImplementedError("weight-normed ResidualAtom is not on the stage-2 hot path")
        self.add_weight_norm = add_weight_norm
        self.dilation = dilation
        self.channels = channels
        first = HipConv1d(channels, channels, 3, 1, dilation=dilation, padding=dilation,
                          activation="lrelu")
        second = HipConv1d(channels, channels, 3, 1, 1, activation="lrelu")
        self.main = nn.Sequential(first, second)

    def forward(self, x):
        a, b = self.main[0], self.main[1]
        return F_.ResidualAtomFn.apply(x, a.weight, a.bias, b.weight, b.bias, self.dilation)


class ResidualStack(nn.Module):
    """Reference util/modules.py:391-405 (the 1/3/9 dilations are hard-coded there, :397-399)."""

    def __init__(self, channels, dilations, add_weight_norm=False):
        super().__init__()
        self.dilations = dilations
        self.channels = channels
        self.main = nn.Sequential(
            ResidualAtom(channels, 1, add_weight_norm),
            ResidualAtom(channels, 3, add_weight_norm),
            ResidualAtom(channels, 9, add_weight_norm),
        )

    def forward(self, x):
        for layer in self.main:
            x = layer(x)
        return x


def zero_grad(*optims):
    """Reference util/modules.py:38-40."""
    for optim in optims:
        optim.zero_grad()


def set_requires_grad(x, requires_grad):
    if isinstance(x, nn.Module):
        x = [x]
    for item in x:
        for p in item.parameters():
            p.requires_grad = requires_grad


def freeze(x):
    set_requires_grad(x, False)


def unfreeze(x):
    set_requires_grad(x, True)
