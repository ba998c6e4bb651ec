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


class HipLinear(nn.Module):
    """nn.Linear drop-in (weight (out, in), bias) with a fused activation: the GEMM runs on the row-tile
    conv kernels as a 1-tap conv over a length-1 signal (x (B, in) == (B, in, 1))."""

    def __init__(self, in_features, out_features, bias=True, activation=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.activation = activation
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):  # nn.Linear's default initialisation
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1 / math.sqrt(self.in_features) if self.in_features > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        act = {None: L.ACT_NONE, "lrelu": L.ACT_LRELU, "tanh": L.ACT_TANH}[self.activation]
        B = x.shape[0]
        y = F_.Conv1dFn.apply(x.reshape(B, self.in_features, 1), self.weight.unsqueeze(-1), self.bias,
                              1, 0, 1, 1, L.PAD_ZERO, act)
        return y.reshape(B, self.out_features)


class HipConvTranspose2d(nn.Module):
    """nn.ConvTranspose2d drop-in (weight layout (in, out, kH, kW)) for the two geometries of the stage-1
    spectrogram generator (featuregenerator/upscale.py:84-98): kernel (4, 4) / stride (2, 2) / padding (1, 1)
    and kernel (3, 4) / stride (1, 2) / padding (1, 1), with a fused activation.

    A 2-D transposed conv with these strides is, for every output row, a 1-D transposed conv (k4, s2, p1)
    along the row whose input channels are the channels of the 2 (stride 2) or 3 (stride 1) input rows that
    reach it:  oy = iy*sH - 1 + ky.  Activations travel as LINES -- (B, H, C, W), every line a (C, W)
    signal -- so a layer is: stack the contributing lines along channels (pure data movement), ONE
    ConvTranspose1d launch per output-row phase on B*H lines (the gfx950 row-tile kernels: all arithmetic),
    interleave the phases.  Backward flows through the same two pieces."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, bias=True, activation=None):
        super().__init__()
        kernel_size, stride, padding = tuple(kernel_size), tuple(stride), tuple(padding)
        if (kernel_size, stride, padding) not in (((4, 4), (2, 2), (1, 1)), ((3, 4), (1, 2), (1, 1))):
            raise NotImplementedError("HipConvTranspose2d: geometry %s/%s/%s is not one of the stage-1 "
                                      "generator's" % (kernel_size, stride, padding))
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.activation = activation
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels, *kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.weight.shape[1] * self.weight.shape[2] * self.weight.shape[3]
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def _phases(self):
        """[(output-row phase, [(input-row offset dy, kernel row ky), ...])]: row oy = sH*q + phase takes
        input rows q + dy through kernel rows ky."""
        if self.stride[0] == 2:
            return [(0, [(0, 1), (-1, 3)]), (1, [(1, 0), (0, 2)])]
        return [(0, [(1, 0), (0, 1), (-1, 2)])]

    def forward_lines(self, xl):
        """xl: (B, H, Cin, W) lines -> (B, sH*H, Cout, 2W) lines.  Kernels only: the row stacking, the phase interleave and
        their backward are stream kernels (csrc/lines.hip) inside one autograd node."""
        act = {None: L.ACT_NONE, "lrelu": L.ACT_LRELU, "tanh": L.ACT_TANH}[self.activation]
        phases = tuple((tuple(dy for dy, _ in taps), tuple(ky for _, ky in taps)) for _, taps in self._phases())
        return F_.ConvTranspose2dLinesFn.apply(xl, self.weight, self.bias, phases, act)

    def forward(self, x):
        """x: (B, Cin, H, W) as nn.ConvTranspose2d."""
        return self.forward_lines(x.permute(0, 2, 1, 3).contiguous()).permute(0, 2, 1, 3).contiguous()


class DilatedStack(nn.Module):
    """Reference util/modules.py:79-141: bias-free k-tap convs with the given dilations (zero padding =
    dilation), each followed by `activation`; where a layer keeps the channel count and `residual` is set
    the activation is applied OVER the skip sum (x = activation(z + x), :131-134).  forward can return
    every layer's output as features.  Only LeakyReLU(0.2) (the one the hot path uses) is fused; the
    `activation` callable is kept for the signature and checked against it on first use."""

    def __init__(self, in_channels, channels, kernel_size, dilations, activation, residual=True,
                 groups=None, reflection_padding=False):
        super().__init__()
        if reflection_padding:
            raise NotImplementedError("DilatedStack: reflection padding is not on the stage-1 hot path")
        if groups is None:
            groups = [1] * len(dilations)
        self.reflection_padding = reflection_padding
        self.groups = groups
        self.residual = residual
        self.activation = activation
        self.dilations = dilations
        self.kernel_size = kernel_size
        self.channels = channels
        self.in_channels = in_channels
        self.main = nn.Sequential(*[
            HipConv1d(in_channels if i == 0 else channels, channels, kernel_size, padding=d, dilation=d,
                      groups=groups[i], bias=False)
            for i, d in enumerate(dilations)])
        self._act_checked = False

    def __iter__(self):
        yield from self.main

    def _check_activation(self, device):
        if not self._act_checked:
            probe = torch.tensor([-2.0, 0.0, 3.0], device=device)
            if not torch.allclose(self.activation(probe), torch.nn.functional.leaky_relu(probe, 0.2)):
                raise NotImplementedError("DilatedStack (MI355X build): only LeakyReLU(0.2) is fused")
            self._act_checked = True

    def forward(self, x, return_features=False):
        x = x.reshape(x.shape[0], self.in_channels, -1)
        self._check_activation(x.device)
        features = []
        for layer in self.main:
            if self.residual and layer.out_channels == x.shape[1]:
                z = F_.Conv1dFn.apply(x, layer.weight, None, 1, layer.padding[0], layer.dilation[0],
                                      layer.groups, L.PAD_ZERO, L.ACT_NONE)
                x = F_.AddActFn.apply(z, x, L.ACT_LRELU)
            else:
                x = F_.Conv1dFn.apply(x, layer.weight, None, 1, layer.padding[0], layer.dilation[0],
                                      layer.groups, L.PAD_ZERO, L.ACT_LRELU)
            features.append(x)
        return (features, x) if return_features else x


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


class HipWNConv1d(nn.Module):
    """torch.nn.utils.weight_norm(nn.Conv1d(...)) drop-in for the ResidualAtom: parameters `bias`, `weight_g`
    (out, 1, 1), `weight_v` (out, in, k) in the reference's state_dict order, initialised like weight_norm
    does (v = the conv's default init, g = ||v|| per output channel); the effective weight g * v / ||v|| is
    derived by the HIP weight-norm kernel (and its backward) on every forward."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation = (kernel_size,), (stride,), (padding,), (dilation,)
        ref = torch.empty(out_channels, in_channels, kernel_size)
        nn.init.kaiming_uniform_(ref, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_channels * kernel_size)
        self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        self.weight_g = nn.Parameter(ref.reshape(out_channels, -1).norm(dim=1).reshape(out_channels, 1, 1))
        self.weight_v = nn.Parameter(ref)

    @property
    def weight(self):
        return F_.WeightNormFn.apply(self.weight_v, self.weight_g)


class ResidualAtom(nn.Module):
    """Reference util/modules.py:350-388: x + lrelu(conv(lrelu(conv_dilated(x)))), zero padding =
    dilation, bias on both convs; with `add_weight_norm` both convs are weight-normed (:364-366; the
    MelGAN generator shadows weight_norm with the identity at generator/full.py:12-13, so the hot path
    runs without)."""

    def __init__(self, channels, dilation, add_weight_norm=False):
        super().__init__()
        self.add_weight_norm = add_weight_norm
        self.dilation = dilation
        self.channels = channels
        if add_weight_norm:
            first = HipWNConv1d(channels, channels, 3, 1, dilation=dilation, padding=dilation)
            second = HipWNConv1d(channels, channels, 3, 1, 1)
        else:
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
