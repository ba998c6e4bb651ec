"""MelGAN-style mel -> waveform generator, drop-in for the reference's
featuresynth/generator/full.py:16-50 (same class name, ctor signature, `main` Sequential
indices and state_dict keys), running on the gfx950 kernels.

Layer plan (reference lines 22-45): ReflectionPad1d(3) + Conv1d(in, 512, 7) + LeakyReLU(0.2);
four [ConvTranspose1d + LeakyReLU + ResidualStack] stages (512->256 k16 s8 p4, 256->128 k16 s8 p4,
128->64 k4 s2 p1, 64->32 k4 s2 p1); Conv1d(32, 1, 7, padding=3) + Tanh.  Pads and activations are
fused into the convolution kernels, so their Sequential slots hold `Fused` placeholders.
"""
import torch
from torch import nn

from .._ops import functional as F_
from .._ops import graph as G_
from ..util.modules import Fused, HipConv1d, HipConvTranspose1d, ResidualStack


class MelGanGenerator(nn.Module):
    _ms_native = True   # train.py: both networks native -> skip-wasted-work + hipGraph path

    def __init__(self, input_size, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.input_size = input_size

        def up(cin, cout, k, s, p):
            return [HipConvTranspose1d(cin, cout, k, s, p, activation="lrelu"),
                    Fused("LeakyReLU(0.2) fused into the transposed conv"),
                    ResidualStack(cout, [1, 3, 9])]

        self.main = nn.Sequential(
            Fused("ReflectionPad1d(3) fused into main.1"),
            HipConv1d(in_channels, 512, 7, 1, padding=3, activation="lrelu", reflect_pad=True),
            Fused("LeakyReLU(0.2) fused into main.1"),
            *up(512, 256, 16, 8, 4),
            *up(256, 128, 16, 8, 4),
            *up(128, 64, 4, 2, 1),
            *up(64, 32, 4, 2, 1),
            HipConv1d(32, 1, 7, 1, 3, activation="tanh"),
            Fused("Tanh fused into main.15"),
        )

    def forward(self, x):
        # one autograd node for the whole stack; iterating self.main layer by layer (as the
        # reference's forward does, full.py:47-50) gives the same values through per-layer nodes
        if not torch.is_grad_enabled():
            # inference (evaluate.py:133, BASELINE config 2): nothing is saved for a backward pass -- the fused atoms run in
            # their inference mode (no stores of the two intermediate activations)
            return G_.gen_forward(x, list(self.parameters()), False)[0]
        return F_.GeneratorFn.apply(x, *self.parameters())

    def forward_layerwise(self, x):
        for layer in self.main:
            x = layer(x)
        return x
