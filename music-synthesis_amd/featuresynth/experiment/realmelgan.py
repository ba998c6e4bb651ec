"""The weight-normed MelGAN of the reference's featuresynth/experiment/realmelgan.py (SURVEY.md
section 8(f), row 1) on the gfx950 kernels: same class names, constructor signatures, forward
return structures and state_dict keys (`...bias`, `...weight_g`, `...weight_v` in that order, the
`model.N` / `model.disc_i.model.layer_j.k` indices of the reference's Sequentials / ModuleDicts).

  Generator            realmelgan.py:48-89    ReflectionPad1d(3) -> WNConv k7 -> 4 x [LeakyReLU ->
                                              WNConvTranspose1d(k=2r, s=r) -> 3 ResnetBlocks] ->
                                              LeakyReLU -> ReflectionPad1d(3) -> WNConv k7 -> Tanh
  ResnetBlock          realmelgan.py:32-45    shortcut_1x1(x) + conv_1x1(lrelu(conv_k3_dil(reflpad(lrelu(x)))))
  NLayerDiscriminator  realmelgan.py:92-155   ReflectionPad1d(7) + k15, 4 grouped k41/s4 convs, k5, k3
  Discriminator        realmelgan.py:158-181  3 independent NLayerDiscriminators, AvgPool1d(4,2,1,
                                              count_include_pad=False) between scales
  losses               realmelgan.py:185-218

Pads and activations are fused into the conv kernels (activation in front of a conv = `in_act`,
applied to the operand on load); their slots in the Sequentials hold `Fused` placeholders so the
indices do not move.  The Experiment subclass (data loading, zounds) is out of scope.
"""
import os

import numpy as np
from torch import nn

from .._ops import functional as F_
from .._ops import graph as _G
from .._ops import lib as L
from ..loss import hinge_generator_loss, mel_gan_disc_loss  # noqa: F401  (re-exported like the reference)
from ..loss.loss import mel_gan_feature_loss as _unused  # noqa: F401
from ..util.modules import Fused
from .experiment import Experiment as _Experiment

import torch

_ACT = {None: L.ACT_NONE, "lrelu": L.ACT_LRELU, "tanh": L.ACT_TANH}


def weights_init(m):
    """realmelgan.py:15-21 (note: on a weight-normed module `.weight` is derived from g and v, so in
    the reference this only touches a temporary; here the module has no `.weight` at all)."""
    classname = m.__class__.__name__
    if classname.find("Conv") != -1 and hasattr(m, "weight"):
        m.weight.data.normal_(0.0, 0.02)
    elif classname.find("BatchNorm2d") != -1:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)


class _WNConvBase(nn.Module):
    def _init_params(self, shape, fan_in):
        # same initial values as weight_norm(nn.Conv1d(...)): v = default conv init, g = ||v||
        ref = torch.empty(shape)
        nn.init.kaiming_uniform_(ref, a=np.sqrt(5))
        bound = 1 / np.sqrt(fan_in) if fan_in > 0 else 0
        self.bias = nn.Parameter(torch.empty(self._nbias).uniform_(-bound, bound))
        self.weight_g = nn.Parameter(ref.reshape(shape[0], -1).norm(dim=1).reshape(shape[0], 1, 1))
        self.weight_v = nn.Parameter(ref)

    def effective_weight(self):
        w = getattr(self, "_w", None)
        if w is not None:                   # derived for the whole network by _derive_weights()
            self._w = None
            return w
        return F_.WeightNormFn.apply(self.weight_v, self.weight_g)


def _derive_weights(root):
    """w = g * v / ||v|| for every weight-normed layer under `root` in one launch (the reference's
    weight_norm pre-hooks do it layer by layer, realmelgan.py:24-29); each layer's forward consumes
    its entry once."""
    mods = [m for m in root.modules() if isinstance(m, _WNConvBase)]
    if not mods or not mods[0].weight_v.is_cuda:
        return
    vg = []
    for m in mods:
        vg += [m.weight_v, m.weight_g]
    for m, w in zip(mods, F_.WeightNormMultiFn.apply(*vg)):
        m._w = w


class WNConv1d(_WNConvBase):
    """weight_norm(nn.Conv1d(...)) with fused reflection pad / activations."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, activation=None, in_activation=None, reflect_pad=False):
        super().__init__()
        self.cfg = (stride, padding, dilation, groups)
        self.activation, self.in_activation, self.reflect_pad = activation, in_activation, reflect_pad
        self._nbias = out_channels
        self._init_params((out_channels, in_channels // groups, kernel_size),
                          (in_channels // groups) * kernel_size)

    def forward(self, x, residual=None):
        stride, pad, dil, groups = self.cfg
        return F_.Conv1dExFn.apply(x, self.effective_weight(), self.bias, residual, stride, pad, dil,
                                   groups, L.PAD_REFLECT if self.reflect_pad else L.PAD_ZERO,
                                   _ACT[self.activation], _ACT[self.in_activation])


class WNConvTranspose1d(_WNConvBase):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 output_padding=0, activation=None, in_activation=None):
        super().__init__()
        if output_padding != 0:
            raise NotImplementedError("output_padding != 0 (odd upsampling ratios) is not on this path")
        self.cfg = (stride, padding)
        self.activation, self.in_activation = activation, in_activation
        self._nbias = out_channels
        self._init_params((in_channels, out_channels, kernel_size), out_channels * kernel_size)

    def forward(self, x):
        stride, pad = self.cfg
        return F_.ConvTranspose1dExFn.apply(x, self.effective_weight(), self.bias, stride, pad,
                                            _ACT[self.activation], _ACT[self.in_activation])


class ResnetBlock(nn.Module):
    def __init__(self, dim, dilation=1):
        super().__init__()
        self.block = nn.Sequential(
            Fused("LeakyReLU(0.2) applied on load by block.2"),
            Fused("ReflectionPad1d(dilation) fused into block.2"),
            WNConv1d(dim, dim, kernel_size=3, dilation=dilation, padding=dilation, reflect_pad=True,
                     in_activation="lrelu", activation="lrelu"),
            Fused("LeakyReLU(0.2) fused behind block.2"),
            WNConv1d(dim, dim, kernel_size=1),
        )
        self.shortcut = WNConv1d(dim, dim, kernel_size=1)

    def forward(self, x):
        s = self.shortcut(x)
        return self.block[4](self.block[2](x), residual=s)


class Generator(nn.Module):
    _ms_native = True   # featuresynth.train: skip-discarded-work + hipGraph path applies

    def __init__(self, input_size, ngf, n_residual_layers):
        super().__init__()
        ratios = [8, 8, 2, 2]
        self.hop_length = np.prod(ratios)
        mult = int(2 ** len(ratios))
        model = [Fused("ReflectionPad1d(3) fused into model.1"),
                 WNConv1d(input_size, mult * ngf, kernel_size=7, padding=3, reflect_pad=True)]
        for r in ratios:
            model += [Fused("LeakyReLU(0.2) applied on load by the transposed conv"),
                      WNConvTranspose1d(mult * ngf, mult * ngf // 2, kernel_size=r * 2, stride=r,
                                        padding=r // 2 + r % 2, output_padding=r % 2,
                                        in_activation="lrelu")]
            for j in range(n_residual_layers):
                model += [ResnetBlock(mult * ngf // 2, dilation=3 ** j)]
            mult //= 2
        model += [Fused("LeakyReLU(0.2) applied on load by the last conv"),
                  Fused("ReflectionPad1d(3) fused into the last conv"),
                  WNConv1d(ngf, 1, kernel_size=7, padding=3, reflect_pad=True, in_activation="lrelu",
                           activation="tanh"),
                  Fused("Tanh fused behind the last conv")]
        self.model = nn.Sequential(*model)

    def forward(self, x):
        _derive_weights(self)
        return self.model(x)


class NLayerDiscriminator(nn.Module):
    def __init__(self, ndf, n_layers, downsampling_factor, conditioning_channels=0):
        super().__init__()
        self.conditioning_channels = conditioning_channels
        model = nn.ModuleDict()
        model["layer_0"] = nn.Sequential(
            Fused("ReflectionPad1d(7) fused into layer_0.1"),
            WNConv1d(1, ndf, kernel_size=15, padding=7, reflect_pad=True, activation="lrelu"),
            Fused("LeakyReLU(0.2) fused behind layer_0.1"))
        nf = ndf
        stride = downsampling_factor
        for n in range(1, n_layers + 1):
            nf_prev = nf
            nf = min(nf * stride, 1024)
            model["layer_%d" % n] = nn.Sequential(
                WNConv1d(nf_prev, nf, kernel_size=stride * 10 + 1, stride=stride, padding=stride * 5,
                         groups=nf_prev // 4, activation="lrelu"),
                Fused("LeakyReLU(0.2) fused behind the conv"))
        nf = min(nf * 2, 1024)
        model["layer_%d" % (n_layers + 1)] = nn.Sequential(
            WNConv1d(nf_prev, nf, kernel_size=5, stride=1, padding=2, activation="lrelu"),
            Fused("LeakyReLU(0.2) fused behind the conv"))
        if self.conditioning_channels > 0:      # realmelgan.py:128-136
            model["layer_cond"] = nn.Sequential(
                WNConv1d(nf + conditioning_channels, nf, 3, 1, 1, activation="lrelu"),
                Fused("LeakyReLU(0.2) fused behind the conv"),
                WNConv1d(nf, nf, 3, 1, 1, activation="lrelu"),
                Fused("LeakyReLU(0.2) fused behind the conv"),
                WNConv1d(nf, nf, 3, 1, 1, activation="lrelu"),
                Fused("LeakyReLU(0.2) fused behind the conv"))
        model["layer_%d" % (n_layers + 2)] = WNConv1d(nf, 1, kernel_size=3, stride=1, padding=1)
        self.model = model

    def forward(self, x, feat=None):
        results = []
        for key, layer in self.model.items():
            if 'cond' in key:                   # realmelgan.py:149-152: mel features pooled to this rate, stacked in front
                feat = F_.AvgPoolKFn.apply(feat.contiguous(), feat.shape[-1] // x.shape[-1])
                x = torch.cat([feat, x], dim=1)
            x = layer(x)
            results.append(x)
        return results


class Discriminator(nn.Module):
    _ms_native = True

    def __init__(self, num_D, ndf, n_layers, downsampling_factor, conditioning_channels=0):
        super().__init__()
        self.conditioning_channels = conditioning_channels
        self.model = nn.ModuleDict()
        for i in range(num_D):
            self.model["disc_%d" % i] = NLayerDiscriminator(ndf, n_layers, downsampling_factor,
                                                            conditioning_channels)

    def downsample(self, x):
        return F_.AvgPool421Fn.apply(x)

    def forward(self, x, feat=None):
        """realmelgan.py:163-181: three independent discriminators on x, pool(x), pool(pool(x)), issued back to back on the
        caller's stream.  (r01-r03 carried an opt-in variant with the pooled discriminators on forked streams, +3 % under
        hipGraph replay; its replay died once inside hipGraphLaunch and no cause could be proven from that one stack --
        DESIGN_HISTORY.md section 4, "Streams inside the graph" lists what was audited and fixed -- so r04 removed it: a switch that
        can take the process down is not shipped.)"""
        _derive_weights(self)
        discs = list(self.model.values())
        xs = [x]
        for _ in discs[1:]:
            xs.append(self.downsample(xs[-1]))
        zs = [disc(xi, feat) for disc, xi in zip(discs, xs)]
        features = [z[:-1] for z in zs]
        judgements = [z[-1] for z in zs]
        return features, judgements


def real_mel_gan_feature_loss(real_features, fake_features):
    """realmelgan.py:185-203: every l1 term weighted by (1/3) * (4/5)."""
    wt = (1 / 3) * (4.0 / (4 + 1))
    loss = None
    for r_group, f_group in zip(real_features, fake_features):
        for r_f, f_f in zip(r_group, f_group):
            term = wt * F_.L1MeanFn.apply(r_f, f_f)
            loss = term if loss is None else loss + term
    return loss


def mel_gan_gen_loss(real_features, fake_features, real_judgements, fake_judgements,
                     gan_loss=hinge_generator_loss, feature_loss_weight=10):
    """realmelgan.py:206-218."""
    real_features = [list(g) for g in real_features]
    fake_features = [list(g) for g in fake_features]
    fake_judgements = list(fake_judgements)
    S = len(fake_features)
    if gan_loss is hinge_generator_loss and S and len(fake_judgements) == S and \
            all(len(g) == len(fake_features[0]) for g in real_features + fake_features) and \
            fake_judgements[0].is_cuda:
        # fused form (all l1 terms in one launch pair): per-term scale weight * (1/3)(4/5)
        Lyr = len(fake_features[0])
        wt = (1 / 3) * (4.0 / (4 + 1))
        flat_r = [t for g in real_features for t in g]
        flat_f = [t for g in fake_features for t in g]
        return F_.MelGanGenLossFn.apply(S, Lyr, float(feature_loss_weight) * wt * S * Lyr, *flat_r, *flat_f,
                                        *fake_judgements)
    j_loss = None
    for _, f in zip(real_judgements, fake_judgements):
        term = gan_loss(f)
        j_loss = term if j_loss is None else j_loss + term
    return j_loss + feature_loss_weight * real_mel_gan_feature_loss(real_features, fake_features)


class RealMelGanExperiment(_Experiment):
    """Hyper-parameters of realmelgan.py:223-253 (128 mels, ngf 32, 3 residual layers, 3 x
    NLayerDiscriminator(16, 4, 4), Adam 1e-4) minus the file-based feature functions."""

    def __init__(self, optimizer="flat"):
        n_mels, size, total_samples = 128, 32, 8192
        super().__init__(
            Generator(n_mels, size, n_residual_layers=3),
            Discriminator(num_D=3, ndf=16, n_layers=4, downsampling_factor=4),
            learning_rate=1e-4,
            feature_size=size,
            generator_loss=mel_gan_gen_loss,
            discriminator_loss=mel_gan_disc_loss,
            g_init=weights_init,
            d_init=weights_init,
            total_samples=total_samples,
            feature_channels=n_mels,
            samplerate=22050,
            inference_sequence_factor=4,
            optimizer=optimizer)
