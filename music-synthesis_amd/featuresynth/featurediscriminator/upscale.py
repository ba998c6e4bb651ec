"""Stage-1 spectrogram discriminator, drop-in for the reference's featuresynth/featurediscriminator/upscale.py:7-27
(state_dict keys `stack.main.{0..6}.weight`, `judge.{weight,bias}`): a residual DilatedStack of seven k3 convs
(dilations 1, 3, 9, 27, 81, 1, 1; LeakyReLU(0.2) over the skip sum) and a 1x1 judge conv.
forward(x, conditioning) -> (features: list[7], judgement (B, 1, T)); the conditioning is ignored (the
generator is unconditioned, reference comment :21-22)."""
from torch import nn
from torch.nn import functional as F

from ..util.modules import DilatedStack, HipConv1d


class SpectrogramFeatureDiscriminator(nn.Module):
    _ms_native = True

    def __init__(self, feature_channels, channels):
        super().__init__()
        self.channels = channels
        self.feature_channels = feature_channels
        self.stack = DilatedStack(feature_channels, channels, 3, [1, 3, 9, 27, 81, 1, 1],
                                  activation=lambda x: F.leaky_relu(x, 0.2), residual=True)
        self.judge = HipConv1d(channels, 1, 1, 1, 0)

    def forward(self, x, conditioning=None):
        features, x = self.stack(x, return_features=True)
        x = self.judge(x)
        return features, x
