"""Grouped strided 1-D conv discriminator, drop-in for the reference's
featuresynth/discriminator/full.py:10-40 (state_dict keys main.{0..5}.{weight,bias},
judge.{weight,bias}); returns (features: list[6], judgement)."""
from torch import nn

from .._ops import functional as F_
from ..util.modules import HipConv1d


class FullDiscriminator(nn.Module):
    _ms_native = True   # train.py: both networks native -> skip-wasted-work + hipGraph path

    def __init__(self):
        super().__init__()
        self.main = nn.Sequential(
            HipConv1d(1, 16, 15, 1, padding=7, activation="lrelu"),
            HipConv1d(16, 64, 41, 4, padding=20, groups=4, activation="lrelu"),
            HipConv1d(64, 256, 41, 4, padding=20, groups=16, activation="lrelu"),
            HipConv1d(256, 1024, 41, 4, padding=20, groups=64, activation="lrelu"),
            HipConv1d(1024, 1024, 41, 4, padding=20, groups=256, activation="lrelu"),
            HipConv1d(1024, 1024, 5, 1, padding=2, activation="lrelu"),
        )
        self.judge = HipConv1d(1024, 1, 3, 1, padding=1)

    def initialize_weights(self):  # reference full.py:24-32 (unused by the hot path)
        from torch.nn.init import calculate_gain, xavier_normal_
        for name, weight in self.named_parameters():
            if weight.data.dim() > 2:
                if "judge" in name:
                    xavier_normal_(weight.data, 1)
                else:
                    xavier_normal_(weight.data, calculate_gain("leaky_relu", 0.2))
        return self

    def forward(self, x):
        out = F_.MelGanDiscFn.apply(x, 0, *self.parameters())
        return list(out[:6]), out[6]
