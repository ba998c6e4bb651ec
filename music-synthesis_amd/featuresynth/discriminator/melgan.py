"""Three-scale discriminator, drop-in for the reference's
featuresynth/discriminator/melgan.py:7-27: ONE shared FullDiscriminator (state_dict prefix
`disc.`) applied to x, pool(x), pool(pool(x)) with F.avg_pool1d(kernel_size=4, stride=2,
padding=2).  forward accepts (and ignores) a second positional argument so the reference's
trainers, which call discriminator(x, features) (train/train.py:30-31,67-68), work unchanged."""
from torch import nn

from .._ops import functional as F_
from .full import FullDiscriminator


class MelGanDiscriminator(nn.Module):
    _ms_native = True   # train.py: both networks native -> skip-wasted-work + hipGraph path

    def __init__(self):
        super().__init__()
        self.disc = FullDiscriminator()
        self.scales = 2

    def forward(self, x, features=None):
        n = self.scales + 1
        out = F_.MelGanDiscFn.apply(x, self.scales, *self.disc.parameters())
        feats = [list(out[6 * s:6 * s + 6]) for s in range(n)]
        judgements = list(out[6 * n:6 * n + n])
        return feats, judgements
