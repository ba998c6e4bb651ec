"""Stage-1 spectrogram generator, drop-in for the reference's featuresynth/featuregenerator/upscale.py:77-112
(same class name, constructor signature and state_dict keys `initial.{weight,bias}`, `stack.{0..6}.{weight,bias}`):
noise (B, noise_dim[, 1]) -> Linear(noise_dim, 4*4*1024) + LeakyReLU(0.2) -> (B, 1024, 4, 4) -> five
ConvTranspose2d k(4,4) s(2,2) p(1,1) (1024->512->256->128->128->64) and two k(3,4) s(1,2) p(1,1) (64->32->1),
LeakyReLU(0.2) after all but the last -> (B, 1, 128, 512) -> (B, out_channels, 512).

All arithmetic runs in the gfx950 kernels: the Linear as a 1-tap conv, every ConvTranspose2d as one
ConvTranspose1d launch per output-row phase over lines (util/modules.py:HipConvTranspose2d)."""
from torch import nn

from ..util.modules import HipConvTranspose2d, HipLinear


class SpectrogramFeatureGenerator(nn.Module):
    _ms_native = True

    def __init__(self, out_channels, noise_dim):
        super().__init__()
        self.noise_dim = noise_dim
        self.out_channels = out_channels
        self.initial = HipLinear(noise_dim, 4 * 4 * 1024, activation="lrelu")

        def up(cin, cout, k, s, last=False):
            return HipConvTranspose2d(cin, cout, k, s, (1, 1), activation=None if last else "lrelu")

        self.stack = nn.Sequential(
            up(1024, 512, (4, 4), (2, 2)),      # (8, 8)
            up(512, 256, (4, 4), (2, 2)),       # (16, 16)
            up(256, 128, (4, 4), (2, 2)),       # (32, 32)
            up(128, 128, (4, 4), (2, 2)),       # (64, 64)
            up(128, 64, (4, 4), (2, 2)),        # (128, 128)
            up(64, 32, (3, 4), (1, 2)),         # (128, 256)
            up(32, 1, (3, 4), (1, 2), last=True),   # (128, 512)
        )

    def forward(self, x):
        x = x.reshape(-1, self.noise_dim)
        x = self.initial(x)                                  # LeakyReLU fused
        # (B, 1024*4*4) viewed (B, 1024, 4, 4) -> lines (B, H=4, C=1024, W=4)
        x = x.reshape(x.shape[0], 1024, 4, 4).permute(0, 2, 1, 3).contiguous()
        for layer in self.stack:
            x = layer.forward_lines(x)
        # lines (B, 128, 1, 512) == (B, 1, 128, 512) in memory (one channel)
        return x.reshape(x.shape[0], self.out_channels, -1)
