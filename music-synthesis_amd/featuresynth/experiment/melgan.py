"""MultiScaleMelGanExperiment with the hyper-parameters of the reference's
featuresynth/experiment/melgan.py:11-42 (8192-sample windows at 22.05 kHz, 1024/256 STFT, 128 mel
channels, MelGanGenerator + MelGanDiscriminator, Adam 1e-4) minus its file-based feature functions."""
from ..discriminator import MelGanDiscriminator
from ..generator.full import MelGanGenerator
from ..loss import mel_gan_disc_loss, mel_gan_gen_loss
from .experiment import Experiment
from .init import weights_init


class MultiScaleMelGanExperiment(Experiment):
    def __init__(self, n_mels=128, optimizer="flat"):
        total_samples = 8192
        hop = 256
        feature_size = total_samples // hop
        super().__init__(
            generator=MelGanGenerator(feature_size, n_mels),
            discriminator=MelGanDiscriminator(),
            learning_rate=1e-4,
            feature_size=feature_size,
            generator_loss=mel_gan_gen_loss,
            discriminator_loss=mel_gan_disc_loss,
            g_init=weights_init,
            d_init=weights_init,
            total_samples=total_samples,
            feature_channels=n_mels,
            samplerate=22050,
            optimizer=optimizer)
