"""Stage-1 experiment glue with the surface of the reference's featuresynth/experiment/featureexperiment.py
(BaseFeatureExperiment :106-253, TwoDimGeneratorFeatureExperiment :271-316; SURVEY.md 8(f) row 2 / BASELINE
config 5): a SpectrogramFeatureGenerator trained against a SpectrogramFeatureDiscriminator with the
least-squares GAN losses, Adam(1e-4, (0.5, 0.9)) each, `training_steps = cycle([D, G])`, the reference's
trainers (featuresynth.train), `preprocess_batch` drawing the noise conditioning, checkpoint / resume, and a
vocoder that turns generated spectrograms into audio.

Differences, all additive: FlatAdam by default (optimizer="torch" for stock Adam); the vocoder is any stage-2
generator module (the reference loads FilterBankMultiscaleExperiment's, which is outside this build) and may
be None; the sound-file data layer is out of scope (`synthetic_batch_stream`)."""
import os
from itertools import cycle

import numpy as np
import torch

from .. import _dist
from ..featurediscriminator import SpectrogramFeatureDiscriminator
from ..featuregenerator import SpectrogramFeatureGenerator
from ..loss import least_squares_disc_loss, least_squares_generator_loss
from ..optim import FlatAdam
from ..train import DiscriminatorTrainer, GeneratorTrainer
from .init import weights_init


class NeuralVocoder(object):
    """Reference featureexperiment.py:93-99."""

    def __init__(self, network):
        self.network = network

    def __call__(self, features):
        with torch.no_grad():
            return self.network(features)


def _gen_loss(r_features, f_features, r_score, f_score, gan_loss):      # reference :289-290
    return least_squares_generator_loss(f_score)


def _disc_loss(r_score, f_score, gan_loss):                             # reference :292-293
    return least_squares_disc_loss(r_score, f_score)


class BaseFeatureExperiment(object):
    def __init__(self, vocoder, feature_generator, generator_init, generator_loss, feature_disc, disc_init,
                 disc_loss, feature_funcs, feature_spec, audio_repr_class, learning_rate, condition_shape,
                 samplerate, anchor_feature='spectrogram', optimizer="flat"):
        super().__init__()
        self.anchor_feature = anchor_feature
        self.feature_spec = feature_spec
        self.samplerate = samplerate
        self.condition_shape = condition_shape
        self.disc_loss = disc_loss
        self.generator_loss = generator_loss
        self.disc_init = disc_init
        self.generator_init = generator_init
        self.learning_rate = learning_rate
        self.feature_disc = feature_disc
        self.audio_repr_class = audio_repr_class
        self.feature_funcs = feature_funcs
        self.feature_generator = feature_generator
        self.vocoder = vocoder
        self.device = None
        self.feature_generator.apply(self.generator_init)
        self.feature_disc.apply(self.disc_init)
        make = FlatAdam if optimizer == "flat" else torch.optim.Adam
        self.g_optim = make(self.feature_generator.parameters(), lr=learning_rate, betas=(0.5, 0.9))
        self.d_optim = make(self.feature_disc.parameters(), lr=learning_rate, betas=(0.5, 0.9))
        self.g_trainer = GeneratorTrainer(self.feature_generator, self.g_optim, self.feature_disc, self.d_optim,
                                          self.generator_loss, sub_loss=None)
        self.d_trainer = DiscriminatorTrainer(self.feature_generator, self.g_optim, self.feature_disc,
                                              self.d_optim, self.disc_loss, sub_loss=None)
        self.training_steps = cycle([self.d_trainer.train, self.g_trainer.train])

    def to(self, device):
        self.feature_generator.to(device)
        self.feature_disc.to(device)
        self.device = device
        if _dist.is_distributed():          # replicas start from rank 0's parameters
            for net in (self.feature_generator, self.feature_disc):
                for p in net.parameters():
                    _dist.broadcast_(p.data, 0)
        return self

    def _name(self):
        return self.__class__.__name__.lower().replace('experiment', '')

    def _gen_name(self):
        return 'trained_models/%s_gen.dat' % self._name()

    def _disc_name(self):
        return 'trained_models/%s_disc.dat' % self._name()

    def checkpoint(self):
        if _dist.rank() == 0:
            os.makedirs('trained_models', exist_ok=True)
            torch.save(self.feature_generator.state_dict(), self._gen_name())
            torch.save(self.feature_disc.state_dict(), self._disc_name())
        _dist.barrier()

    def resume(self):
        self.feature_generator.load_state_dict(torch.load(self._gen_name()))
        self.feature_disc.load_state_dict(torch.load(self._disc_name()))

    def batch_stream(self, path, pattern, batch_size):
        raise NotImplementedError(
            "the sound-file / LMDB data layer (featuresynth.data) is outside this build's hot path; feed "
            "training_loop from synthetic_batch_stream() or any iterator of 1-tuples of (B, C, T) spectrograms")

    def preprocess_batch(self, batch):
        """A batch of real spectrograms plus a freshly drawn conditioning (noise) vector (reference :222-232)."""
        spec, = batch
        conditioning = np.random.normal(0, 1, (len(spec),) + tuple(self.condition_shape))
        return np.asarray(spec, dtype=np.float32), conditioning.astype(np.float32)

    def synthetic_batch_stream(self, batch_size, n_batches=None):
        size, channels = self.feature_spec[self.anchor_feature]
        i, base = 0, 1000 * _dist.rank()
        while n_batches is None or i < n_batches:
            rng = np.random.default_rng(300 + base + i)
            yield ((rng.standard_normal((batch_size, channels, size)) * 0.5).astype(np.float32),)
            i += 1

    @property
    def is_autoregressive(self):
        return hasattr(self.feature_generator, "generate")

    def features_to_audio(self, features):
        """Generated spectrograms -> waveform through the vocoder (numpy in, numpy out)."""
        if self.vocoder is None:
            raise RuntimeError("this experiment was built without a vocoder")
        t = torch.from_numpy(np.ascontiguousarray(features, dtype=np.float32))
        if self.device is not None:
            t = t.to(self.device)
        out = self.vocoder(t)
        return out.detach().cpu().numpy() if isinstance(out, torch.Tensor) else out


class TwoDimGeneratorFeatureExperiment(BaseFeatureExperiment):
    """Reference featureexperiment.py:271-316: 128-dimensional noise -> (128 mel, 512 frames) spectrograms."""
    N_MELS = 128
    SAMPLERATE = 22050

    def __init__(self, vocoder_network=None, optimizer="flat"):
        noise_dim = 128
        super().__init__(
            vocoder=NeuralVocoder(vocoder_network) if vocoder_network is not None else None,
            feature_generator=SpectrogramFeatureGenerator(out_channels=self.N_MELS, noise_dim=noise_dim),
            generator_init=weights_init,
            generator_loss=_gen_loss,
            feature_disc=SpectrogramFeatureDiscriminator(feature_channels=self.N_MELS, channels=256),
            disc_init=weights_init,
            disc_loss=_disc_loss,
            feature_funcs=None,
            feature_spec={'spectrogram': (512, self.N_MELS)},
            audio_repr_class=None,
            learning_rate=1e-4,
            condition_shape=(noise_dim, 1),
            samplerate=self.SAMPLERATE,
            optimizer=optimizer)
