"""featuresynth -- MI355X-native build of the stage-2 mel -> waveform GAN hot path of
JohnVinyard/music-synthesis, behind the reference's own module surface
(featuresynth.generator / .discriminator / .loss / .train / .feature / .experiment.init).

All arithmetic runs in hand-written gfx950 HIP kernels reached through the C ABI in
include/msynth.h (libmsynth_hip.so); there is no CPU or PyTorch-op fallback.
"""
from . import (discriminator, experiment, feature, featurediscriminator, featuregenerator, generator, loss,  # noqa: F401
               train, util)
from .discriminator import FullDiscriminator, MelGanDiscriminator  # noqa: F401
from .generator import MelGanGenerator  # noqa: F401
from .optim import FlatAdam  # noqa: F401

__version__ = "0.1.0"
