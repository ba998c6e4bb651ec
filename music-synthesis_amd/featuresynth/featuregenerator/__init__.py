"""Stage-1 feature generators: the reference's featuresynth/featuregenerator package
exports a dozen experimental generators; the one on the two-stage path (BASELINE config 5,
experiment/featureexperiment.py:274-316) is SpectrogramFeatureGenerator."""
from .upscale import SpectrogramFeatureGenerator  # noqa: F401
