"""Stage-1 feature discriminators: of the reference's featurediscriminator package
the two-stage path (experiment/featureexperiment.py:274-316) uses SpectrogramFeatureDiscriminator."""
from .upscale import SpectrogramFeatureDiscriminator  # noqa: F401
