from .feature import Audio2Mel, audio_from_samples, resample  # noqa: F401
