from .feature import Audio2Mel
