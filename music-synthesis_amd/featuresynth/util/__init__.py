from .device import device
