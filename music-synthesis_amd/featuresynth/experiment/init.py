"""Initialiser with the behaviour of the reference's featuresynth/experiment/init.py:3-9:
every module whose class name contains "Conv" gets N(0, 0.02) weights and zero biases."""
import torch


def weights_init(m):
    if "Conv" not in type(m).__name__:
        return
    with torch.no_grad():
        m.weight.normal_(0.0, 0.02)
        bias = getattr(m, "bias", None)
        if bias is not None:
            bias.zero_()
