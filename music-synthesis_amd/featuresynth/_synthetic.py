"""Deterministic synthetic weights / inputs for parity tests and bench.py.

Pure numpy (no compute path).  Follows SURVEY.md section 8(d): weights
`default_rng(seed).standard_normal(shape) * 0.02` drawn in state_dict order
(same distribution as experiment/init.py:3-9 of the reference), biases 0 unless
`bias_scale` is given (parity fixtures use non-zero biases to exercise the bias
path); samples ~ U(-0.95, 0.95), features ~ N(0, 1).
"""
import numpy as np

WEIGHT_SEED = 7


def synthetic_state_dict(param_shapes, seed=WEIGHT_SEED, weight_scale=0.02, bias_scale=0.0):
    """param_shapes: iterable of (name, shape) in state_dict order -> {name: float32 array}."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in param_shapes:
        shape = tuple(int(s) for s in shape)
        if name.endswith("bias"):
            if bias_scale:
                out[name] = (rng.standard_normal(shape) * bias_scale).astype(np.float32)
            else:
                out[name] = np.zeros(shape, np.float32)
        elif name.endswith("weight_g"):      # weight-norm gain: positive, around 1
            out[name] = (0.5 + rng.random(shape)).astype(np.float32)
        else:
            out[name] = (rng.standard_normal(shape) * weight_scale).astype(np.float32)
    return out


def module_param_shapes(module):
    return [(k, tuple(v.shape)) for k, v in module.state_dict().items()]


def synthetic_samples(batch, length=8192, rank=0):
    return np.random.default_rng(100 + rank).uniform(-0.95, 0.95, (batch, 1, length)).astype(np.float32)


def synthetic_features(batch, mels=80, frames=32, rank=0):
    return np.random.default_rng(200 + rank).standard_normal((batch, mels, frames)).astype(np.float32)


def strided_sample(a, n=256):
    """A deterministic <=n element sample of a tensor (fixture compression)."""
    flat = np.asarray(a).reshape(-1)
    step = max(1, flat.size // n)
    return flat[::step][:n].copy()
