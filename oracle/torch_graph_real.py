"""CPU ORACLE (test infrastructure, NOT product code) for SURVEY.md 8(f) row 1: the weight-normed
MelGAN of /root/reference/featuresynth/experiment/realmelgan.py restated with stock
torch.nn.functional CPU ops.  Pinned against the imported reference by tests/test_oracle_golden.py
(tests/golden/realmelgan.npz).  Citations: realmelgan.py:24-45 (weight norm, ResnetBlock),
:48-89 (Generator), :92-181 (discriminators), :185-218 (losses)."""
import numpy as np
import torch
import torch.nn.functional as F

from .torch_graph import _lrelu

RATIOS = (8, 8, 2, 2)

# `masks` (optional, every function below): {site: bool tensor, True = positive branch} read back from the DEVICE's
# activations -- the LeakyReLU backward then takes the device's branch at every element (torch_graph._LReluMasked), so a
# float64 run of this restatement differentiates the same piecewise-linear function as the kernels and a gradient
# comparison measures the kernels, not sign flips at rounding-level pre-activations.  Sites: "pre.<module>" = the
# LeakyReLU in front of that module (a transposed conv, a ResnetBlock, the last conv), "mid.<block>" = the one between a
# ResnetBlock's two convs, "<prefix>layer_<n>" = the one behind a discriminator layer.


def to_params(sd, requires_grad=True, dtype=torch.float32):
    return {k: torch.from_numpy(np.array(v, dtype=np.float32)).to(dtype).requires_grad_(requires_grad)
            for k, v in sd.items()}


def wn(p, name):
    """weight = g * v / ||v|| with the norm over every dim but 0 (torch.nn.utils.weight_norm)."""
    v, g = p[name + ".weight_v"], p[name + ".weight_g"]
    nrm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
    return g * v / nrm


def resblock(p, name, x, dil, masks=None):
    h = F.pad(_lrelu(x, masks, "pre." + name), (dil, dil), mode="reflect")
    h = F.conv1d(h, wn(p, name + ".block.2"), p[name + ".block.2.bias"], dilation=dil)
    h = F.conv1d(_lrelu(h, masks, "mid." + name), wn(p, name + ".block.4"), p[name + ".block.4.bias"])
    return F.conv1d(x, wn(p, name + ".shortcut"), p[name + ".shortcut.bias"]) + h


def generator(p, x, n_residual_layers=3, masks=None):
    h = F.conv1d(F.pad(x, (3, 3), mode="reflect"), wn(p, "model.1"), p["model.1.bias"])
    idx = 2
    for r in RATIOS:
        name = "model.%d" % (idx + 1)
        h = F.conv_transpose1d(_lrelu(h, masks, "pre." + name), wn(p, name), p[name + ".bias"], stride=r,
                               padding=r // 2 + r % 2, output_padding=r % 2)
        idx += 2
        for j in range(n_residual_layers):
            h = resblock(p, "model.%d" % idx, h, 3 ** j, masks)
            idx += 1
    name = "model.%d" % (idx + 2)
    h = F.conv1d(F.pad(_lrelu(h, masks, "pre." + name), (3, 3), mode="reflect"), wn(p, name), p[name + ".bias"])
    return torch.tanh(h)


def nlayer_discriminator(p, prefix, x, masks=None):
    res = []
    h = _lrelu(F.conv1d(F.pad(x, (7, 7), mode="reflect"), wn(p, prefix + "layer_0.1"),
                        p[prefix + "layer_0.1.bias"]), masks, prefix + "layer_0")
    res.append(h)
    for n in range(1, 5):
        w = wn(p, prefix + "layer_%d.0" % n)
        h = _lrelu(F.conv1d(h, w, p[prefix + "layer_%d.0.bias" % n], stride=4, padding=20,
                            groups=h.shape[1] // 4), masks, prefix + "layer_%d" % n)
        res.append(h)
    h = _lrelu(F.conv1d(h, wn(p, prefix + "layer_5.0"), p[prefix + "layer_5.0.bias"], padding=2), masks, prefix + "layer_5")
    res.append(h)
    res.append(F.conv1d(h, wn(p, prefix + "layer_6"), p[prefix + "layer_6.bias"], padding=1))
    return res


def discriminator(p, x, num_D=3, masks=None):
    feats, judges = [], []
    for i in range(num_D):
        z = nlayer_discriminator(p, "model.disc_%d.model." % i, x, masks)
        feats.append(z[:-1])
        judges.append(z[-1])
        x = F.avg_pool1d(x, 4, stride=2, padding=1, count_include_pad=False)
    return feats, judges


def disc_loss(rj, fj):
    return sum((F.relu(1 - r) + F.relu(1 + f)).mean() for r, f in zip(rj, fj))


def gen_loss(rf, ff, fj, weight=10.0):
    wt = (1 / 3) * (4.0 / 5)
    j = sum((-f).mean() for f in fj)
    fl = 0
    for rg, fg in zip(rf, ff):
        for r, f in zip(rg, fg):
            fl = fl + wt * F.l1_loss(r, f)
    return j + weight * fl
