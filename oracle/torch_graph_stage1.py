"""CPU ORACLE for the stage-1 2-D conv mel GAN (test infrastructure, NOT product code): the reference's
SpectrogramFeatureGenerator / SpectrogramFeatureDiscriminator graph restated with stock
torch.nn.functional ops on the CPU -- which is what the reference executes, minus its module classes.
Pinned against the imported reference classes by tests/test_oracle_golden.py (tests/golden/stage1.npz).

Citations (under /root/reference/featuresynth/): featuregenerator/upscale.py:77-112 (Linear 128 -> 16384,
seven ConvTranspose2d, LeakyReLU(0.2) after all but the last), featurediscriminator/upscale.py:7-27 with
util/modules.py:79-141 (DilatedStack: bias-free k3 convs, dilations 1,3,9,27,81,1,1, activation over the skip
sum where the channel count is kept), loss/loss.py:5-14 (least squares), experiment/featureexperiment.py:274-316
(which loss terms the two trainers use), train/train.py:26-42,63-74.
"""
import torch
import torch.nn.functional as F

from .torch_graph import _lrelu, to_params  # noqa: F401  (mask-aware LeakyReLU, parameter loading)

G_STACK = (((4, 4), (2, 2)), ((4, 4), (2, 2)), ((4, 4), (2, 2)), ((4, 4), (2, 2)), ((4, 4), (2, 2)),
           ((3, 4), (1, 2)), ((3, 4), (1, 2)))
D_DILATIONS = (1, 3, 9, 27, 81, 1, 1)


def generator_param_shapes(out_channels=128, noise_dim=128):
    chans = (1024, 512, 256, 128, 128, 64, 32, 1)
    shapes = [("initial.weight", (4 * 4 * 1024, noise_dim)), ("initial.bias", (4 * 4 * 1024,))]
    for i, (k, _) in enumerate(G_STACK):
        shapes += [("stack.%d.weight" % i, (chans[i], chans[i + 1]) + k), ("stack.%d.bias" % i, (chans[i + 1],))]
    return shapes


def discriminator_param_shapes(feature_channels=128, channels=256):
    shapes = [("stack.main.%d.weight" % i, (channels, feature_channels if i == 0 else channels, 3))
              for i in range(len(D_DILATIONS))]
    return shapes + [("judge.weight", (1, channels, 1)), ("judge.bias", (1,))]


def generator(p, z, out_channels=128, masks=None):
    """masks (optional): {"lin", "ct<i>"} -> bool tensors, in the reference's (B, C, H, W) layout."""
    x = z.reshape(-1, p["initial.weight"].shape[1])
    x = _lrelu(F.linear(x, p["initial.weight"], p["initial.bias"]), masks, "lin")
    x = x.view(x.shape[0], -1, 4, 4)
    for i, (_, s) in enumerate(G_STACK):
        x = F.conv_transpose2d(x, p["stack.%d.weight" % i], p["stack.%d.bias" % i], s, (1, 1))
        if i != len(G_STACK) - 1:
            x = _lrelu(x, masks, "ct%d" % i)
    return x.view(x.shape[0], out_channels, -1)


def discriminator(p, x, masks=None):
    """-> (features[7], judgement).  masks (optional): {"d<i>"} -> bool tensors."""
    feats = []
    for i, d in enumerate(D_DILATIONS):
        w = p["stack.main.%d.weight" % i]
        z = F.conv1d(x, w, None, 1, d, d)
        x = _lrelu(z + x if w.shape[0] == x.shape[1] else z, masks, "d%d" % i)
        feats.append(x)
    return feats, F.conv1d(x, p["judge.weight"], p["judge.bias"])


def ls_gen_loss(fj):
    return 0.5 * ((fj - 1) ** 2).mean()


def ls_disc_loss(rj, fj):
    return 0.5 * (((rj - 1) ** 2).mean() + (fj ** 2).mean())
