"""CPU ORACLE, second form (test infrastructure, NOT product code): the reference's stage-2
graph restated with stock torch.nn.functional ops on the CPU -- which is literally what the
reference executes (nn.Conv1d / nn.ConvTranspose1d / F.leaky_relu / F.avg_pool1d, Adam), minus
its module classes.  Used for (a) parity checks at full BASELINE sizes, where the plain-C oracle
is too slow, and (b) bench.py's cpu_baseline leg ("port": the reference path on the GPU box's
host cores).  Pinned against the imported reference by tests/test_oracle_golden.py.

Citations (under /root/reference/featuresynth/): generator/full.py:22-50,
util/modules.py:384-405, discriminator/full.py:13-40, discriminator/melgan.py:13-27,
loss/loss.py:9-78, train/train.py:26-42,63-74, experiment/experiment.py:111-117.
"""
import numpy as np
import torch
import torch.nn.functional as F

_G_UPS = (("main.3", "main.5", 8, 4), ("main.6", "main.8", 8, 4), ("main.9", "main.11", 2, 1),
          ("main.12", "main.14", 2, 1))
_D_MAIN = ((1, 7, 1), (4, 20, 4), (4, 20, 16), (4, 20, 64), (4, 20, 256), (1, 2, 1))


def to_params(sd, requires_grad=True, dtype=torch.float32):
    return {k: torch.from_numpy(np.array(v, dtype=np.float32)).to(dtype).requires_grad_(requires_grad)
            for k, v in sd.items()}


class _LReluMasked(torch.autograd.Function):
    """LeakyReLU(0.2) whose BACKWARD takes the branch from a given mask instead of the sign of its own
    input.  A pre-activation within rounding of 0 may carry the other sign on the device than here
    (different summation order, or this restatement running in float64): one such element changes the
    gradients behind it by O(1) although both forwards agree to rounding.  With the mask read back from
    the device's saved activation the two backward passes differentiate the SAME piecewise-linear
    function, and the comparison measures the kernels, not the flips."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return F.leaky_relu(x, 0.2)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return torch.where(mask, g, 0.2 * g), None


def _lrelu(x, masks, key):
    if masks is None:
        return F.leaky_relu(x, 0.2)
    m = masks[key]
    if tuple(m.shape) != tuple(x.shape):
        raise ValueError("mask %s: shape %s, activation %s" % (key, tuple(m.shape), tuple(x.shape)))
    return _LReluMasked.apply(x, m)


def generator(p, x, masks=None):
    """masks (optional): {"conv0", "ct<k>", "a<k>.<atom>.0|1"} -> bool tensors (True = positive branch)."""
    h = _lrelu(F.conv1d(F.pad(x, (3, 3), mode="reflect"), p["main.1.weight"], p["main.1.bias"]), masks, "conv0")
    for k, (ct, st, s, pad) in enumerate(_G_UPS):
        h = _lrelu(F.conv_transpose1d(h, p[ct + ".weight"], p[ct + ".bias"], s, pad), masks, "ct%d" % k)
        for a, d in enumerate((1, 3, 9)):
            n0, n1 = "%s.main.%d.main.0" % (st, a), "%s.main.%d.main.1" % (st, a)
            t = _lrelu(F.conv1d(h, p[n0 + ".weight"], p[n0 + ".bias"], 1, d, d), masks, "a%d.%d.0" % (k, a))
            h = h + _lrelu(F.conv1d(t, p[n1 + ".weight"], p[n1 + ".bias"], 1, 1), masks, "a%d.%d.1" % (k, a))
    return torch.tanh(F.conv1d(h, p["main.15.weight"], p["main.15.bias"], 1, 3))


def full_discriminator(p, x, prefix="disc.", masks=None, scale=0):
    feats = []
    for i, (s, pad, g) in enumerate(_D_MAIN):
        x = _lrelu(F.conv1d(x, p["%smain.%d.weight" % (prefix, i)], p["%smain.%d.bias" % (prefix, i)],
                            s, pad, 1, g), masks, "d%d.%d" % (scale, i))
        feats.append(x)
    return feats, F.conv1d(x, p[prefix + "judge.weight"], p[prefix + "judge.bias"], 1, 1)


def discriminator(p, x, scales=2, masks=None):
    """masks (optional): {"d<scale>.<layer>"} -> bool tensors."""
    feats, judges = [], []
    for s in range(scales + 1):
        if s:
            x = F.avg_pool1d(x, 4, 2, 2)
        f, j = full_discriminator(p, x, masks=masks, scale=s)
        feats.append(f)
        judges.append(j)
    return feats, judges


def decode_sign_words(words):
    """Sign words of the fused atom kernels (include/msynth.h, ms_residual_atom_fwd_signs: int16 (B, C / 32, 2, L), bit 15 - r
    of word (b, blk, h, l) = activation[b, 32 blk + (r & 3) + 8 (r >> 2) + 4 h, l] > 0) -> CPU bool tensor (B, C, L)."""
    w = words.detach().cpu().to(torch.int32) & 0xFFFF
    B, NB, _, Lg = w.shape
    out = torch.zeros((B, NB * 32, Lg), dtype=torch.bool)
    for h in range(2):
        for r in range(16):
            ch = (r & 3) + 8 * (r >> 2) + 4 * h
            out[:, ch::32, :] = ((w[:, :, h, :] >> (15 - r)) & 1).bool()
    return out


def generator_masks_from_tape(tape, to_bool):
    """Device-side generator tape (featuresynth._ops.graph.gen_forward) -> the masks `generator` takes.
    to_bool: tensor -> CPU bool tensor (activation > 0)."""
    masks, k, a = {}, -1, 0
    for rec in tape:
        if rec[0] == "conv0":
            masks["conv0"] = to_bool(rec[3])
        elif rec[0] == "convT":
            k, a = k + 1, 0
            masks["ct%d" % k] = to_bool(rec[3])
        elif rec[0] == "atom":
            t, u = rec[1][3], rec[1][4]
            masks["a%d.%d.0" % (k, a)] = to_bool(t)
            masks["a%d.%d.1" % (k, a)] = decode_sign_words(u) if u.dtype == torch.int16 else to_bool(u)
            a += 1
    return masks


def discriminator_masks_from_ctx(ctx, to_bool, rows=None):
    """Device-side discriminator context (graph.melgan_forward) -> the masks `discriminator` takes;
    rows = slice of the batch (the D-step runs ONE pass over [fake; real])."""
    tapes, _ = ctx
    masks = {}
    for s, tape in enumerate(tapes):
        for li in range(6):
            h = tape[li][2]
            masks["d%d.%d" % (s, li)] = to_bool(h if rows is None else h[rows])
    return masks


def disc_loss(rj, fj):
    return sum((F.relu(1 - r) + F.relu(1 + f)).mean() for r, f in zip(rj, fj))


def gen_loss(rf, ff, fj, weight=10.0):
    j = sum((-f).mean() for f in fj)
    fl = 0
    for rg, fg in zip(rf, ff):
        for r, f in zip(rg, fg):
            fl = fl + (1.0 / len(rf)) * (1.0 / len(rg)) * F.l1_loss(r, f)
    return j + weight * fl


class Trainer:
    """Alternating D / G steps with Adam(1e-4, (0.5, 0.9)), executing exactly what the reference
    executes (nothing detached or frozen: train.py:66-71, :29-36)."""

    def __init__(self, gsd, dsd):
        self.gp, self.dp = to_params(gsd), to_params(dsd)
        self.g_optim = torch.optim.Adam(list(self.gp.values()), lr=1e-4, betas=(0.5, 0.9))
        self.d_optim = torch.optim.Adam(list(self.dp.values()), lr=1e-4, betas=(0.5, 0.9))

    def d_step(self, samples, features):
        self.g_optim.zero_grad(); self.d_optim.zero_grad()
        fake = generator(self.gp, features)
        _, fj = discriminator(self.dp, fake)
        _, rj = discriminator(self.dp, samples)
        loss = disc_loss(rj, fj)
        loss.backward()
        self.d_optim.step()
        return {"d_loss": loss.item()}

    def g_step(self, samples, features):
        self.g_optim.zero_grad(); self.d_optim.zero_grad()
        fake = generator(self.gp, features)
        ff, fj = discriminator(self.dp, fake)
        rf, rj = discriminator(self.dp, samples)
        loss = gen_loss(rf, ff, fj)
        loss.backward()
        self.g_optim.step()
        return {"g_loss": loss.item(), "fake": fake.detach().numpy()}
