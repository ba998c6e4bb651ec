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


def to_params(sd, requires_grad=True):
    return {k: torch.from_numpy(np.array(v, dtype=np.float32)).requires_grad_(requires_grad)
            for k, v in sd.items()}


def generator(p, x):
    h = F.leaky_relu(F.conv1d(F.pad(x, (3, 3), mode="reflect"), p["main.1.weight"], p["main.1.bias"]), 0.2)
    for ct, st, s, pad in _G_UPS:
        h = F.leaky_relu(F.conv_transpose1d(h, p[ct + ".weight"], p[ct + ".bias"], s, pad), 0.2)
        for a, d in enumerate((1, 3, 9)):
            n0, n1 = "%s.main.%d.main.0" % (st, a), "%s.main.%d.main.1" % (st, a)
            t = F.leaky_relu(F.conv1d(h, p[n0 + ".weight"], p[n0 + ".bias"], 1, d, d), 0.2)
            h = h + F.leaky_relu(F.conv1d(t, p[n1 + ".weight"], p[n1 + ".bias"], 1, 1), 0.2)
    return torch.tanh(F.conv1d(h, p["main.15.weight"], p["main.15.bias"], 1, 3))


def full_discriminator(p, x, prefix="disc."):
    feats = []
    for i, (s, pad, g) in enumerate(_D_MAIN):
        x = F.leaky_relu(F.conv1d(x, p["%smain.%d.weight" % (prefix, i)], p["%smain.%d.bias" % (prefix, i)],
                                  s, pad, 1, g), 0.2)
        feats.append(x)
    return feats, F.conv1d(x, p[prefix + "judge.weight"], p[prefix + "judge.bias"], 1, 1)


def discriminator(p, x, scales=2):
    feats, judges = [], []
    for s in range(scales + 1):
        if s:
            x = F.avg_pool1d(x, 4, 2, 2)
        f, j = full_discriminator(p, x)
        feats.append(f)
        judges.append(j)
    return feats, judges


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
