#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the imported, UNMODIFIED reference on CPU.

Runs only in the build container (needs /root/reference); the fixtures it writes
are data (inputs / expected outputs), never reference source.  See SURVEY.md
section 8(c) for the import technique (tools/ref_import.py).

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py

Fixtures:
  ops_tiny.npz   full tensors for every op shape-class on the hot path (tiny C / L, odd L)
  g_fwd.npz      MelGanGenerator forward, BASELINE config 2 (B=1, 80 mel, 8192 samples)
  d_fwd.npz      MelGanDiscriminator forward: judgements + summaries of the 18 feature maps
  train.npz      losses, grads and parameters across alternating D,G,D,G Adam steps
  stage1.npz     stage-1 spectrogram GAN (featuregenerator/featurediscriminator upscale.py): forwards and
                 one D-step / G-step of the reference's trainers with its least-squares losses
  audio2mel.npz  Audio2Mel on 1 s of noise (STFT part via torch.stft(return_complex=True),
                 mel basis cross-checked against transformers.audio_utils slaney filter bank)
Each expected value is stored as computed by the reference in float32 ("*_ref32",
its real behaviour) and, where cheap, by the same reference modules cast to
float64 ("*_ref64", a tighter pin for the oracle).
"""
import copy
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import importlib.util

import numpy as np
import torch
import torch.nn.functional as F

import ref_import


def _load_synthetic():
    p = os.path.join(ROOT, "music-synthesis_amd", "featuresynth", "_synthetic.py")
    spec = importlib.util.spec_from_file_location("_msynth_synthetic", p)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


SYN = _load_synthetic()
OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


def t2n(t):
    return t.detach().cpu().numpy().copy()


def load_sd(module, sd):
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return module


def summary(a, n=256):
    a = np.asarray(a, dtype=np.float64)
    return np.array([np.linalg.norm(a.reshape(-1)), a.sum(), np.abs(a).max()], np.float64)


class TwoArgD(torch.nn.Module):
    """The reference's trainers call discriminator(x, features) (train/train.py:30-31,67-68)
    while MelGanDiscriminator.forward takes one arg (discriminator/melgan.py:13): adapt."""

    def __init__(self, d):
        super().__init__()
        self.d = d

    def forward(self, x, feat=None):
        return self.d(x)


# ------------------------------------------------------------------ ops_tiny

def ops_tiny(ns):
    rng = np.random.default_rng(1234)
    out = {}

    def rnd(*shape, scale=1.0):
        return (rng.standard_normal(shape) * scale).astype(np.float32)

    def conv_case(name, B, Cin, L, Cout, K, stride, pad, dil, groups, act, reflect=False):
        x = torch.from_numpy(rnd(B, Cin, L)).requires_grad_(True)
        w = torch.from_numpy(rnd(Cout, Cin // groups, K, scale=0.3)).requires_grad_(True)
        b = torch.from_numpy(rnd(Cout, scale=0.3)).requires_grad_(True)
        if reflect:
            xp = torch.nn.ReflectionPad1d(pad)(x)
            pre = F.conv1d(xp, w, b, stride, 0, dil, groups)
        else:
            pre = F.conv1d(x, w, b, stride, pad, dil, groups)
        y = {"none": lambda v: v, "lrelu": lambda v: F.leaky_relu(v, 0.2),
             "tanh": torch.tanh}[act](pre)
        gy = torch.from_numpy(rnd(*y.shape))
        y.backward(gy)
        out.update({name + "/x": t2n(x), name + "/w": t2n(w), name + "/b": t2n(b),
                    name + "/y": t2n(y), name + "/gy": t2n(gy), name + "/gx": t2n(x.grad),
                    name + "/gw": t2n(w.grad), name + "/gb": t2n(b.grad),
                    name + "/cfg": np.array([stride, pad, dil, groups,
                                             {"none": 0, "lrelu": 1, "tanh": 2}[act],
                                             int(reflect)], np.int64)})

    for d in (1, 3, 9):
        conv_case("conv_k3_d%d" % d, 2, 8, 67, 8, 3, 1, d, d, 1, "lrelu")
    conv_case("conv_k7_reflect", 2, 5, 19, 6, 7, 1, 3, 1, 1, "lrelu", reflect=True)
    conv_case("conv_k7_tanh", 2, 4, 50, 1, 7, 1, 3, 1, 1, "tanh")
    conv_case("conv_k15", 2, 1, 67, 16, 15, 1, 7, 1, 1, "lrelu")
    conv_case("conv_k41_g2", 2, 8, 67, 16, 41, 4, 20, 1, 2, "lrelu")
    conv_case("conv_k41_g2_even", 1, 8, 130, 16, 41, 4, 20, 1, 2, "lrelu")
    conv_case("conv_k41_g2_4x4", 2, 8, 33, 8, 41, 4, 20, 1, 2, "lrelu")
    conv_case("conv_k41_short", 2, 8, 9, 8, 41, 4, 20, 1, 2, "lrelu")
    conv_case("conv_k5", 2, 8, 17, 8, 5, 1, 2, 1, 1, "lrelu")
    conv_case("conv_k5_l9", 1, 8, 9, 8, 5, 1, 2, 1, 1, "lrelu")
    conv_case("conv_judge", 2, 8, 9, 1, 3, 1, 1, 1, 1, "none")

    def convt_case(name, B, Cin, L, Cout, K, stride, pad):
        x = torch.from_numpy(rnd(B, Cin, L)).requires_grad_(True)
        w = torch.from_numpy(rnd(Cin, Cout, K, scale=0.3)).requires_grad_(True)
        b = torch.from_numpy(rnd(Cout, scale=0.3)).requires_grad_(True)
        y = F.leaky_relu(F.conv_transpose1d(x, w, b, stride, pad), 0.2)
        gy = torch.from_numpy(rnd(*y.shape))
        y.backward(gy)
        out.update({name + "/x": t2n(x), name + "/w": t2n(w), name + "/b": t2n(b),
                    name + "/y": t2n(y), name + "/gy": t2n(gy), name + "/gx": t2n(x.grad),
                    name + "/gw": t2n(w.grad), name + "/gb": t2n(b.grad),
                    name + "/cfg": np.array([stride, pad], np.int64)})

    convt_case("convt_k16_s8", 2, 6, 9, 4, 16, 8, 4)
    convt_case("convt_k4_s2", 2, 4, 13, 3, 4, 2, 1)
    convt_case("convt_k16_s8_l1", 1, 3, 1, 2, 16, 8, 4)

    # reference ResidualAtom / ResidualStack classes (util/modules.py:350-405)
    torch.manual_seed(5)
    for d in (1, 3, 9):
        atom = ns.ResidualAtom(8, d, False)
        for p in atom.parameters():
            p.data.normal_(0, 0.3)
        x = torch.from_numpy(rnd(2, 8, 67)).requires_grad_(True)
        y = atom(x)
        gy = torch.from_numpy(rnd(*y.shape))
        y.backward(gy)
        nm = "atom_d%d" % d
        out.update({nm + "/x": t2n(x), nm + "/y": t2n(y), nm + "/gy": t2n(gy),
                    nm + "/gx": t2n(x.grad)})
        for k, v in atom.state_dict().items():
            out[nm + "/sd/" + k] = t2n(v)
        for k, p in atom.named_parameters():
            out[nm + "/grad/" + k] = t2n(p.grad)
    stack = ns.ResidualStack(8, [1, 3, 9])
    for p in stack.parameters():
        p.data.normal_(0, 0.2)
    x = torch.from_numpy(rnd(1, 8, 41))
    out["stack/x"], out["stack/y"] = t2n(x), t2n(stack(x))
    for k, v in stack.state_dict().items():
        out["stack/sd/" + k] = t2n(v)

    # avg_pool1d(4, 2, 2)  discriminator/melgan.py:22
    for L in (67, 64, 5):
        x = torch.from_numpy(rnd(2, 3, L)).requires_grad_(True)
        y = F.avg_pool1d(x, kernel_size=4, stride=2, padding=2)
        gy = torch.from_numpy(rnd(*y.shape))
        y.backward(gy)
        out.update({"pool_L%d/x" % L: t2n(x), "pool_L%d/y" % L: t2n(y),
                    "pool_L%d/gy" % L: t2n(gy), "pool_L%d/gx" % L: t2n(x.grad)})

    # losses (loss/loss.py)
    r = torch.from_numpy(rnd(3, 1, 17, scale=1.5)).requires_grad_(True)
    f = torch.from_numpy(rnd(3, 1, 17, scale=1.5)).requires_grad_(True)
    hd = ns.loss.hinge_discriminator_loss(r, f)
    hd.backward()
    out.update({"hinge_d/r": t2n(r), "hinge_d/f": t2n(f), "hinge_d/loss": t2n(hd),
                "hinge_d/gr": t2n(r.grad), "hinge_d/gf": t2n(f.grad)})
    f2 = torch.from_numpy(rnd(3, 1, 9)).requires_grad_(True)
    hg = ns.loss.hinge_generator_loss(f2)
    hg.backward()
    out.update({"hinge_g/f": t2n(f2), "hinge_g/loss": t2n(hg), "hinge_g/gf": t2n(f2.grad)})
    a = torch.from_numpy(rnd(2, 6, 33))
    b = torch.from_numpy(rnd(2, 6, 33)).requires_grad_(True)
    l1 = F.l1_loss(a, b)
    l1.backward()
    out.update({"l1/r": t2n(a), "l1/f": t2n(b), "l1/loss": t2n(l1), "l1/gf": t2n(b.grad)})
    out["ls/g"] = t2n(ns.loss.least_squares_generator_loss(f2))
    out["ls/d"] = t2n(ns.loss.least_squares_disc_loss(r, f))

    # the composite losses on small lists-of-lists
    rf = [[torch.from_numpy(rnd(2, 4, 9 + i + 3 * s)) for i in range(6)] for s in range(3)]
    ff = [[torch.from_numpy(rnd(2, 4, 9 + i + 3 * s)).requires_grad_(True) for i in range(6)]
          for s in range(3)]
    rj = [torch.from_numpy(rnd(2, 1, 5 + s, scale=1.2)).requires_grad_(True) for s in range(3)]
    fj = [torch.from_numpy(rnd(2, 1, 5 + s, scale=1.2)).requires_grad_(True) for s in range(3)]
    gl = ns.loss.mel_gan_gen_loss(rf, ff, rj, fj, gan_loss=ns.loss.hinge_generator_loss)
    gl.backward()
    out["genloss/loss"] = t2n(gl)
    for s in range(3):
        out["genloss/rj%d" % s], out["genloss/fj%d" % s] = t2n(rj[s]), t2n(fj[s])
        out["genloss/gfj%d" % s] = t2n(fj[s].grad)
        for i in range(6):
            out["genloss/rf%d_%d" % (s, i)] = t2n(rf[s][i])
            out["genloss/ff%d_%d" % (s, i)] = t2n(ff[s][i])
            out["genloss/gff%d_%d" % (s, i)] = t2n(ff[s][i].grad)
    for t in fj:
        t.grad = None
    dl = ns.loss.mel_gan_disc_loss(rj, fj, gan_loss=ns.loss.hinge_discriminator_loss)
    dl.backward()
    out["discloss/loss"] = t2n(dl)
    for s in range(3):
        out["discloss/grj%d" % s] = t2n(rj[s].grad)
        out["discloss/gfj%d" % s] = t2n(fj[s].grad)

    # Adam as configured at experiment/experiment.py:111-117
    p = torch.nn.Parameter(torch.from_numpy(rnd(257, scale=0.02)))
    opt = torch.optim.Adam([p], lr=1e-4, betas=(0.5, 0.9))
    out["adam/p0"] = t2n(p)
    for i in range(3):
        g = rnd(257, scale=10.0 ** (-i * 3))
        out["adam/g%d" % i] = g
        p.grad = torch.from_numpy(g.copy())
        opt.step()
        out["adam/p%d" % (i + 1)] = t2n(p)
    np.savez_compressed(os.path.join(OUT, "ops_tiny.npz"), **out)
    print("ops_tiny: %d arrays" % len(out))


# --------------------------------------------------------------------- g_fwd

def g_fwd(ns):
    out = {}
    g = ns.MelGanGenerator(32, 80)
    shapes = SYN.module_param_shapes(g)
    # config 2: seed-7 weights, zero biases, features default_rng(1)
    sd = SYN.synthetic_state_dict(shapes, seed=7)
    load_sd(g, sd)
    feat = np.random.default_rng(1).standard_normal((1, 80, 32)).astype(np.float32)
    with torch.no_grad():
        out["cfg2/y_ref32"] = t2n(g(torch.from_numpy(feat)))
        out["cfg2/y_ref64"] = t2n(copy.deepcopy(g).double()(torch.from_numpy(feat).double()))
    # non-zero biases, larger weights (keeps tanh/lrelu away from the linear regime), short input
    sd2 = SYN.synthetic_state_dict(shapes, seed=11, weight_scale=0.05, bias_scale=0.05)
    load_sd(g, sd2)
    feat2 = np.random.default_rng(2).standard_normal((2, 80, 5)).astype(np.float32)
    with torch.no_grad():
        out["short/y_ref32"] = t2n(g(torch.from_numpy(feat2)))
        out["short/y_ref64"] = t2n(copy.deepcopy(g).double()(torch.from_numpy(feat2).double()))
    # 128-mel variant (experiment/melgan.py:23 uses 128), inference-length input
    g128 = ns.MelGanGenerator(32, 128)
    sd3 = SYN.synthetic_state_dict(SYN.module_param_shapes(g128), seed=7, bias_scale=0.02)
    load_sd(g128, sd3)
    feat3 = np.random.default_rng(3).standard_normal((1, 128, 7)).astype(np.float32)
    with torch.no_grad():
        out["mel128/y_ref32"] = t2n(g128(torch.from_numpy(feat3)))
    out["param_names"] = np.array([k for k, _ in shapes])
    np.savez_compressed(os.path.join(OUT, "g_fwd.npz"), **out)
    print("g_fwd: cfg2 |y|max=%.4g  short |y|max=%.4g" %
          (np.abs(out["cfg2/y_ref32"]).max(), np.abs(out["short/y_ref32"]).max()))


# --------------------------------------------------------------------- d_fwd

def d_fwd(ns):
    out = {}
    d = ns.MelGanDiscriminator()
    shapes = SYN.module_param_shapes(d)
    out["param_names"] = np.array([k for k, _ in shapes])
    for tag, kw, x in (
            ("cfg", dict(seed=7), SYN.synthetic_samples(1)),
            ("big", dict(seed=13, weight_scale=0.08, bias_scale=0.1),
             SYN.synthetic_samples(2, 3000, rank=5))):
        load_sd(d, SYN.synthetic_state_dict(shapes, **kw))
        d64 = copy.deepcopy(d).double()
        with torch.no_grad():
            feats, judges = d(torch.from_numpy(x))
            feats64, judges64 = d64(torch.from_numpy(x).double())
        for s in range(3):
            out["%s/j%d_ref32" % (tag, s)] = t2n(judges[s])
            out["%s/j%d_ref64" % (tag, s)] = t2n(judges64[s])
            for i in range(6):
                out["%s/f%d_%d_shape" % (tag, s, i)] = np.array(feats[s][i].shape, np.int64)
                out["%s/f%d_%d_sum_ref32" % (tag, s, i)] = summary(t2n(feats[s][i]))
                out["%s/f%d_%d_sum_ref64" % (tag, s, i)] = summary(t2n(feats64[s][i]))
                out["%s/f%d_%d_smp_ref32" % (tag, s, i)] = SYN.strided_sample(t2n(feats[s][i]))
                out["%s/f%d_%d_smp_ref64" % (tag, s, i)] = SYN.strided_sample(t2n(feats64[s][i]))
    np.savez_compressed(os.path.join(OUT, "d_fwd.npz"), **out)
    print("d_fwd: judgement shapes", [tuple(out["cfg/j%d_ref32" % s].shape) for s in range(3)])


# --------------------------------------------------------------------- train

def train(ns):
    out = {}
    for tag, B, T, gkw, dkw in (
            ("small", 2, 8, dict(seed=7, bias_scale=0.02), dict(seed=8, bias_scale=0.02)),
            ("cfg3", 1, 32, dict(seed=7), dict(seed=7))):
        L = T * 256
        g = ns.MelGanGenerator(32, 80)
        d = ns.MelGanDiscriminator()
        gshapes, dshapes = SYN.module_param_shapes(g), SYN.module_param_shapes(d)
        load_sd(g, SYN.synthetic_state_dict(gshapes, **gkw))
        load_sd(d, SYN.synthetic_state_dict(dshapes, **dkw))
        d2 = TwoArgD(d)
        # Adam config: experiment/experiment.py:111-117
        g_optim = torch.optim.Adam(g.parameters(), lr=1e-4, betas=(0.5, 0.9))
        d_optim = torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.9))
        gt = ns.train.GeneratorTrainer(g, g_optim, d2, d_optim, ns.loss.mel_gan_gen_loss)
        dt = ns.train.DiscriminatorTrainer(g, g_optim, d2, d_optim, ns.loss.mel_gan_disc_loss)
        nsteps = 4 if tag == "small" else 2
        losses = []
        for step in range(nsteps):
            samples = torch.from_numpy(SYN.synthetic_samples(B, L, rank=step))
            feats = torch.from_numpy(SYN.synthetic_features(B, 80, T, rank=step))
            if step % 2 == 0:  # D first: experiment/experiment.py:141-144
                r = dt.train(samples, feats)
                losses.append(r["d_loss"])
                if step == 0:
                    for k, p in d.named_parameters():
                        out["%s/dgrad_sum/%s" % (tag, k)] = summary(t2n(p.grad))
                        out["%s/dgrad_smp/%s" % (tag, k)] = SYN.strided_sample(t2n(p.grad))
            else:
                r = gt.train(samples, feats)
                losses.append(r["g_loss"])
                if step == 1:
                    out["%s/fake_sum" % tag] = summary(r["fake"])
                    out["%s/fake_smp" % tag] = SYN.strided_sample(r["fake"])
                    for k, p in g.named_parameters():
                        out["%s/ggrad_sum/%s" % (tag, k)] = summary(t2n(p.grad))
                        out["%s/ggrad_smp/%s" % (tag, k)] = SYN.strided_sample(t2n(p.grad))
        out["%s/losses" % tag] = np.array(losses, np.float64)
        out["%s/cfg" % tag] = np.array([B, T, nsteps], np.int64)
        for k, v in g.state_dict().items():
            out["%s/gparam_sum/%s" % (tag, k)] = summary(t2n(v))
            out["%s/gparam_smp/%s" % (tag, k)] = SYN.strided_sample(t2n(v))
        for k, v in d.state_dict().items():
            out["%s/dparam_sum/%s" % (tag, k)] = summary(t2n(v))
            out["%s/dparam_smp/%s" % (tag, k)] = SYN.strided_sample(t2n(v))
        print("train[%s]: losses" % tag, losses)
    np.savez_compressed(os.path.join(OUT, "train.npz"), **out)


# ----------------------------------------------------------------- audio2mel

def audio2mel():
    """feature/feature.py:11-59 cannot be imported (librosa / lmdb absent; legacy torch.stft
    call).  Expected values restate :44-58 with torch.stft(return_complex=True); the mel
    basis (librosa.filters.mel, third-party) is taken from transformers' slaney filter bank."""
    from transformers.audio_utils import mel_filter_bank
    out = {}
    for n_mel in (80, 128):
        fb = mel_filter_bank(num_frequency_bins=513, num_mel_filters=n_mel, min_frequency=0.0,
                             max_frequency=22050 / 2.0, sampling_rate=22050, norm="slaney",
                             mel_scale="slaney")
        basis = np.ascontiguousarray(fb.T).astype(np.float32)  # (n_mel, 513)
        out["basis%d" % n_mel] = basis
        x = np.random.default_rng(0).uniform(-0.95, 0.95, 22050).astype(np.float32)
        audio = torch.from_numpy(x).view(1, 1, -1)
        p = (1024 - 256) // 2
        audio = F.pad(audio, (0, p)).squeeze(1)
        fft = torch.stft(audio, n_fft=1024, hop_length=256, win_length=1024,
                         window=torch.hann_window(1024).float(), center=False,
                         return_complex=True)
        mag = torch.sqrt(fft.real ** 2 + fft.imag ** 2)
        mel = torch.matmul(torch.from_numpy(basis), mag)
        out["logmel%d" % n_mel] = t2n(torch.log10(torch.clamp(mel, min=1e-5)))
    out["hann1024"] = t2n(torch.hann_window(1024).float())
    np.savez_compressed(os.path.join(OUT, "audio2mel.npz"), **out)
    print("audio2mel: logmel80", out["logmel80"].shape)


# ------------------------------------------------------------------ realmelgan

def realmelgan():
    """experiment/realmelgan.py (weight-normed MelGAN): forward passes and one
    D-step + one G-step of the reference's own trainers."""
    rm = ref_import.load_realmelgan()
    ns = ref_import.load_reference()
    out = {}
    g = rm.Generator(128, 32, 3)
    d = rm.Discriminator(3, 16, 4, 4)
    gshapes, dshapes = SYN.module_param_shapes(g), SYN.module_param_shapes(d)
    out["g_param_names"] = np.array([k for k, _ in gshapes])
    out["d_param_names"] = np.array([k for k, _ in dshapes])
    load_sd(g, SYN.synthetic_state_dict(gshapes, seed=21, weight_scale=0.3, bias_scale=0.05))
    load_sd(d, SYN.synthetic_state_dict(dshapes, seed=22, weight_scale=0.3, bias_scale=0.05))
    feat = np.random.default_rng(5).standard_normal((2, 128, 6)).astype(np.float32)
    with torch.no_grad():
        y = g(torch.from_numpy(feat))
        out["g/y_ref32"] = t2n(y)
        out["g/y_ref64"] = t2n(copy.deepcopy(g).double()(torch.from_numpy(feat).double()))
        x = SYN.synthetic_samples(2, 2048, rank=9)
        feats, judges = d(torch.from_numpy(x), None)
        feats64, judges64 = copy.deepcopy(d).double()(torch.from_numpy(x).double(), None)
    for s in range(3):
        out["d/j%d_ref32" % s] = t2n(judges[s])
        out["d/j%d_ref64" % s] = t2n(judges64[s])
        for i in range(6):
            out["d/f%d_%d_shape" % (s, i)] = np.array(feats[s][i].shape, np.int64)
            out["d/f%d_%d_smp_ref32" % (s, i)] = SYN.strided_sample(t2n(feats[s][i]))
            out["d/f%d_%d_smp_ref64" % (s, i)] = SYN.strided_sample(t2n(feats64[s][i]))
    # one D-step and one G-step from the same parameters (fresh copies), reference trainers
    B, T = 2, 8
    samples = torch.from_numpy(SYN.synthetic_samples(B, T * 256, rank=3))
    feats_in = torch.from_numpy(SYN.synthetic_features(B, 128, T, rank=3))
    for kind in ("d", "g"):
        g2, d2 = copy.deepcopy(g), copy.deepcopy(d)
        go = torch.optim.Adam(g2.parameters(), lr=1e-4, betas=(0.5, 0.9))
        do = torch.optim.Adam(d2.parameters(), lr=1e-4, betas=(0.5, 0.9))
        if kind == "d":
            tr = ns.train.DiscriminatorTrainer(g2, go, d2, do, rm.mel_gan_disc_loss)
            r = tr.train(samples, feats_in)
            out["step/d_loss"] = np.array([r["d_loss"]])
            net = d2
        else:
            tr = ns.train.GeneratorTrainer(g2, go, d2, do, rm.mel_gan_gen_loss)
            r = tr.train(samples, feats_in)
            out["step/g_loss"] = np.array([r["g_loss"]])
            out["step/fake_smp"] = SYN.strided_sample(r["fake"])
            net = g2
        for k, p in net.named_parameters():
            out["step/%sgrad_sum/%s" % (kind, k)] = summary(t2n(p.grad))
            out["step/%sgrad_smp/%s" % (kind, k)] = SYN.strided_sample(t2n(p.grad))
    np.savez_compressed(os.path.join(OUT, "realmelgan.npz"), **out)
    print("realmelgan: |y|max %.4g, d_loss %.6f, g_loss %.6f" % (np.abs(out["g/y_ref32"]).max(),
                                                                 out["step/d_loss"][0], out["step/g_loss"][0]))


def stage1():
    """Stage-1 2-D conv mel GAN (SURVEY.md 8(f) row 2 / BASELINE config 5): forward passes of the reference's
    SpectrogramFeatureGenerator / SpectrogramFeatureDiscriminator and one D-step + one G-step of its own
    trainers with the least-squares losses of experiment/featureexperiment.py:289-293."""
    ns = ref_import.load_stage1()
    out = {}
    g = ns.SpectrogramFeatureGenerator(out_channels=128, noise_dim=128)
    d = ns.SpectrogramFeatureDiscriminator(feature_channels=128, channels=256)
    gshapes, dshapes = SYN.module_param_shapes(g), SYN.module_param_shapes(d)
    out["g_param_names"] = np.array([k for k, _ in gshapes])
    out["g_param_shapes"] = np.array([str(tuple(v)) for _, v in gshapes])
    out["d_param_names"] = np.array([k for k, _ in dshapes])
    out["d_param_shapes"] = np.array([str(tuple(v)) for _, v in dshapes])
    load_sd(g, SYN.synthetic_state_dict(gshapes, seed=31, weight_scale=0.03, bias_scale=0.02))
    load_sd(d, SYN.synthetic_state_dict(dshapes, seed=32, weight_scale=0.03, bias_scale=0.02))
    B = 2
    z = np.random.default_rng(6).standard_normal((B, 128, 1)).astype(np.float32)
    real = (np.random.default_rng(7).standard_normal((B, 128, 512)) * 0.5).astype(np.float32)
    with torch.no_grad():
        y = g(torch.from_numpy(z))
        y64 = copy.deepcopy(g).double()(torch.from_numpy(z).double())
        feats, judge = d(torch.from_numpy(real), None)
        feats64, judge64 = copy.deepcopy(d).double()(torch.from_numpy(real).double(), None)
    out["g/shape"] = np.array(y.shape, np.int64)
    out["g/y_smp_ref32"] = SYN.strided_sample(t2n(y), 8192)
    out["g/y_smp_ref64"] = SYN.strided_sample(t2n(y64), 8192)
    out["g/y_sum_ref64"] = summary(t2n(y64))
    out["d/j_ref32"], out["d/j_ref64"] = t2n(judge), t2n(judge64)
    for i, (f32, f64) in enumerate(zip(feats, feats64)):
        out["d/f%d_shape" % i] = np.array(f32.shape, np.int64)
        out["d/f%d_smp_ref32" % i] = SYN.strided_sample(t2n(f32), 2048)
        out["d/f%d_smp_ref64" % i] = SYN.strided_sample(t2n(f64), 2048)

    def gen_loss(r_features, f_features, r_score, f_score, gan_loss):
        return ns.loss.least_squares_generator_loss(f_score)

    def disc_loss(r_score, f_score, gan_loss):
        return ns.loss.least_squares_disc_loss(r_score, f_score)

    for kind in ("d", "g"):
        g2, d2 = copy.deepcopy(g), copy.deepcopy(d)
        go = torch.optim.Adam(g2.parameters(), lr=1e-4, betas=(0.5, 0.9))
        do = torch.optim.Adam(d2.parameters(), lr=1e-4, betas=(0.5, 0.9))
        if kind == "d":
            tr = ns.train.DiscriminatorTrainer(g2, go, d2, do, disc_loss, sub_loss=None)
            r = tr.train(torch.from_numpy(real), torch.from_numpy(z))
            out["step/d_loss"] = np.array([r["d_loss"]])
            net = d2
        else:
            tr = ns.train.GeneratorTrainer(g2, go, d2, do, gen_loss, sub_loss=None)
            r = tr.train(torch.from_numpy(real), torch.from_numpy(z))
            out["step/g_loss"] = np.array([r["g_loss"]])
            out["step/fake_smp"] = SYN.strided_sample(r["fake"], 8192)
            net = g2
        for k, p in net.named_parameters():
            out["step/%sgrad_sum/%s" % (kind, k)] = summary(t2n(p.grad))
            out["step/%sgrad_smp/%s" % (kind, k)] = SYN.strided_sample(t2n(p.grad))
    np.savez_compressed(os.path.join(OUT, "stage1.npz"), **out)
    print("stage1: |y|max %.4g, d_loss %.6f, g_loss %.6f" % (np.abs(out["g/y_smp_ref32"]).max(),
                                                              out["step/d_loss"][0], out["step/g_loss"][0]))


def partial_rows():
    """Rows finished in round 2: the weight-normed ResidualAtom (util/modules.py:350-388 with add_weight_norm) and
    the conditioning branch `layer_cond` of the weight-normed MelGAN's discriminators (realmelgan.py:128-152),
    both from the imported reference classes."""
    ns = ref_import.load_reference()
    rm = ref_import.load_realmelgan()
    out = {}
    rng = np.random.default_rng(41)
    for d in (1, 9):
        atom = ns.ResidualAtom(8, d, True)
        sd = SYN.synthetic_state_dict(SYN.module_param_shapes(atom), seed=40 + d, weight_scale=0.3, bias_scale=0.1)
        load_sd(atom, sd)
        out["atom_wn_d%d/param_names" % d] = np.array(list(sd.keys()))
        for k, v in sd.items():
            out["atom_wn_d%d/sd/%s" % (d, k)] = v
        x = torch.from_numpy(rng.standard_normal((2, 8, 37)).astype(np.float32)).requires_grad_(True)
        y = atom(x)
        gy = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
        y.backward(gy)
        out["atom_wn_d%d/x" % d], out["atom_wn_d%d/y" % d] = t2n(x), t2n(y)
        out["atom_wn_d%d/gy" % d], out["atom_wn_d%d/gx" % d] = t2n(gy), t2n(x.grad)
        for k, p in atom.named_parameters():
            out["atom_wn_d%d/grad/%s" % (d, k)] = t2n(p.grad)
    disc = rm.Discriminator(3, 16, 4, 4, conditioning_channels=128)
    dshapes = SYN.module_param_shapes(disc)
    out["cond/param_names"] = np.array([k for k, _ in dshapes])
    load_sd(disc, SYN.synthetic_state_dict(dshapes, seed=45, weight_scale=0.3, bias_scale=0.05))
    x = torch.from_numpy(SYN.synthetic_samples(2, 2048, rank=11)).requires_grad_(True)
    feat = torch.from_numpy(SYN.synthetic_features(2, 128, 8, rank=11)).requires_grad_(True)
    feats, judges = disc(x, feat)
    loss = sum(j.mean() for j in judges) + 0.1 * sum(f.abs().mean() for grp in feats for f in grp)
    loss.backward()
    out["cond/loss"] = np.array([loss.item()])
    out["cond/nfeat"] = np.array([len(feats[0])])
    for s in range(3):
        out["cond/j%d" % s] = t2n(judges[s])
        for i, f in enumerate(feats[s]):
            out["cond/f%d_%d_shape" % (s, i)] = np.array(f.shape, np.int64)
            out["cond/f%d_%d_smp" % (s, i)] = SYN.strided_sample(t2n(f))
    out["cond/gx_smp"], out["cond/gfeat"] = SYN.strided_sample(t2n(x.grad), 1024), t2n(feat.grad)
    for k, p in disc.named_parameters():
        if "cond" in k or "layer_6" in k:
            out["cond/grad_smp/" + k] = SYN.strided_sample(t2n(p.grad))
            out["cond/grad_sum/" + k] = summary(t2n(p.grad))
    np.savez_compressed(os.path.join(OUT, "partial_rows.npz"), **out)
    print("partial_rows: cond loss %.6f, %d features per scale" % (loss.item(), len(feats[0])))


def main():
    os.makedirs(OUT, exist_ok=True)
    ns = ref_import.load_reference()
    which = sys.argv[1:] or ["ops", "g", "d", "train", "mel", "real", "stage1", "partial"]
    if "ops" in which:
        ops_tiny(ns)
    if "g" in which:
        g_fwd(ns)
    if "d" in which:
        d_fwd(ns)
    if "train" in which:
        train(ns)
    if "mel" in which:
        audio2mel()
    if "real" in which:
        realmelgan()
    if "stage1" in which:
        stage1()
    if "partial" in which:
        partial_rows()


if __name__ == "__main__":
    main()
