"""CPU ORACLE (test infrastructure, NOT product code).

numpy/ctypes front-end of oracle/msynth_oracle.c plus a restatement, on top of
those primitives, of the reference's stage-2 GAN graph:

  * MelGanGenerator            /root/reference/featuresynth/generator/full.py:16-50
  * ResidualAtom / Stack       /root/reference/featuresynth/util/modules.py:350-405
  * FullDiscriminator          /root/reference/featuresynth/discriminator/full.py:10-40
  * MelGanDiscriminator        /root/reference/featuresynth/discriminator/melgan.py:7-27
  * hinge / feature-matching   /root/reference/featuresynth/loss/loss.py:9-78
  * D/G trainer steps          /root/reference/featuresynth/train/train.py:26-42,63-74
  * Adam(1e-4, (0.5, 0.9))     /root/reference/featuresynth/experiment/experiment.py:111-117
  * Audio2Mel                  /root/reference/featuresynth/feature/feature.py:11-59

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only as the checker.  It is pinned against the imported
reference through tests/golden/ (tools/make_golden.py, tests/test_oracle_golden.py).
The hand-written backward passes are additionally checked against torch autograd
of the imported reference via the golden gradients.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ACT_NONE, ACT_LRELU, ACT_TANH = 0, 1, 2
PAD_ZERO, PAD_REFLECT = 0, 1
SLOPE = 0.2

_fp = ctypes.POINTER(ctypes.c_float)
_i = ctypes.c_int
_sz = ctypes.c_size_t
_d = ctypes.c_double
_f = ctypes.c_float


def build(force=False):
    so = os.path.join(_HERE, "libmsynth_oracle.so")
    src = os.path.join(_HERE, "msynth_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libmsynth_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libmsynth_oracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        L.orc_conv1d_fwd.argtypes = [_fp, _fp, _fp, _fp, _fp] + [_i] * 11 + [_f]
        L.orc_act_bwd.argtypes = [_fp, _fp, _fp, _sz, _i, _f]
        L.orc_conv1d_bwd_data.argtypes = [_fp, _fp, _fp] + [_i] * 10
        L.orc_conv1d_bwd_weight.argtypes = [_fp, _fp, _fp, _fp] + [_i] * 10
        L.orc_conv_transpose1d_fwd.argtypes = [_fp, _fp, _fp, _fp] + [_i] * 8 + [_f]
        L.orc_conv_transpose1d_bwd_data.argtypes = [_fp, _fp, _fp] + [_i] * 7
        L.orc_conv_transpose1d_bwd_weight.argtypes = [_fp, _fp, _fp, _fp] + [_i] * 7
        L.orc_avg_pool1d_fwd.argtypes = [_fp, _fp] + [_i] * 5
        L.orc_avg_pool1d_bwd.argtypes = [_fp, _fp] + [_i] * 5
        L.orc_hinge_d.argtypes = [_fp, _fp, _sz, _fp, _fp, _d]
        L.orc_hinge_d.restype = _d
        L.orc_hinge_g.argtypes = [_fp, _sz, _fp, _d]
        L.orc_hinge_g.restype = _d
        L.orc_l1_mean.argtypes = [_fp, _fp, _sz, _fp, _d]
        L.orc_l1_mean.restype = _d
        L.orc_ls_g.argtypes = [_fp, _sz]
        L.orc_ls_g.restype = _d
        L.orc_ls_d.argtypes = [_fp, _fp, _sz]
        L.orc_ls_d.restype = _d
        L.orc_adam_step.argtypes = [_fp, _fp, _fp, _fp, _sz, _d, _d, _d, _d, _i]
        L.orc_mel_basis.argtypes = [_d, _i, _i, _d, _d, _fp]
        L.orc_hann_periodic.argtypes = [_i, _fp]
        L.orc_audio2mel_frames.argtypes = [_i, _i, _i]
        L.orc_audio2mel_frames.restype = _i
        L.orc_audio2mel.argtypes = [_fp, _i, _i, _fp, _i, _i, _fp, _i, _fp]
        _LIB = L
    return _LIB


def _c(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def _p(a):
    return None if a is None else a.ctypes.data_as(_fp)


# ----------------------------------------------------------------- primitives

def conv_out_len(Lin, K, stride, pad, dil):
    return (Lin + 2 * pad - dil * (K - 1) - 1) // stride + 1


def conv1d_fwd(x, w, bias=None, stride=1, pad=0, dil=1, groups=1, pad_mode=PAD_ZERO,
               act=ACT_NONE, res=None):
    x, w = _c(x), _c(w)
    B, Cin, Lin = x.shape
    Cout, Cg, K = w.shape
    assert Cg * groups == Cin and Cout % groups == 0
    Lout = conv_out_len(Lin, K, stride, pad, dil)
    y = np.empty((B, Cout, Lout), np.float32)
    bias = None if bias is None else _c(bias)
    res = None if res is None else _c(res)
    if res is not None:
        assert res.shape == y.shape
    lib().orc_conv1d_fwd(_p(x), _p(w), _p(bias), _p(res), _p(y), B, Cin, Lin, Cout, K, stride,
                         pad, dil, groups, pad_mode, act, SLOPE)
    return y


def act_bwd(y_act, gy, act):
    y_act, gy = _c(y_act), _c(gy)
    assert y_act.shape == gy.shape
    out = np.empty_like(gy)
    lib().orc_act_bwd(_p(y_act), _p(gy), _p(out), gy.size, act, SLOPE)
    return out


def conv1d_bwd_data(gy, w, x_shape, stride=1, pad=0, dil=1, groups=1, pad_mode=PAD_ZERO):
    gy, w = _c(gy), _c(w)
    B, Cin, Lin = x_shape
    Cout, Cg, K = w.shape
    assert gy.shape == (B, Cout, conv_out_len(Lin, K, stride, pad, dil))
    gx = np.empty((B, Cin, Lin), np.float32)
    lib().orc_conv1d_bwd_data(_p(gy), _p(w), _p(gx), B, Cin, Lin, Cout, K, stride, pad, dil,
                              groups, pad_mode)
    return gx


def conv1d_bwd_weight(x, gy, w_shape, stride=1, pad=0, dil=1, groups=1, pad_mode=PAD_ZERO):
    x, gy = _c(x), _c(gy)
    B, Cin, Lin = x.shape
    Cout, Cg, K = w_shape
    assert gy.shape == (B, Cout, conv_out_len(Lin, K, stride, pad, dil))
    gw = np.empty(w_shape, np.float32)
    gb = np.empty((Cout,), np.float32)
    lib().orc_conv1d_bwd_weight(_p(x), _p(gy), _p(gw), _p(gb), B, Cin, Lin, Cout, K, stride, pad,
                                dil, groups, pad_mode)
    return gw, gb


def convt_out_len(Lin, K, stride, pad):
    return (Lin - 1) * stride - 2 * pad + K


def conv_transpose1d_fwd(x, w, bias=None, stride=1, pad=0, act=ACT_NONE):
    x, w = _c(x), _c(w)
    B, Cin, Lin = x.shape
    Cin2, Cout, K = w.shape
    assert Cin2 == Cin
    y = np.empty((B, Cout, convt_out_len(Lin, K, stride, pad)), np.float32)
    bias = None if bias is None else _c(bias)
    lib().orc_conv_transpose1d_fwd(_p(x), _p(w), _p(bias), _p(y), B, Cin, Lin, Cout, K, stride,
                                   pad, act, SLOPE)
    return y


def conv_transpose1d_bwd_data(gy, w, x_shape, stride=1, pad=0):
    gy, w = _c(gy), _c(w)
    B, Cin, Lin = x_shape
    _, Cout, K = w.shape
    assert gy.shape == (B, Cout, convt_out_len(Lin, K, stride, pad))
    gx = np.empty((B, Cin, Lin), np.float32)
    lib().orc_conv_transpose1d_bwd_data(_p(gy), _p(w), _p(gx), B, Cin, Lin, Cout, K, stride, pad)
    return gx


def conv_transpose1d_bwd_weight(x, gy, w_shape, stride=1, pad=0):
    x, gy = _c(x), _c(gy)
    B, Cin, Lin = x.shape
    _, Cout, K = w_shape
    gw = np.empty(w_shape, np.float32)
    gb = np.empty((Cout,), np.float32)
    lib().orc_conv_transpose1d_bwd_weight(_p(x), _p(gy), _p(gw), _p(gb), B, Cin, Lin, Cout, K,
                                          stride, pad)
    return gw, gb


def avg_pool1d_fwd(x, k=4, s=2, p=2):
    x = _c(x)
    B, C, Lin = x.shape
    y = np.empty((B, C, (Lin + 2 * p - k) // s + 1), np.float32)
    lib().orc_avg_pool1d_fwd(_p(x), _p(y), B * C, Lin, k, s, p)
    return y


def avg_pool1d_bwd(gy, x_shape, k=4, s=2, p=2):
    gy = _c(gy)
    B, C, Lin = x_shape
    gx = np.empty(x_shape, np.float32)
    lib().orc_avg_pool1d_bwd(_p(gy), _p(gx), B * C, Lin, k, s, p)
    return gx


def hinge_d(r, f, want_grad=False, gscale=1.0):
    r, f = _c(r), _c(f)
    gr = np.empty_like(r) if want_grad else None
    gf = np.empty_like(f) if want_grad else None
    v = lib().orc_hinge_d(_p(r), _p(f), r.size, _p(gr), _p(gf), gscale)
    return (v, gr, gf) if want_grad else v


def hinge_g(f, want_grad=False, gscale=1.0):
    f = _c(f)
    gf = np.empty_like(f) if want_grad else None
    v = lib().orc_hinge_g(_p(f), f.size, _p(gf), gscale)
    return (v, gf) if want_grad else v


def l1_mean(r, f, want_grad=False, gscale=1.0):
    r, f = _c(r), _c(f)
    gf = np.empty_like(f) if want_grad else None
    v = lib().orc_l1_mean(_p(r), _p(f), r.size, _p(gf), gscale)
    return (v, gf) if want_grad else v


def ls_g(j):
    j = _c(j)
    return lib().orc_ls_g(_p(j), j.size)


def ls_d(r, f):
    r, f = _c(r), _c(f)
    return lib().orc_ls_d(_p(r), _p(f), r.size)


def adam_step(p, g, m, v, step, lr=1e-4, b1=0.5, b2=0.9, eps=1e-8):
    """In-place on p, m, v (float32 contiguous arrays)."""
    for a in (p, m, v):
        assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    g = _c(g)
    lib().orc_adam_step(_p(p), _p(g), _p(m), _p(v), p.size, lr, b1, b2, eps, step)


def mel_basis(sr=22050, n_fft=1024, n_mels=80, fmin=0.0, fmax=None):
    out = np.empty((n_mels, 1 + n_fft // 2), np.float32)
    lib().orc_mel_basis(float(sr), n_fft, n_mels, float(fmin), float(fmax or 0.0), _p(out))
    return out


def hann_periodic(n):
    out = np.empty((n,), np.float32)
    lib().orc_hann_periodic(n, _p(out))
    return out


def audio2mel(audio, n_fft=1024, hop=256, win=1024, sr=22050, n_mel=80, fmin=0.0, fmax=None):
    """audio (B, 1, N) or (N,) -> (B, n_mel, frames).  feature/feature.py:39-59."""
    a = _c(audio)
    if a.ndim == 1:
        a = a.reshape(1, 1, -1)
    B, _, N = a.shape
    assert win == n_fft
    frames = lib().orc_audio2mel_frames(N, n_fft, hop)
    out = np.empty((B, n_mel, frames), np.float32)
    window = hann_periodic(win)
    basis = mel_basis(sr, n_fft, n_mel, fmin, fmax)
    a2 = np.ascontiguousarray(a.reshape(B, N))
    lib().orc_audio2mel(_p(a2), B, N, _p(window), n_fft, hop, _p(basis), n_mel, _p(out))
    return out


# ------------------------------------------------------------- network graphs

# (convT name, stack name, Cin, Cout, K, stride, pad)   generator/full.py:27-41
_G_UPS = [
    ("main.3", "main.5", 512, 256, 16, 8, 4),
    ("main.6", "main.8", 256, 128, 16, 8, 4),
    ("main.9", "main.11", 128, 64, 4, 2, 1),
    ("main.12", "main.14", 64, 32, 4, 2, 1),
]
_DILATIONS = (1, 3, 9)  # hard-coded at util/modules.py:397-399

# (name, Cin, Cout, K, stride, pad, groups)   discriminator/full.py:14-19
_D_MAIN = [
    ("main.0", 1, 16, 15, 1, 7, 1),
    ("main.1", 16, 64, 41, 4, 20, 4),
    ("main.2", 64, 256, 41, 4, 20, 16),
    ("main.3", 256, 1024, 41, 4, 20, 64),
    ("main.4", 1024, 1024, 41, 4, 20, 256),
    ("main.5", 1024, 1024, 5, 1, 2, 1),
]


def generator_param_shapes(in_channels):
    """state_dict order and shapes of MelGanGenerator."""
    out = [("main.1.weight", (512, in_channels, 7)), ("main.1.bias", (512,))]
    for ct, st, cin, cout, K, s, p in _G_UPS:
        out += [(ct + ".weight", (cin, cout, K)), (ct + ".bias", (cout,))]
        for a in range(3):
            for c in range(2):
                out += [("%s.main.%d.main.%d.weight" % (st, a, c), (cout, cout, 3)),
                        ("%s.main.%d.main.%d.bias" % (st, a, c), (cout,))]
    out += [("main.15.weight", (1, 32, 7)), ("main.15.bias", (1,))]
    return out


def discriminator_param_shapes(prefix="disc."):
    out = []
    for name, cin, cout, K, s, p, g in _D_MAIN:
        out += [(prefix + name + ".weight", (cout, cin // g, K)), (prefix + name + ".bias", (cout,))]
    out += [(prefix + "judge.weight", (1, 1024, 3)), (prefix + "judge.bias", (1,))]
    return out


class Generator:
    def __init__(self, weights, in_channels=None):
        self.w = {k: _c(v) for k, v in weights.items()}
        self.in_channels = in_channels or self.w["main.1.weight"].shape[1]

    def forward(self, x, keep=True):
        w = self.w
        tape = []
        h = conv1d_fwd(x, w["main.1.weight"], w["main.1.bias"], pad=3, pad_mode=PAD_REFLECT,
                       act=ACT_LRELU)
        tape.append(("conv0", _c(x), h))
        for ct, st, cin, cout, K, s, p in _G_UPS:
            hin = h
            h = conv_transpose1d_fwd(hin, w[ct + ".weight"], w[ct + ".bias"], s, p, ACT_LRELU)
            tape.append(("convT", ct, hin, h, s, p))
            for a, d in enumerate(_DILATIONS):
                n0 = "%s.main.%d.main.0" % (st, a)
                n1 = "%s.main.%d.main.1" % (st, a)
                t = conv1d_fwd(h, w[n0 + ".weight"], w[n0 + ".bias"], pad=d, dil=d, act=ACT_LRELU)
                u = conv1d_fwd(t, w[n1 + ".weight"], w[n1 + ".bias"], pad=1, act=ACT_LRELU)
                out = (h + u).astype(np.float32)
                tape.append(("atom", n0, n1, d, h, t, u))
                h = out
        y = conv1d_fwd(h, w["main.15.weight"], w["main.15.bias"], pad=3, act=ACT_TANH)
        tape.append(("last", h, y))
        if keep:
            self.tape = tape
        return y

    def backward(self, gy):
        """Returns dict of parameter grads (d loss / d param) given d loss / d output."""
        w = self.w
        grads = {}
        g = _c(gy)
        for rec in reversed(self.tape):
            kind = rec[0]
            if kind == "last":
                _, h, y = rec
                gp = act_bwd(y, g, ACT_TANH)
                gw, gb = conv1d_bwd_weight(h, gp, w["main.15.weight"].shape, pad=3)
                grads["main.15.weight"], grads["main.15.bias"] = gw, gb
                g = conv1d_bwd_data(gp, w["main.15.weight"], h.shape, pad=3)
            elif kind == "atom":
                _, n0, n1, d, h, t, u = rec
                gp1 = act_bwd(u, g, ACT_LRELU)
                gw, gb = conv1d_bwd_weight(t, gp1, w[n1 + ".weight"].shape, pad=1)
                grads[n1 + ".weight"], grads[n1 + ".bias"] = gw, gb
                gt = conv1d_bwd_data(gp1, w[n1 + ".weight"], t.shape, pad=1)
                gp0 = act_bwd(t, gt, ACT_LRELU)
                gw, gb = conv1d_bwd_weight(h, gp0, w[n0 + ".weight"].shape, pad=d, dil=d)
                grads[n0 + ".weight"], grads[n0 + ".bias"] = gw, gb
                gh = conv1d_bwd_data(gp0, w[n0 + ".weight"], h.shape, pad=d, dil=d)
                g = (g + gh).astype(np.float32)
            elif kind == "convT":
                _, ct, hin, h, s, p = rec
                gp = act_bwd(h, g, ACT_LRELU)
                gw, gb = conv_transpose1d_bwd_weight(hin, gp, w[ct + ".weight"].shape, s, p)
                grads[ct + ".weight"], grads[ct + ".bias"] = gw, gb
                g = conv_transpose1d_bwd_data(gp, w[ct + ".weight"], hin.shape, s, p)
            elif kind == "conv0":
                _, x, h = rec
                gp = act_bwd(h, g, ACT_LRELU)
                gw, gb = conv1d_bwd_weight(x, gp, w["main.1.weight"].shape, pad=3,
                                           pad_mode=PAD_REFLECT)
                grads["main.1.weight"], grads["main.1.bias"] = gw, gb
        return grads


class FullDiscriminator:
    """One grouped strided conv discriminator (discriminator/full.py:10-40)."""

    def __init__(self, weights, prefix=""):
        self.w = weights
        self.prefix = prefix

    def forward(self, x):
        w, P = self.w, self.prefix
        feats = []
        h = _c(x)
        tape = []
        for name, cin, cout, K, s, p, g in _D_MAIN:
            hin = h
            h = conv1d_fwd(hin, w[P + name + ".weight"], w[P + name + ".bias"], stride=s, pad=p,
                           groups=g, act=ACT_LRELU)
            tape.append((name, hin, h, s, p, g))
            feats.append(h)
        j = conv1d_fwd(h, w[P + "judge.weight"], w[P + "judge.bias"], pad=1)
        return feats, j, tape

    def backward(self, tape, g_feats, g_judge, grads, need_gx=True):
        """Accumulates parameter grads into `grads`; returns d loss / d x."""
        w, P = self.w, self.prefix

        def acc(name, val):
            grads[name] = val if name not in grads else (grads[name] + val).astype(np.float32)

        h_last = tape[-1][2]
        gw, gb = conv1d_bwd_weight(h_last, g_judge, w[P + "judge.weight"].shape, pad=1)
        acc(P + "judge.weight", gw)
        acc(P + "judge.bias", gb)
        g = conv1d_bwd_data(g_judge, w[P + "judge.weight"], h_last.shape, pad=1)
        for li in range(len(tape) - 1, -1, -1):
            name, hin, h, s, p, grp = tape[li]
            if g_feats is not None and g_feats[li] is not None:
                g = (g + g_feats[li]).astype(np.float32)
            gp = act_bwd(h, g, ACT_LRELU)
            gw, gb = conv1d_bwd_weight(hin, gp, w[P + name + ".weight"].shape, stride=s, pad=p,
                                       groups=grp)
            acc(P + name + ".weight", gw)
            acc(P + name + ".bias", gb)
            if li > 0 or need_gx:
                g = conv1d_bwd_data(gp, w[P + name + ".weight"], hin.shape, stride=s, pad=p,
                                    groups=grp)
        return g


class MelGanDiscriminator:
    """Shared FullDiscriminator at 3 scales (discriminator/melgan.py:7-27)."""

    def __init__(self, weights):
        self.w = {k: _c(v) for k, v in weights.items()}
        self.disc = FullDiscriminator(self.w, "disc.")
        self.scales = 2

    def forward(self, x):
        feats, judges, tapes, xs = [], [], [], []
        h = _c(x)
        for s in range(self.scales + 1):
            if s > 0:
                h = avg_pool1d_fwd(h)
            xs.append(h)
            f, j, tape = self.disc.forward(h)
            feats.append(f)
            judges.append(j)
            tapes.append(tape)
        return feats, judges, (tapes, xs)

    def backward(self, ctx, g_feats, g_judges, grads, need_gx=True):
        tapes, xs = ctx
        gx_next = None
        for s in range(self.scales, -1, -1):
            gf = None if g_feats is None else g_feats[s]
            gx = self.disc.backward(tapes[s], gf, g_judges[s], grads,
                                    need_gx=need_gx or s > 0)
            if gx_next is not None:
                gx = (gx + avg_pool1d_bwd(gx_next, xs[s].shape)).astype(np.float32)
            gx_next = gx
        return gx_next


def mel_gan_disc_loss(r_j, f_j, want_grad=False):
    """loss/loss.py:21-25 with the hinge sub-loss (:17)."""
    total, grs, gfs = 0.0, [], []
    for r, f in zip(r_j, f_j):
        if want_grad:
            v, gr, gf = hinge_d(r, f, True)
            grs.append(gr)
            gfs.append(gf)
        else:
            v = hinge_d(r, f)
        total += v
    return (total, grs, gfs) if want_grad else total


def mel_gan_feature_loss(r_feats, f_feats, want_grad=False, gscale=1.0):
    """loss/loss.py:28-65."""
    nd = 1.0 / len(r_feats)
    total, grads = 0.0, []
    for rg, fg in zip(r_feats, f_feats):
        nl = 1.0 / len(rg)
        gl = []
        for r, f in zip(rg, fg):
            if want_grad:
                v, gf = l1_mean(r, f, True, gscale * nl * nd)
                gl.append(gf)
            else:
                v = l1_mean(r, f)
            total += nl * nd * v
        grads.append(gl)
    return (total, grads) if want_grad else total


def mel_gan_gen_loss(r_feats, f_feats, r_j, f_j, want_grad=False, feature_loss_weight=10.0):
    """loss/loss.py:68-78 with the hinge sub-loss (:9)."""
    j_loss, gj = 0.0, []
    for f in f_j:
        if want_grad:
            v, g = hinge_g(f, True)
            gj.append(g)
        else:
            v = hinge_g(f)
        j_loss += v
    if want_grad:
        f_loss, gfe = mel_gan_feature_loss(r_feats, f_feats, True, feature_loss_weight)
        return j_loss + feature_loss_weight * f_loss, gfe, gj
    return j_loss + feature_loss_weight * mel_gan_feature_loss(r_feats, f_feats)


class AdamState:
    """torch.optim.Adam(lr=1e-4, betas=(0.5, 0.9)) state, experiment/experiment.py:111-117."""

    def __init__(self, weights, lr=1e-4, b1=0.5, b2=0.9, eps=1e-8):
        self.m = {k: np.zeros_like(v) for k, v in weights.items()}
        self.v = {k: np.zeros_like(v) for k, v in weights.items()}
        self.t = 0
        self.hp = (lr, b1, b2, eps)

    def step(self, weights, grads):
        self.t += 1
        lr, b1, b2, eps = self.hp
        for k in weights:
            adam_step(weights[k], grads[k], self.m[k], self.v[k], self.t, lr, b1, b2, eps)


def d_step(gw, dw, d_adam, samples, features, update=True):
    """DiscriminatorTrainer.train (train/train.py:63-74).  Returns (d_loss, D grads)."""
    G, D = Generator(gw), MelGanDiscriminator(dw)
    fake = G.forward(features, keep=False)
    _, f_j, f_ctx = D.forward(fake)
    _, r_j, r_ctx = D.forward(samples)
    loss, grs, gfs = mel_gan_disc_loss(r_j, f_j, True)
    grads = {}
    D.backward(f_ctx, None, gfs, grads, need_gx=False)
    D.backward(r_ctx, None, grs, grads, need_gx=False)
    if update:
        d_adam.step(D.w, grads)
        for k in dw:
            dw[k] = D.w[k]
    return loss, grads


def g_step(gw, dw, g_adam, samples, features, update=True):
    """GeneratorTrainer.train (train/train.py:26-42).  Returns (g_loss, fake, G grads)."""
    G, D = Generator(gw), MelGanDiscriminator(dw)
    fake = G.forward(features, keep=True)
    f_feats, f_j, f_ctx = D.forward(fake)
    r_feats, r_j, _ = D.forward(samples)
    loss, gfe, gj = mel_gan_gen_loss(r_feats, f_feats, r_j, f_j, True)
    scratch = {}
    g_fake = D.backward(f_ctx, gfe, gj, scratch, need_gx=True)
    grads = G.backward(g_fake)
    if update:
        g_adam.step(G.w, grads)
        for k in gw:
            gw[k] = G.w[k]
    return loss, fake, grads


# ----------------------------------------------------------------- audio() front-end (feature/feature.py:64-71)
# librosa.resample (default res_type 'kaiser_best') = resampy.resample with its published kaiser_best filter;
# librosa.util.normalize(axis=-1) * 0.95.  librosa / resampy are third-party, un-pinned and absent in this
# container: the algorithm is restated from resampy's published sources (filters.sinc_window, interpn.resample_f)
# -- PARITY UNPINNED for this function.

def resample_kaiser_best(x, orig_sr, target_sr):
    from scipy.signal.windows import kaiser
    x = np.asarray(x, np.float64)
    ratio = float(target_sr) / float(orig_sr)
    num_zeros, precision, rolloff, beta = 64, 9, 0.9475937167399596, 14.769656459379492
    num_table = 2 ** precision
    n = num_table * num_zeros
    win = kaiser(2 * n + 1, beta)[n:] * rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    if ratio < 1:
        win = win * ratio
    win = win.astype(np.float32).astype(np.float64)          # (the table is stored in float32)
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    n_out = int(np.ceil(x.shape[-1] * ratio))
    scale = min(1.0, ratio)
    index_step = int(scale * num_table)
    nwin, n_in = win.shape[0], x.shape[-1]
    y = np.zeros(x.shape[:-1] + (n_out,), np.float64)
    for t in range(n_out):
        time_register = t / ratio
        nn = int(time_register)
        frac = scale * (time_register - nn)
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        i_max = min(nn + 1, (nwin - offset) // index_step)
        idx = offset + index_step * np.arange(i_max)
        y[..., t] += ((win[idx] + eta * delta[idx]) * x[..., nn - np.arange(i_max)]).sum(-1)
        frac = scale - frac
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        k_max = min(n_in - nn - 1, (nwin - offset) // index_step)
        idx = offset + index_step * np.arange(k_max)
        y[..., t] += ((win[idx] + eta * delta[idx]) * x[..., nn + 1 + np.arange(k_max)]).sum(-1)
    return y.astype(np.float32)


def audio_from_samples(x, orig_sr, samplerate):
    y = resample_kaiser_best(x, orig_sr, samplerate).astype(np.float64)
    peak = np.abs(y).max(axis=-1, keepdims=True)
    return (y / np.where(peak > np.finfo(np.float32).tiny, peak, 1.0) * 0.95).astype(np.float32)
