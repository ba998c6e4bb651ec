"""Fused ResidualAtom forward (csrc/atom_fused.hip, one launch per atom) -- reference util/modules.py:350-388,384-388:
x + lrelu(conv_k3_d1(lrelu(conv_k3_dil(x)))) -- against the CPU oracle and against the two row-tile launches it
replaces, at the generator's channel counts, dilations 1 / 3 / 9, aligned and ragged tile counts, single and multiple
batch rows, inference (no saved activations) and training mode.

r04: the fused kernel multiplies block-scaled two-piece fp16 operands (22 significand bits, three products per multiply,
fp32 accumulation) where the row-tile launches multiply exact three-piece bf16 operands (six products): the two paths no
longer agree bitwise but to what two fp32 summation orders differ by (gate 2e-6; measured 1-7e-7).  What stays bitwise:
inference == training output, launch-to-launch determinism.  New gates: float64 accuracy at the bench shapes and
invariance of the relative error under input scales from 1e-12 to 1e+6 (the block scaling)."""
import numpy as np
import pytest

from conftest import rel_l2, stable_seed

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def _inputs(name, B, C, Lg):
    rng = np.random.default_rng(stable_seed(name))
    x = rng.standard_normal((B, C, Lg)).astype(np.float32)
    sc = 1.0 / np.sqrt(3 * C)            # O(1) activations through both convs
    w0 = (rng.standard_normal((C, C, 3)) * sc).astype(np.float32)
    w1 = (rng.standard_normal((C, C, 3)) * sc).astype(np.float32)
    b0 = (rng.standard_normal((C,)) * 0.1).astype(np.float32)
    b1 = (rng.standard_normal((C,)) * 0.1).astype(np.float32)
    return x, w0, b0, w1, b1


# name, B, C, L, dil   (tile = 124 output columns for C = 32 / 64, 60 for C = 128 / 256; 60 / 28 when B * L is small)
ATOM_CASES = [("c32_d1_one_tile", 1, 32, 124, 1), ("c32_d9_ragged", 2, 32, 1032, 9), ("c32_d3_short", 3, 32, 8, 3),
              ("c64_d1", 1, 64, 516, 1), ("c64_d3_two_rows", 2, 64, 248, 3), ("c64_d9_tail4", 1, 64, 252, 9),
              ("c128_d1", 1, 128, 300, 1), ("c128_d3", 2, 128, 64, 3), ("c128_d9_ragged", 1, 128, 188, 9),
              # the generator's own row lengths (several batch rows): the two-launch path runs its split-bf16 kernels here
              ("c32_l8192_d9", 3, 32, 8192, 9), ("c64_l4096_d3", 3, 64, 4096, 3), ("c128_l2048_d1", 3, 128, 2048, 1),
              # 256 channels (eight waves per workgroup): the narrow (B = 1 style) and the wide tiling
              ("c256_d1_b1", 1, 256, 256, 1), ("c256_d9_ragged", 2, 256, 100, 9), ("c256_d3_wide", 40, 256, 256, 3),
              # few columns in total: the narrow tiles of the latency-bound (B = 1) dispatch for C = 64 / 128
              ("c64_narrow_d9", 1, 64, 4096, 9), ("c128_narrow_d3", 1, 128, 2048, 3)]


@pytest.mark.parametrize("case", ATOM_CASES, ids=[c[0] for c in ATOM_CASES])
@pytest.mark.parametrize("save", [False, True], ids=["inference", "training"])
def test_fused_atom_vs_oracle_and_unfused(case, save, monkeypatch):
    from featuresynth._ops import graph as G
    from featuresynth._ops import prims as P
    from oracle import oracle as O
    name, B, C, Lg, dil = case
    x, w0, b0, w1, b1 = _inputs(name, B, C, Lg)
    xt, w0t, b0t, w1t, b1t = (dev(a) for a in (x, w0, b0, w1, b1))
    assert G.atom_fused_ok(xt.shape, w0t, b0t, b1t, dil), "the fused kernel must take the hot geometries"
    img = P.atom_image(C, xt.device)
    P.atom_pack([(w0t, w1t, img)])
    y, rec = G.atom_forward(xt, w0t, b0t, w1t, b1t, dil, save, image=img)
    # CPU oracle (double accumulation)
    t_ref = O.conv1d_fwd(x, w0, b0, 1, dil, dil, 1, O.PAD_ZERO, 1)
    u_ref = O.conv1d_fwd(t_ref, w1, b1, 1, 1, 1, 1, O.PAD_ZERO, 1)
    assert rel_l2(host(y), x + u_ref) < 1e-5
    # the two launches it replaces (exact bf16 x 3 products, or split-K slices / fp32-MFMA kernels on small grids: other
    # rounding sequences of the same accuracy)
    y2, rec2 = G.atom_forward(xt, w0t, b0t, w1t, b1t, dil, save, image=None)

    def agree(a, b_, what):
        assert rel_l2(host(a), host(b_)) < 2e-6, what
    agree(y, y2, "y")
    if save:
        assert rec[3] is not None and rec[4] is not None
        assert rel_l2(host(rec[3]), t_ref) < 1e-5 and rel_l2(host(rec[4]), u_ref) < 1e-5
        agree(rec[3], rec2[3], "t"); agree(rec[4], rec2[4], "y_act")
    else:
        assert rec[3] is None and rec[4] is None
    # MSYNTH_ATOM=0 switches the fused kernel off (tuning / test switch)
    monkeypatch.setenv("MSYNTH_ATOM", "0")
    assert not G.atom_fused_ok(xt.shape, w0t, b0t, b1t, dil)


def _f64_atom(x, w0, b0, w1, b1, dil):
    """The atom in float64 on the device (stock torch ops): the yardstick both kernel paths are measured against."""
    import torch.nn.functional as F
    xd = x.double()
    t = F.leaky_relu(F.conv1d(xd, w0.double(), b0.double(), padding=dil, dilation=dil), 0.2)
    u = F.leaky_relu(F.conv1d(t, w1.double(), b1.double(), padding=1), 0.2)
    return xd + u, t, u


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize("C,Lg,dil", [(32, 8192, 9), (64, 4096, 3), (128, 2048, 1), (128, 2048, 9), (256, 256, 9)])
def test_fused_atom_at_bench_shapes(C, Lg, dil):
    """BASELINE config 3's shapes (B = 32).  Against float64 the fused kernel (fp16 x 2, three products) is as close as
    the two row-tile launches (bf16 x 3, six exact products) -- both are bounded by fp32 accumulation, ~2-5e-7 -- and the
    two agree to 2e-6; inference and training mode give the same output bitwise, and so do two launches."""
    from featuresynth._ops import graph as G
    from featuresynth._ops import lib as L
    from featuresynth._ops import prims as P
    x, w0, b0, w1, b1 = _inputs("bench_%d" % C, 32, C, Lg)
    xt, w0t, b0t, w1t, b1t = (dev(a) for a in (x, w0, b0, w1, b1))
    img = P.atom_image(C, xt.device)
    P.atom_pack([(w0t, w1t, img)])
    y, rec = G.atom_forward(xt, w0t, b0t, w1t, b1t, dil, True, image=img)
    y2, rec2 = G.atom_forward(xt, w0t, b0t, w1t, b1t, dil, True, image=None)
    names = [L.load().ms_conv1d_kernel_name(rec2[k], 0).decode() for k in (0, 1)]
    assert all(n.startswith("k_conv_rows3") for n in names), names
    yr, tr, ur = _f64_atom(xt, w0t, b0t, w1t, b1t, dil)
    e_f = (_rel(rec[3], tr), _rel(rec[4], ur), _rel(y, yr))
    e_u = (_rel(rec2[3], tr), _rel(rec2[4], ur), _rel(y2, yr))
    print("C=%d vs float64: fused t %.2e u %.2e y %.2e | two launches t %.2e u %.2e y %.2e" % ((C,) + e_f + e_u))
    assert max(e_f) < 1e-6 and max(e_u) < 1e-6
    assert max(e_f) < 3 * max(e_u) + 1e-7, "the fused kernel must be as accurate as the exact-product path"
    assert _rel(rec[3], rec2[3]) < 2e-6 and _rel(rec[4], rec2[4]) < 2e-6 and _rel(y, y2) < 2e-6
    y3, _ = G.atom_forward(xt, w0t, b0t, w1t, b1t, dil, False, image=img)
    assert torch.equal(y3, y), "inference and training mode compute the same output"
    y4, rec4 = G.atom_forward(xt, w0t, b0t, w1t, b1t, dil, True, image=img)
    assert torch.equal(y4, y) and torch.equal(rec4[3], rec[3]) and torch.equal(rec4[4], rec[4]), "launch-to-launch determinism"


@pytest.mark.parametrize("C,Lg", [(32, 1032), (128, 300)])
@pytest.mark.parametrize("scale", [1e-12, 1e-6, 1.0, 1e6])
def test_fused_atom_block_scaling(C, Lg, scale):
    """The fp16 pieces are taken from a tile scaled by a power of two so that its largest magnitude sits at 2^14: the
    relative error against float64 must not depend on the magnitude of the data (fp16 alone spans 2^-24 .. 2^16).
    Biases scale with the input so that the outputs stay a homogeneous function of it; one input element per row is
    made 2^10 times larger than the rest (a tile whose maximum is an outlier)."""
    from featuresynth._ops import graph as G
    from featuresynth._ops import prims as P
    x, w0, b0, w1, b1 = _inputs("scale_%d" % C, 2, C, Lg)
    x[:, 0, 5] *= 1024.0
    xt, w0t, w1t = dev(x * scale), dev(w0), dev(w1)
    b0t, b1t = dev(b0 * scale), dev(b1 * scale)
    img = P.atom_image(C, xt.device)
    P.atom_pack([(w0t, w1t, img)])
    y, rec = G.atom_forward(xt, w0t, b0t, w1t, b1t, 3, True, image=img)
    yr, tr, ur = _f64_atom(xt, w0t, b0t, w1t, b1t, 3)
    e = (_rel(rec[3], tr), _rel(rec[4], ur), _rel(y, yr))
    print("C=%d scale %g: t %.2e u %.2e y %.2e" % ((C, scale) + e))
    assert max(e) < 2e-6, e
    # backward data with gradient-sized magnitudes
    g = dev(np.random.default_rng(7).standard_normal(x.shape).astype(np.float32) * scale * 1e-3)
    imgb = P.atom_image(C, xt.device)
    P.atom_pack([(w0t, w1t, imgb)], backward=True)
    gt, gx, _ = P.atom_bwd_data(g, rec[4], rec[3], imgb, 3)
    import torch.nn.functional as F
    gd = g.double() * torch.where(rec[4] > 0, 1.0, 0.2).double()
    gt_r = F.conv_transpose1d(gd, w1t.double(), padding=1)
    gx_r = F.conv_transpose1d(gt_r * torch.where(rec[3] > 0, 1.0, 0.2).double(), w0t.double(), padding=3, dilation=3)
    assert _rel(gt, gt_r) < 2e-6 and _rel(gx - g, gx_r) < 2e-6, (_rel(gt, gt_r), _rel(gx - g, gx_r))


def test_fused_atom_pack_is_per_call_and_multi():
    """All 12 atoms of a generator are packed by ONE launch per forward pass, from the weights as they are at that
    moment: a forward after an in-place weight change sees the new weights (no stale image)."""
    import featuresynth as fs
    from featuresynth._ops import lib as L
    from featuresynth._synthetic import module_param_shapes, synthetic_state_dict
    g = fs.MelGanGenerator(32, 80)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in
                       synthetic_state_dict(module_param_shapes(g), seed=7, bias_scale=0.02).items()})
    g.cuda()
    feat = dev(np.random.default_rng(1).standard_normal((2, 80, 4)).astype(np.float32))
    L.profile_begin()
    with torch.no_grad():
        y0 = g(feat)
    rec = L.profile_end()
    names = [r[0] for r in rec]
    assert names.count("ms_residual_atom_pack_multi") == 1
    # every atom of the four stacks is one launch, or a third of one: inference runs the 64- and 32-channel stacks whole
    assert names.count("ms_residual_stack_fwd") == 2 and names.count("ms_residual_atom_fwd") == 6, names
    with torch.no_grad():
        p = dict(g.named_parameters())["main.14.main.1.main.0.weight"]
        p.data.mul_(1.5)                         # a `.data` write: no version counter would notice it
        y1 = g(feat)
    assert not torch.equal(y0, y1)
    import os
    os.environ["MSYNTH_ATOM"] = "0"
    try:
        with torch.no_grad():
            y2 = g(feat)
    finally:
        del os.environ["MSYNTH_ATOM"]
    assert float((y1 - y2).norm() / y2.norm()) < 1e-6    # (small grids: the two-launch path runs other kernels)


BWD_CASES = list(ATOM_CASES)          # (r04: every case; r03 left 128 / 256 channels at dilation 9 to the two launches)


@pytest.mark.parametrize("case", BWD_CASES, ids=[c[0] for c in BWD_CASES])
def test_fused_atom_backward_vs_oracle_and_unfused(case):
    """Backward data of the atom in one launch (ms_residual_atom_bwd_data): gt = conv1^T(g lrelu'(u)) stored raw,
    gx = g + conv_d^T(gt lrelu'(t)) -- against the oracle's conv1d_bwd_data (masks from the device's own activations)
    and against the two ms_conv1d_bwd_data launches it replaces."""
    from featuresynth._ops import graph as G
    from featuresynth._ops import prims as P
    from oracle import oracle as O
    name, B, C, Lg, dil = case
    x, w0, b0, w1, b1 = _inputs(name, B, C, Lg)
    xt, w0t, b0t, w1t, b1t = (dev(a) for a in (x, w0, b0, w1, b1))
    assert P.atom_bwd_supported(B, C, Lg, dil)
    y, rec = G.atom_forward(xt, w0t, b0t, w1t, b1t, dil, True, image=None)
    d0, d1, _, t, u, _ = rec
    g = np.random.default_rng(stable_seed(name) + 1).standard_normal((B, C, Lg)).astype(np.float32)
    gt_d = dev(g)
    img = P.atom_image(C, xt.device)
    P.atom_pack([(w0t, w1t, img)], backward=True)
    gt, gx, _ = P.atom_bwd_data(gt_d, u, t, img, dil)
    # oracle
    gp1 = O.act_bwd(host(u), g, 1)
    gt_ref = O.conv1d_bwd_data(gp1, w1, x.shape, 1, 1, 1, 1, O.PAD_ZERO)
    gp0 = O.act_bwd(host(t), gt_ref, 1)
    gx_ref = O.conv1d_bwd_data(gp0, w0, x.shape, 1, dil, dil, 1, O.PAD_ZERO) + g
    assert rel_l2(host(gt), gt_ref) < 1e-5 and rel_l2(host(gx), gx_ref) < 1e-5
    # the two launches
    gt2 = P.conv1d_bwd_data(gt_d, u, w1t, d1)
    gx2 = P.conv1d_bwd_data(gt2, t, w0t, d0, gx_add=gt_d)
    assert rel_l2(host(gt), host(gt2)) < 2e-6 and rel_l2(host(gx), host(gx2)) < 2e-6


@pytest.mark.parametrize("C,Lg,dil", [(32, 8192, 9), (64, 4096, 3), (128, 2048, 1), (256, 256, 3)])
def test_fused_atom_backward_at_bench_shapes(C, Lg, dil):
    """Backward data at BASELINE config 3's shapes with gradient-sized inputs (1e-6): against float64 torch both paths sit at
    fp32 accumulation level, they agree to 2e-6, and the fused launch is deterministic."""
    import torch.nn.functional as F
    from featuresynth._ops import graph as G
    from featuresynth._ops import lib as L
    from featuresynth._ops import prims as P
    x, w0, b0, w1, b1 = _inputs("bench_%d" % C, 32, C, Lg)
    xt, w0t, b0t, w1t, b1t = (dev(a) for a in (x, w0, b0, w1, b1))
    y, rec = G.atom_forward(xt, w0t, b0t, w1t, b1t, dil, True, image=None)
    d0, d1, _, t, u, _ = rec
    g = dev(np.random.default_rng(5).standard_normal((32, C, Lg)).astype(np.float32) * 1e-6)
    img = P.atom_image(C, xt.device)
    P.atom_pack([(w0t, w1t, img)], backward=True)
    gt, gx, _ = P.atom_bwd_data(g, u, t, img, dil)
    gt2 = P.conv1d_bwd_data(g, u, w1t, d1)
    gx2 = P.conv1d_bwd_data(gt2, t, w0t, d0, gx_add=g)
    names = [L.load().ms_conv1d_kernel_name(dd, 1).decode() for dd in (d0, d1)]
    assert all(n.startswith("k_conv_rows3") for n in names), names
    gd = g.double() * torch.where(u > 0, 1.0, 0.2).double()
    gt_r = F.conv_transpose1d(gd, w1t.double(), padding=1)
    gx_r = F.conv_transpose1d(gt_r * torch.where(t > 0, 1.0, 0.2).double(), w0t.double(), padding=dil, dilation=dil)
    e_f = (_rel(gt, gt_r), _rel(gx - g, gx_r))
    e_u = (_rel(gt2, gt_r), _rel(gx2 - g, gx_r))
    print("C=%d backward vs float64: fused gt %.2e gx-g %.2e | two launches gt %.2e gx-g %.2e" % ((C,) + e_f + e_u))
    assert max(e_f) < 1e-6 and max(e_u) < 1e-6
    assert _rel(gt, gt2) < 2e-6 and _rel(gx - g, gx2 - g) < 2e-6
    gt3, gx3, _ = P.atom_bwd_data(g, u, t, img, dil)
    assert torch.equal(gt3, gt) and torch.equal(gx3, gx), "launch-to-launch determinism"


def _decode_signs(words):
    from oracle import torch_graph as TG
    return TG.decode_sign_words(words).cuda()


SIGN_CASES = [("s32", 2, 32, 1032, 3), ("s32_long", 3, 32, 8192, 9), ("s64", 2, 64, 4096, 1), ("s64_ragged", 1, 64, 252, 9),
              ("s128", 2, 128, 2048, 3), ("s128_short", 1, 128, 188, 9), ("s256", 3, 256, 256, 1), ("s256_narrow", 1, 256, 100, 9),
              # (B L >= 512 x 92 columns at 128 channels: the forward runs on 96-column tiles, the last tile of a row is partial)
              ("s128_wide_ragged", 24, 128, 2052, 9)]


@pytest.mark.parametrize("case", SIGN_CASES, ids=[c[0] for c in SIGN_CASES])
def test_fused_atom_sign_words(case):
    """r04: the training forward can save u as one sign bit per element (and t's signs beside t).  Same y and t bitwise, the
    words decode to exactly (u > 0) / (t > 0), and the backward data from the words equals the backward data from the fp32
    tensors bitwise."""
    from featuresynth._ops import prims as P
    name, B, C, Lg, dil = case
    x, w0, b0, w1, b1 = _inputs(name, B, C, Lg)
    xt, w0t, b0t, w1t, b1t = (dev(a) for a in (x, w0, b0, w1, b1))
    img = P.atom_image(C, xt.device); P.atom_pack([(w0t, w1t, img)])
    imgb = P.atom_image(C, xt.device); P.atom_pack([(w0t, w1t, imgb)], backward=True)
    y, t, u, _ = P.atom_fwd(xt, img, b0t, b1t, dil, True)
    ys, ts, su, aux = P.atom_fwd(xt, img, b0t, b1t, dil, True, signs=True)
    assert P.is_signs(su) and P.is_signs(aux.t_signs) and tuple(su.shape) == (B, C // 32, 2, Lg)
    assert torch.equal(ys, y) and torch.equal(ts, t)
    assert torch.equal(_decode_signs(su), u > 0) and torch.equal(_decode_signs(aux.t_signs), t > 0)
    g = dev(np.random.default_rng(stable_seed(name + "g")).standard_normal((B, C, Lg)) * 1e-3)
    gt, gx, _ = P.atom_bwd_data(g, u, t, imgb, dil)
    gts, gxs, _ = P.atom_bwd_data(g, su, ts, imgb, dil, t_signs=aux.t_signs)
    assert torch.equal(gts, gt) and torch.equal(gxs, gx)
    with pytest.raises(RuntimeError):
        P.atom_bwd_data(g, su, ts, imgb, dil)             # sign words of u without t's: refused, not guessed


@pytest.mark.parametrize("C,Lg,B", [(64, 4096, 2), (128, 2048, 2), (256, 256, 4), (32, 8192, 2), (32, 1024, 3)])
def test_stack_weight_gradients_from_sign_words(C, Lg, B):
    """The six weight gradients of a stack in one launch, LeakyReLU derivatives from sign words: bitwise what the fp32
    activations give (k_wgrad_rows3<., 2, true>; k_wgrad32<1, true> at 32 channels)."""
    from featuresynth._ops import graph as G
    from featuresynth._ops import prims as P
    rng = np.random.default_rng(stable_seed("wsig%d" % C))
    xt = dev(rng.standard_normal((B, C, Lg)))
    assert P.stack_signs_ok(xt, (1, 3, 9))
    jobs_f, jobs_s = [], []
    h = xt
    for dil in (1, 3, 9):
        _, w0, b0, w1, b1 = _inputs("wsig%d_%d" % (C, dil), 1, C, 8)
        w0t, b0t, w1t, b1t = (dev(a) for a in (w0, b0, w1, b1))
        img = P.atom_image(C, xt.device); P.atom_pack([(w0t, w1t, img)])
        imgb = P.atom_image(C, xt.device); P.atom_pack([(w0t, w1t, imgb)], backward=True)
        y, t, u, af = P.atom_fwd(h, img, b0t, b1t, dil, True)
        ys, ts, su, afs = P.atom_fwd(h, img, b0t, b1t, dil, True, signs=True)
        g = dev(rng.standard_normal((B, C, Lg)) * 1e-3)
        gt, _, ab = P.atom_bwd_data(g, u, t, imgb, dil)
        gts, _, abs_ = P.atom_bwd_data(g, su, ts, imgb, dil, t_signs=afs.t_signs)
        d0, _ = P.conv_desc(h.shape, w0t.shape, pad=dil, dil=dil, act=1)
        d1, _ = P.conv_desc(h.shape, w1t.shape, pad=1, act=1)
        jobs_f += [(t, g, u, d1, w1t.shape, None, None, False, af.amax[1], ab[0]),
                   (h, gt, t, d0, w0t.shape, None, None, False, af.amax[0], ab[1])]
        jobs_s += [(ts, g, su, d1, w1t.shape, None, None, False, afs.amax[1], abs_[0]),
                   (h, gts, afs.t_signs, d0, w0t.shape, None, None, False, afs.amax[0], abs_[1])]
        h = y
    rf = P.conv1d_bwd_weight_multi(jobs_f)
    rs = P.conv1d_bwd_weight_multi(jobs_s)
    for (gw, gb), (gws, gbs) in zip(rf, rs):
        assert torch.equal(gw, gws) and torch.equal(gb, gbs)
