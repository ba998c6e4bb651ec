"""GPU parity tests (run with -m gpu on an MI355X): every kernel family through the C ABI
against the golden fixtures (imported reference) and the CPU oracle on seeded inputs.

Tolerances: all arithmetic is fp32; forward results must be within 1e-4 rel-L2 of the
reference (the north-star bound for the generator output; single ops land at ~1e-6),
gradients within 1e-3 rel-L2 per tensor."""
import ctypes

import numpy as np
import pytest

from conftest import rel_l2, stable_seed

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

FWD_TOL = 1e-5   # single op, fp32 accumulation order only
GRAD_TOL = 1e-4


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def ops(golden):
    return golden("ops_tiny")


def _conv_cases():
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ops_tiny.npz"))
    return sorted({k.split("/")[0] for k in z.files if k.endswith("/cfg") and k.startswith("conv_")})


def test_library_is_in_tree_and_loaded():
    import os
    from featuresynth._ops import lib
    L = lib.load()
    assert L.ms_version() >= 100
    assert os.path.realpath(lib.LIB_PATH).startswith(os.path.realpath(os.path.join(os.path.dirname(__file__), "..")))
    maps = open("/proc/self/maps").read()
    assert "libmsynth_hip.so" in maps
    assert len({ln.split()[-1] for ln in maps.splitlines() if "libamdhip64" in ln}) == 1, "two HIP runtimes loaded"


@pytest.mark.parametrize("name", _conv_cases())
def test_conv_golden(ops, name):
    from featuresynth._ops import functional as F_
    z = ops
    stride, pad, dil, groups, act, reflect = [int(v) for v in z[name + "/cfg"]]
    x = dev(z[name + "/x"]).requires_grad_(not reflect)
    w = dev(z[name + "/w"]).requires_grad_(True)
    b = dev(z[name + "/b"]).requires_grad_(True)
    y = F_.Conv1dFn.apply(x, w, b, stride, pad, dil, groups, 1 if reflect else 0, act)
    assert tuple(y.shape) == z[name + "/y"].shape
    assert rel_l2(host(y), z[name + "/y"]) < FWD_TOL
    if reflect:   # input gradient through a reflection pad is not on the hot path
        gw, gb = torch.autograd.grad(y, (w, b), dev(z[name + "/gy"]))
    else:
        gx, gw, gb = torch.autograd.grad(y, (x, w, b), dev(z[name + "/gy"]))
        assert rel_l2(host(gx), z[name + "/gx"]) < GRAD_TOL
    assert rel_l2(host(gw), z[name + "/gw"]) < GRAD_TOL
    assert rel_l2(host(gb), z[name + "/gb"]) < GRAD_TOL


@pytest.mark.parametrize("name", ["convt_k16_s8", "convt_k4_s2", "convt_k16_s8_l1"])
def test_convt_golden(ops, name):
    from featuresynth._ops import functional as F_
    z = ops
    stride, pad = [int(v) for v in z[name + "/cfg"]]
    x = dev(z[name + "/x"]).requires_grad_(True)
    w = dev(z[name + "/w"]).requires_grad_(True)
    b = dev(z[name + "/b"]).requires_grad_(True)
    y = F_.ConvTranspose1dFn.apply(x, w, b, stride, pad, 1)
    assert tuple(y.shape) == z[name + "/y"].shape
    assert rel_l2(host(y), z[name + "/y"]) < FWD_TOL
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), dev(z[name + "/gy"]))
    assert rel_l2(host(gx), z[name + "/gx"]) < GRAD_TOL
    assert rel_l2(host(gw), z[name + "/gw"]) < GRAD_TOL
    assert rel_l2(host(gb), z[name + "/gb"]) < GRAD_TOL


@pytest.mark.parametrize("d", [1, 3, 9])
def test_residual_atom_golden(ops, d):
    from featuresynth.util.modules import ResidualAtom
    z = ops
    nm = "atom_d%d" % d
    atom = ResidualAtom(8, d).cuda()
    atom.load_state_dict({k: torch.from_numpy(z[nm + "/sd/" + k]) for k in
                          ("main.0.weight", "main.0.bias", "main.1.weight", "main.1.bias")})
    x = dev(z[nm + "/x"]).requires_grad_(True)
    y = atom(x)
    assert rel_l2(host(y), z[nm + "/y"]) < FWD_TOL
    y.backward(dev(z[nm + "/gy"]))
    assert rel_l2(host(x.grad), z[nm + "/gx"]) < GRAD_TOL
    for k, p in atom.named_parameters():
        assert rel_l2(host(p.grad), z[nm + "/grad/" + k]) < GRAD_TOL, k


def test_residual_stack_golden(ops):
    from featuresynth.util.modules import ResidualStack
    z = ops
    st = ResidualStack(8, [1, 3, 9]).cuda()
    st.load_state_dict({k[len("stack/sd/"):]: torch.from_numpy(z[k]) for k in z.files
                        if k.startswith("stack/sd/")})
    assert rel_l2(host(st(dev(z["stack/x"]))), z["stack/y"]) < FWD_TOL


@pytest.mark.parametrize("L", [67, 64, 5])
def test_avg_pool_golden(ops, L):
    from featuresynth._ops import functional as F_
    z = ops
    p = "pool_L%d/" % L
    x = dev(z[p + "x"]).requires_grad_(True)
    y = F_.AvgPoolFn.apply(x)
    assert tuple(y.shape) == z[p + "y"].shape
    assert rel_l2(host(y), z[p + "y"]) < 1e-6
    y.backward(dev(z[p + "gy"]))
    assert rel_l2(host(x.grad), z[p + "gx"]) < 1e-6


def test_scalar_losses_golden(ops):
    from featuresynth import loss as LS
    z = ops
    r = dev(z["hinge_d/r"]).requires_grad_(True)
    f = dev(z["hinge_d/f"]).requires_grad_(True)
    v = LS.hinge_discriminator_loss(r, f)
    assert abs(v.item() - float(z["hinge_d/loss"])) < 1e-6
    v.backward()
    assert np.allclose(host(r.grad), z["hinge_d/gr"], atol=1e-8)
    assert np.allclose(host(f.grad), z["hinge_d/gf"], atol=1e-8)
    f2 = dev(z["hinge_g/f"]).requires_grad_(True)
    v = LS.hinge_generator_loss(f2)
    assert abs(v.item() - float(z["hinge_g/loss"])) < 1e-6
    v.backward()
    assert np.allclose(host(f2.grad), z["hinge_g/gf"], atol=1e-8)
    assert abs(LS.least_squares_generator_loss(dev(z["hinge_g/f"])).item() - float(z["ls/g"])) < 1e-6
    assert abs(LS.least_squares_disc_loss(dev(z["hinge_d/r"]), dev(z["hinge_d/f"])).item() - float(z["ls/d"])) < 1e-6


def test_composite_losses_golden(ops):
    from featuresynth import loss as LS
    z = ops
    rf = [[dev(z["genloss/rf%d_%d" % (s, i)]) for i in range(6)] for s in range(3)]
    ff = [[dev(z["genloss/ff%d_%d" % (s, i)]).requires_grad_(True) for i in range(6)] for s in range(3)]
    rj = [dev(z["genloss/rj%d" % s]).requires_grad_(True) for s in range(3)]
    fj = [dev(z["genloss/fj%d" % s]).requires_grad_(True) for s in range(3)]
    gl = LS.mel_gan_gen_loss(rf, ff, rj, fj, gan_loss=LS.hinge_generator_loss)
    ref = float(z["genloss/loss"])
    assert abs(gl.item() - ref) <= 1e-5 * abs(ref)
    gl.backward()
    for s in range(3):
        assert np.allclose(host(fj[s].grad), z["genloss/gfj%d" % s], atol=1e-8)
        for i in range(6):
            assert np.allclose(host(ff[s][i].grad), z["genloss/gff%d_%d" % (s, i)], rtol=1e-5, atol=1e-9)
    for t in fj:
        t.grad = None
    dl = LS.mel_gan_disc_loss(rj, fj, gan_loss=LS.hinge_discriminator_loss)
    assert abs(dl.item() - float(z["discloss/loss"])) < 1e-6
    dl.backward()
    for s in range(3):
        assert np.allclose(host(rj[s].grad), z["discloss/grj%d" % s], atol=1e-8)
        assert np.allclose(host(fj[s].grad), z["discloss/gfj%d" % s], atol=1e-8)
    # unfused fallbacks (arbitrary gan_loss callables) agree with the fused nodes
    gl2 = LS.mel_gan_gen_loss(rf, ff, rj, fj, gan_loss=lambda j: LS.hinge_generator_loss(j))
    assert abs(gl2.item() - ref) <= 1e-5 * abs(ref)
    fl = LS.mel_gan_feature_loss(rf, ff)
    assert fl.dim() == 0


def test_adam_golden(ops):
    from featuresynth._ops import prims as P
    z = ops
    n = z["adam/p0"].size
    pad = (n + 3) // 4 * 4
    p = torch.zeros(pad, device="cuda"); p[:n] = dev(z["adam/p0"])
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    for i in range(3):
        g = torch.zeros(pad, device="cuda"); g[:n] = dev(z["adam/g%d" % i])
        P.adam_step(p, g, m, v, step, 1e-4, 0.5, 0.9, 1e-8)
        assert rel_l2(host(p[:n]), z["adam/p%d" % (i + 1)]) < 1e-6
    assert int(step.item()) == 3


def test_audio2mel_golden(golden):
    from featuresynth.feature.feature import Audio2Mel
    z = golden("audio2mel")
    x = np.random.default_rng(0).uniform(-0.95, 0.95, 22050).astype(np.float32)
    for n_mel in (80, 128):
        a2m = Audio2Mel(n_mel_channels=n_mel).cuda()
        assert np.abs(host(a2m.mel_basis) - z["basis%d" % n_mel]).max() < 1e-6
        y = a2m(x)                                   # numpy in, as feature/feature.py:41-42 allows
        assert tuple(y.shape) == (1, n_mel, 84)
        assert np.abs(host(y) - z["logmel%d" % n_mel]).max() < 2e-4
        y2 = a2m(torch.from_numpy(x).view(1, 1, -1).cuda())
        assert torch.equal(y, y2)
    with pytest.raises(RuntimeError):
        Audio2Mel().cuda()(np.zeros(100, np.float32))   # shorter than one frame


# ---------------------------------------------------------------- hot-path layer shapes vs oracle

HOT_CONVS = [
    # name, B, Cin, L, Cout, K, stride, pad, dil, groups, act, reflect
    ("g_first_k7", 2, 80, 32, 512, 7, 1, 3, 1, 1, 1, True),
    ("atom_c256_d9", 2, 256, 70, 256, 3, 1, 9, 9, 1, 1, False),
    ("atom_c128_d3", 1, 128, 300, 128, 3, 1, 3, 3, 1, 1, False),
    ("atom_c64_d1", 1, 64, 515, 64, 3, 1, 1, 1, 1, 1, False),
    ("atom_c32_d9", 2, 32, 1031, 32, 3, 1, 9, 9, 1, 1, False),
    ("g_last_k7_tanh", 2, 32, 1000, 1, 7, 1, 3, 1, 1, 2, False),
    ("d_k15", 2, 1, 2049, 16, 15, 1, 7, 1, 1, 1, False),
    # one-channel-side stream kernels: 16-byte and scalar paths, chunk (1024) and tile (256) tails
    ("g_last_k7_tanh_odd", 3, 32, 1301, 1, 7, 1, 3, 1, 1, 2, False),
    ("g_last_k7_tanh_3chunks", 1, 32, 2308, 1, 7, 1, 3, 1, 1, 2, False),
    ("d_k15_aligned", 3, 1, 2308, 16, 15, 1, 7, 1, 1, 1, False),
    ("d_k15_short", 2, 1, 9, 16, 15, 1, 7, 1, 1, 1, False),
    ("d_k41_g4", 2, 16, 2049, 64, 41, 4, 20, 1, 4, 1, False),
    ("d_k41_g4_long", 3, 16, 3000, 64, 41, 4, 20, 1, 4, 1, False),
    ("d_k41_g16", 2, 64, 513, 256, 41, 4, 20, 1, 16, 1, False),
    ("d_k41_g64", 1, 256, 129, 1024, 41, 4, 20, 1, 64, 1, False),
    ("d_k41_g256", 2, 1024, 33, 1024, 41, 4, 20, 1, 256, 1, False),
    # split-bf16 grouped kernels (gconv_split.hip): 16-byte paths with whole-row / multi-segment units, segments of
    # 64 / 32 / 16 outputs (1 / 2 / 4 per unit) incl. units that straddle batch rows and a ragged last unit, the
    # 4-outputs-per-group layer, no activation
    ("g3_vec_l2048", 3, 16, 2048, 64, 41, 4, 20, 1, 4, 1, False),
    ("g3_vec_l512_noact", 2, 64, 512, 256, 41, 4, 20, 1, 16, 0, False),
    ("g3_l128_two_rows", 5, 256, 128, 1024, 41, 4, 20, 1, 64, 1, False),
    ("g3_l64_four_rows", 5, 64, 64, 256, 41, 4, 20, 1, 16, 1, False),
    ("g3_l260_wrap", 3, 16, 260, 64, 41, 4, 20, 1, 4, 1, False),
    ("g3_l1025", 2, 16, 1025, 64, 41, 4, 20, 1, 4, 1, False),
    ("g3_og4_l128", 3, 1024, 128, 1024, 41, 4, 20, 1, 256, 1, False),
    ("g3_og4_l65", 2, 512, 65, 512, 41, 4, 20, 1, 128, 1, False),
    ("d_k5", 2, 1024, 17, 1024, 5, 1, 2, 1, 1, 1, False),
    # split-bf16 k5 weight gradient (wgrad_k5.hip: >= 256 channels both sides, rows <= 64): octets from one / several
    # batch rows per step, ragged last octet and last step, aligned and unaligned rows, no activation
    ("w5_l33_b3", 3, 256, 33, 320, 5, 1, 2, 1, 1, 1, False),
    ("w5_l12_noact", 5, 320, 12, 256, 5, 1, 2, 1, 1, 0, False),
    ("w5_l7", 2, 256, 7, 256, 5, 1, 2, 1, 1, 1, False),
    ("w5_l64_b9", 9, 256, 64, 256, 5, 1, 2, 1, 1, 1, False),
    # >= 1000 columns of an odd row length: rows padded to a multiple of 4 for the 16-byte kernels (api.hip pad4)
    ("d_k5_l17_padded", 61, 256, 17, 272, 5, 1, 2, 1, 1, 1, False),
    ("d_k5_l33_padded", 31, 256, 33, 256, 5, 1, 2, 1, 1, 0, False),
    # short-row mode of the pipelined kernel (rows of a length that is not a multiple of 4): partial
    # last tile, partial M tile, 1 / 3 / 14 rows per tile
    ("d_k5_l9_ragged", 17, 256, 9, 328, 5, 1, 2, 1, 1, 1, False),
    ("d_k5_l17_b9", 9, 128, 17, 64, 5, 1, 2, 1, 1, 1, False),
    ("d_k5_l33", 5, 128, 33, 128, 5, 1, 2, 1, 1, 1, False),
    ("d_k5_l101", 3, 128, 101, 64, 5, 1, 2, 1, 1, 1, False),
    # 16-byte aligned rows: the split-bf16 row kernel (conv_rows3.hip) in every tile shape, with M / channel /
    # row tails, packed short rows, dilation halos
    ("r3_c128_l2048", 2, 128, 2048, 128, 3, 1, 1, 1, 1, 1, False),
    ("r3_c256_l256_d9", 3, 256, 256, 256, 3, 1, 9, 9, 1, 1, False),
    ("r3_c64_l4096_d3", 1, 64, 4096, 64, 3, 1, 3, 3, 1, 1, False),
    ("r3_c32_l1024_d9", 3, 32, 1024, 32, 3, 1, 9, 9, 1, 1, False),
    ("r3_m96_ck48_l132", 2, 48, 132, 96, 3, 1, 3, 3, 1, 1, False),
    ("r3_k5_l32", 5, 256, 32, 320, 5, 1, 2, 1, 1, 1, False),
    ("r3_k5_l64_noact", 3, 128, 64, 128, 5, 1, 2, 1, 1, 0, False),
    ("r3_k3_l16_r8", 9, 64, 16, 64, 3, 1, 1, 1, 1, 1, False),
    # ... and rows of any length through its dword loader: the k5 conv at the pooled scales (L = 17 / 9, packed
    # 7 / 14 rows per tile), batch and M tails
    ("r3_k5_l17", 9, 128, 17, 192, 5, 1, 2, 1, 1, 1, False),
    ("r3_k5_l9_m1024", 15, 64, 9, 1024, 5, 1, 2, 1, 1, 1, False),
    ("r3_k3_l131_d9", 2, 64, 131, 64, 3, 1, 9, 9, 1, 1, False),
    ("d_judge", 2, 1024, 9, 1, 3, 1, 1, 1, 1, 0, False),
    ("d_judge_l32", 5, 1024, 32, 1, 3, 1, 1, 1, 1, 0, False),
    ("d_judge_l17", 3, 1024, 17, 1, 3, 1, 1, 1, 1, 0, False),
    ("d_judge_l64_c320", 2, 320, 64, 1, 3, 1, 1, 1, 1, 0, False),
    ("d_judge_l33", 2, 256, 33, 1, 3, 1, 1, 1, 1, 0, False),
    ("d_judge_l1", 4, 512, 1, 1, 3, 1, 1, 1, 1, 0, False),
    # MFMA implicit-GEMM edge cases: M / N / K-dimension tails, the 128x128 tile, no activation
    ("mfma_m48_k120", 3, 40, 77, 48, 3, 1, 3, 3, 1, 1, False),
    ("mfma_m96_k5", 2, 36, 41, 96, 5, 1, 2, 1, 1, 0, False),
    ("mfma_tile128", 4, 128, 12288, 128, 3, 1, 9, 9, 1, 1, False),
    ("mfma_m160_k7_reflect", 2, 24, 50, 160, 7, 1, 3, 1, 1, 2, True),
    # reflection-padded weight gradients: the generator's first conv at the bench batch (wgrad_short.hip: fp32 MFMA, one wave per
    # tile), a ragged length with tanh and a k3 with the pad next to the row length (im2col kernel)
    ("g_first_k7_b32", 32, 80, 32, 512, 7, 1, 3, 1, 1, 1, True),
    ("w_reflect_k7_l37_tanh", 3, 48, 37, 96, 7, 1, 3, 1, 1, 2, True),
    ("w_reflect_k3_l5", 4, 64, 5, 64, 3, 1, 1, 1, 1, 0, True),
    ("w_short_k5_l64_tanh", 8, 36, 64, 96, 5, 1, 2, 1, 1, 2, True),      # wgrad_short.hip: two 32-sample blocks per row, K = 5
    # row-tile weight gradient: short rows packed R per chunk (16-byte and scalar loaders), batch
    # tail, 1x1, K = 7, M / channel tails
    ("w32_d1", 3, 32, 2052, 32, 3, 1, 1, 1, 1, 1, False),
    ("w32_d9_noact", 2, 32, 1024, 32, 3, 1, 9, 9, 1, 0, False),
    ("w32_d3_short", 5, 32, 64, 32, 3, 1, 3, 3, 1, 1, False),
    ("wrows_k5_l32", 5, 128, 32, 192, 5, 1, 2, 1, 1, 1, False),
    ("wrows_k3_l16_r3", 7, 64, 16, 64, 3, 1, 1, 1, 1, 1, False),
    ("wrows_k3_l9_d3", 11, 96, 9, 80, 3, 1, 3, 3, 1, 1, False),
    ("wrows_k1", 2, 160, 200, 136, 1, 1, 0, 1, 1, 0, False),
    ("wrows_k7", 3, 48, 132, 72, 7, 1, 3, 1, 1, 1, False),
    ("wrows_k3_d9_long", 2, 256, 1024, 256, 3, 1, 9, 9, 1, 1, False),
]


@pytest.mark.parametrize("case", HOT_CONVS, ids=[c[0] for c in HOT_CONVS])
def test_conv_hot_shapes_vs_oracle(case):
    from featuresynth._ops import functional as F_
    from oracle import oracle as O
    name, B, Cin, L, Cout, K, stride, pad, dil, groups, act, reflect = case
    rng = np.random.default_rng(stable_seed(name))
    x = rng.standard_normal((B, Cin, L)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin // groups, K)) * 0.1).astype(np.float32)
    b = (rng.standard_normal((Cout,)) * 0.1).astype(np.float32)
    pm = O.PAD_REFLECT if reflect else O.PAD_ZERO
    y_ref = O.conv1d_fwd(x, w, b, stride, pad, dil, groups, pm, act)
    gy = rng.standard_normal(y_ref.shape).astype(np.float32)
    xt, wt, bt = dev(x).requires_grad_(not reflect), dev(w).requires_grad_(True), dev(b).requires_grad_(True)
    y = F_.Conv1dFn.apply(xt, wt, bt, stride, pad, dil, groups, 1 if reflect else 0, act)
    assert rel_l2(host(y), y_ref) < FWD_TOL
    # the LeakyReLU mask is taken from the device's own activations: an output within rounding of
    # zero may carry the other sign in the double-precision oracle, which would flip its slope
    gp = O.act_bwd(host(y), gy, act)
    gw_ref, gb_ref = O.conv1d_bwd_weight(x, gp, w.shape, stride, pad, dil, groups, pm)
    if reflect:
        gw, gb = torch.autograd.grad(y, (wt, bt), dev(gy))
    else:
        gx, gw, gb = torch.autograd.grad(y, (xt, wt, bt), dev(gy))
        gx_ref = O.conv1d_bwd_data(gp, w, x.shape, stride, pad, dil, groups, pm)
        assert rel_l2(host(gx), gx_ref) < GRAD_TOL
    assert rel_l2(host(gw), gw_ref) < GRAD_TOL
    assert rel_l2(host(gb), gb_ref) < GRAD_TOL


HOT_CONVT = [("ct_512_256", 2, 512, 9, 256, 16, 8, 4),
             # input lengths that are multiples of 32: the split-bf16 stride-8 weight gradient (wgrad_convt.hip), one and
             # several steps per row, a ragged last slab
             # one output channel, stride 2 (stage-1 generator's last layer as lines): stream kernels, aligned / ragged rows
             ("ct_thin_c96_l256", 5, 96, 256, 1, 4, 2, 1), ("ct_thin_c40_l131", 3, 40, 131, 1, 4, 2, 1),
             ("ct_thin_c8_l600", 2, 8, 600, 1, 4, 2, 1),
             ("ct_w8_l32", 3, 128, 32, 48, 16, 8, 4), ("ct_w8_l96", 5, 128, 96, 16, 16, 8, 4), ("ct_256_128", 1, 256, 70, 128, 16, 8, 4),
             ("ct_128_64", 2, 128, 130, 64, 4, 2, 1), ("ct_64_32", 1, 64, 1027, 32, 4, 2, 1),
             # lengths the pipelined kernels take (L % 4 == 0): one-chunk rows, chunk tails, batch tails
             ("ct_512_256_l32", 3, 512, 32, 256, 16, 8, 4), ("ct_256_128_l132", 1, 256, 132, 128, 16, 8, 4),
             ("ct_128_64_l192", 2, 128, 192, 64, 4, 2, 1), ("ct_64_32_l1028", 1, 64, 1028, 32, 4, 2, 1)]


@pytest.mark.parametrize("case", HOT_CONVT, ids=[c[0] for c in HOT_CONVT])
def test_convt_hot_shapes_vs_oracle(case):
    from featuresynth._ops import functional as F_
    from oracle import oracle as O
    name, B, Cin, L, Cout, K, stride, pad = case
    rng = np.random.default_rng(stable_seed(name))
    x = rng.standard_normal((B, Cin, L)).astype(np.float32)
    w = (rng.standard_normal((Cin, Cout, K)) * 0.1).astype(np.float32)
    b = (rng.standard_normal((Cout,)) * 0.1).astype(np.float32)
    y_ref = O.conv_transpose1d_fwd(x, w, b, stride, pad, O.ACT_LRELU)
    gy = rng.standard_normal(y_ref.shape).astype(np.float32)
    xt, wt, bt = dev(x).requires_grad_(True), dev(w).requires_grad_(True), dev(b).requires_grad_(True)
    y = F_.ConvTranspose1dFn.apply(xt, wt, bt, stride, pad, 1)
    assert rel_l2(host(y), y_ref) < FWD_TOL
    gp = O.act_bwd(host(y), gy, O.ACT_LRELU)      # mask from the device's activations (see above)
    gx, gw, gb = torch.autograd.grad(y, (xt, wt, bt), dev(gy))
    assert rel_l2(host(gx), O.conv_transpose1d_bwd_data(gp, w, x.shape, stride, pad)) < GRAD_TOL
    gw_ref, gb_ref = O.conv_transpose1d_bwd_weight(x, gp, w.shape, stride, pad)
    assert rel_l2(host(gw), gw_ref) < GRAD_TOL
    assert rel_l2(host(gb), gb_ref) < GRAD_TOL


@pytest.mark.parametrize("shape", [(8, 64, 4100, 32, 4, 2), (16, 256, 260, 128, 16, 8), (32, 512, 36, 256, 16, 8),
                                   (32, 128, 516, 64, 4, 2)],
                         ids=["s2_m64_tail", "s8_m1024_tail", "s8_packed_rows", "s2_m128"])
@pytest.mark.parametrize("in_act", [0, 1])
def test_convt_paired_split_kernel(shape, in_act):
    """ConvTranspose1d forward on the paired split-bf16 kernel (conv_rows3.hip, two-tap form: needs >= 128 workgroups,
    hence the larger batches) against a float64 reference: column-tile tails, rows packed 3 per tile, both strides,
    64- and 128-row workgroups, with and without the LeakyReLU in front (generator/full.py: LeakyReLU -> ConvTranspose1d)."""
    import torch.nn.functional as TF
    from featuresynth._ops import prims as P
    B, Cin, Lin, Cout, K, S = shape
    g = torch.Generator(device="cuda").manual_seed(stable_seed("convt3%s%d" % (shape, in_act)) % (1 << 31))
    x = torch.randn(B, Cin, Lin, device="cuda", generator=g)
    w = torch.randn(Cin, Cout, K, device="cuda", generator=g) * 0.05
    b = torch.randn(Cout, device="cuda", generator=g)
    d, lo = P.convt_desc(x.shape, w.shape, S, S // 2, act=1, in_act=in_act)
    assert "k_conv_rows3p" in P.L.load().ms_convt1d_kernel_name(d, 0).decode()
    y = P.convt1d_fwd(x, w, b, d, lo)
    y = y[0] if isinstance(y, tuple) else y
    xin = TF.leaky_relu(x.double(), 0.2) if in_act else x.double()
    ref = TF.leaky_relu(TF.conv_transpose1d(xin, w.double(), b.double(), stride=S, padding=S // 2), 0.2)
    assert tuple(y.shape) == tuple(ref.shape)
    assert float((y.double() - ref).norm() / ref.norm()) < 1e-6


@pytest.mark.parametrize("shape", [(1, 256, 256, 128, 8), (1, 128, 2048, 64, 2), (1, 64, 4096, 32, 2), (2, 256, 100, 128, 8),
                                   (1, 48, 70, 16, 2), (3, 20, 65, 6, 8), (1, 512, 32, 256, 8), (2, 512, 4, 256, 8), (1, 64, 1, 32, 2)],
                         ids=["cfg2_convt2", "cfg2_convt3", "cfg2_convt4", "tile_tail_B2", "s2_partial_round_and_channel_group",
                              "s8_partial_round_and_channel_group", "cfg2_convt1", "smoke_T4", "one_position"])
@pytest.mark.parametrize("in_act", [0, 1])
def test_convt_small_batch_weight_stream(shape, in_act):
    """ConvTranspose1d (kernel 2 S / stride S / padding S / 2) + LeakyReLU at INFERENCE batch sizes -- BASELINE config 2's
    four upsampling layers at B = 1 (generator/full.py:27-40) -- on the fp32 vector-FMA kernel of csrc/small_rows.hip (16
    positions x 64 / S output channels per workgroup, lanes own (output channel, phase) pairs) against the same op in float64: the
    four config-2 shapes, a tile tail, input-channel counts that leave a partial LDS round, output-channel counts that leave a
    partial channel group, rows shorter than a tile, with and without the activation in front; every output element written;
    launch-to-launch determinism.  fp32 products and sums: 1e-6."""
    import torch.nn.functional as TF
    from featuresynth._ops import prims as P
    B, Cin, Lin, Cout, S = shape
    g = torch.Generator(device="cuda").manual_seed(stable_seed("convtsmall%s%d" % (shape, in_act)) % (1 << 31))
    x = torch.randn(B, Cin, Lin, device="cuda", generator=g)
    w = torch.randn(Cin, Cout, 2 * S, device="cuda", generator=g) * 0.05
    b = torch.randn(Cout, device="cuda", generator=g)
    d, lo = P.convt_desc(x.shape, w.shape, S, S // 2, act=1, in_act=in_act)
    assert "k_convt_lanes" in P.L.load().ms_convt1d_kernel_name(d, 0).decode()
    y = torch.full((B, Cout, lo), float("nan"), device="cuda")            # every output element must be written
    P.convt1d_fwd(x, w, b, d, lo, out=y)
    xin = TF.leaky_relu(x.double(), 0.2) if in_act else x.double()
    ref = TF.leaky_relu(TF.conv_transpose1d(xin, w.double(), b.double(), stride=S, padding=S // 2), 0.2)
    assert tuple(y.shape) == tuple(ref.shape)
    assert float((y.double() - ref).norm() / ref.norm()) < 1e-6
    y2 = torch.empty_like(y)
    P.convt1d_fwd(x, w, b, d, lo, out=y2)
    assert torch.equal(y, y2), "launch-to-launch determinism"


@pytest.mark.parametrize("shape", [(1, 80, 32, 512, 3, 1), (1, 80, 4, 512, 3, 1), (2, 80, 50, 512, 3, 1), (1, 128, 100, 128, 3, 0),
                                   (4, 16, 7, 132, 0, 0)],
                         ids=["cfg2_first_conv", "smoke_T4", "tile_tail_B2", "zero_pad_4_tiles", "valid_conv_one_column"])
def test_conv_small_batch_weight_stream(shape):
    """Conv1d k7 behind ReflectionPad1d(3) (or zero padding) + LeakyReLU at inference batch sizes -- the generator's first layer
    in BASELINE config 2 (generator/full.py:23-25) -- on the fp32 weight-stream kernel (csrc/small_rows.hip) against float64:
    reflection at rows shorter than a tile (4 positions: every tap of the edge windows is mirrored), a tile tail, zero padding,
    no padding."""
    import torch.nn.functional as TF
    from featuresynth._ops import prims as P
    B, Cin, Lin, Cout, pad, reflect = shape
    g = torch.Generator(device="cuda").manual_seed(stable_seed("convsmall%s" % (shape,)) % (1 << 31))
    x = torch.randn(B, Cin, Lin, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, 7, device="cuda", generator=g) * 0.05
    b = torch.randn(Cout, device="cuda", generator=g)
    d, lo = P.conv_desc(x.shape, w.shape, pad=pad, pad_mode=P.L.PAD_REFLECT if reflect else P.L.PAD_ZERO, act=P.L.ACT_LRELU)
    assert "k_conv_small" in P.L.load().ms_conv1d_kernel_name(d, 0).decode()
    y = torch.full((B, Cout, lo), float("nan"), device="cuda")
    P.conv1d_fwd(x, w, b, d, lo, out=y)
    xin = TF.pad(x.double(), (pad, pad), mode="reflect") if reflect else x.double()
    ref = TF.leaky_relu(TF.conv1d(xin, w.double(), b.double(), padding=0 if reflect else pad), 0.2)
    assert tuple(y.shape) == tuple(ref.shape)
    assert float((y.double() - ref).norm() / ref.norm()) < 1e-6


def test_bad_arguments_raise():
    from featuresynth._ops import functional as F_
    x = torch.zeros(1, 4, 16, device="cuda")
    w = torch.zeros(4, 3, 3, device="cuda")          # channel mismatch
    with pytest.raises(RuntimeError):
        F_.Conv1dFn.apply(x, w, None, 1, 1, 1, 1, 0, 0)
    with pytest.raises(RuntimeError):                  # CPU tensor: no fallback
        F_.Conv1dFn.apply(x.cpu(), torch.zeros(4, 4, 3), None, 1, 1, 1, 1, 0, 0)
    with pytest.raises(RuntimeError):                  # kernel longer than the padded input
        F_.Conv1dFn.apply(x, torch.zeros(4, 4, 41, device="cuda"), None, 1, 0, 1, 1, 0, 0)
    with pytest.raises(RuntimeError):                  # non-contiguous
        F_.Conv1dFn.apply(torch.zeros(1, 16, 4, device="cuda").transpose(1, 2),
                          torch.zeros(4, 4, 3, device="cuda"), None, 1, 1, 1, 1, 0, 0)


def test_audio2mel_dataset_scale():
    """the reference runs Audio2Mel over 30-second chunks
    (feature/feature.py:80-85); a batch of such chunks on the device vs the oracle, plus frame
    independence (a chunk's frames do not depend on the batch it sits in)."""
    from featuresynth.feature import Audio2Mel
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    n = 22050 * 30
    x = (rng.standard_normal((3, 1, n)) * 0.1).astype(np.float32)
    a2m = Audio2Mel().cuda()
    y = a2m(dev(x))
    ref = O.audio2mel(x[1, 0], n_mel=80)
    assert tuple(y.shape) == (3,) + tuple(ref.shape[1:]) and ref.shape[2] > 2500
    assert rel_l2(host(y[1:2]), ref) < 1e-4
    y1 = a2m(dev(x[1:2]))
    assert torch.equal(y1, y[1:2])


@pytest.mark.parametrize("shape", [(4, 128, 512, 3, 3), (3, 256, 64, 3, 9), (5, 1024, 32, 5, 1), (2, 64, 1028, 3, 1)],
                         ids=["c128", "c256_d9", "k5_l32", "c64_tail"])
def test_kernel_generations_agree(shape, monkeypatch):
    """The pipelined kernels (conv_rows2.hip, wgrad_rows.hip) against the first-generation ones they
    replace (MSYNTH_ROWS2=0 / MSYNTH_WROWS=0): forward and backward-data accumulate in the same order
    (bit-identical), the weight gradient differs only in split-K grouping."""
    from featuresynth._ops import prims as P
    B, C, Lg, K, dil = shape
    rng = np.random.default_rng(11)
    x = dev(rng.standard_normal((B, C, Lg)).astype(np.float32))
    w = dev((rng.standard_normal((C, C, K)) * 0.05).astype(np.float32))
    b = dev(rng.standard_normal((C,)).astype(np.float32))
    res = dev(rng.standard_normal((B, C, Lg)).astype(np.float32))
    gy = dev(rng.standard_normal((B, C, Lg)).astype(np.float32))
    d, lo = P.conv_desc(x.shape, w.shape, pad=dil * (K - 1) // 2, dil=dil, act=1)
    out = {}
    monkeypatch.setenv("MSYNTH_ROWS3", "0")
    for gen in ("0", "1", "3"):
        monkeypatch.setenv("MSYNTH_ROWS2", "1" if gen == "3" else gen)
        monkeypatch.setenv("MSYNTH_WROWS", "1" if gen == "3" else gen)
        monkeypatch.setenv("MSYNTH_ROWS3", "1" if gen == "3" else "0")
        y, ya = P.conv1d_fwd(x, w, b, d, lo, residual=res, want_y_act=True)
        gx = P.conv1d_bwd_data(gy, out["1"][1] if gen == "3" else ya, w, d, gx_add=res)
        gw, gb = P.conv1d_bwd_weight(x, gy, ya, d, w.shape)
        out[gen] = [host(t) for t in (y, ya, gx, gw, gb)] if gen != "3" else [y, ya, gx]
        if gen == "1":
            out["1"][1] = ya                       # (device tensor: the third generation's backward uses this mask)
    for i in (0, 2):
        assert np.array_equal(out["0"][i], out["1"][i])
    assert np.array_equal(out["0"][1], host(out["1"][1]))
    assert rel_l2(out["1"][3], out["0"][3]) < 2e-6 and rel_l2(out["1"][4], out["0"][4]) < 2e-6
    # third generation (conv_rows3.hip): fp32 operands split exactly into three bf16 pieces, six partial
    # products on the bf16 matrix pipe -- not the same rounding sequence as the fmaf chain, same accuracy
    y3, ya3, gx3 = out["3"]
    assert rel_l2(host(y3), out["1"][0]) < 1e-6 and rel_l2(host(ya3), host(out["1"][1])) < 1e-6
    assert rel_l2(host(gx3), out["1"][2]) < 1e-6


@pytest.mark.parametrize("shape", [(3, 16, 2048, 64, 4), (2, 64, 513, 256, 16), (5, 256, 128, 1024, 64), (3, 1024, 36, 1024, 256)],
                         ids=["g4_vec", "g16_l513", "g64_two_rows", "g256_og4"])
def test_grouped_kernel_generations_agree(shape, monkeypatch):
    """Split-bf16 grouped kernels (gconv_split.hip) against the fp32-MFMA ones they replace (MSYNTH_GCONV3=0):
    forward, backward data with a residual gradient added, weight gradient accumulated into existing values."""
    from featuresynth._ops import prims as P
    B, Cin, Lin, Cout, groups = shape
    rng = np.random.default_rng(23)
    x = dev(rng.standard_normal((B, Cin, Lin)).astype(np.float32))
    w = dev((rng.standard_normal((Cout, 4, 41)) * 0.05).astype(np.float32))
    b = dev(rng.standard_normal((Cout,)).astype(np.float32))
    d, lo = P.conv_desc(x.shape, w.shape, stride=4, pad=20, groups=groups, act=1)
    gy = dev(rng.standard_normal((B, Cout, lo)).astype(np.float32))
    res = dev(rng.standard_normal((B, Cin, Lin)).astype(np.float32))
    gw0 = dev(rng.standard_normal((Cout, 4, 41)).astype(np.float32))
    gb0 = dev(rng.standard_normal((Cout,)).astype(np.float32))
    lib = P.L.load()
    out = {}
    og4 = Cout // groups == 4          # the 4 x 4 layer has a vector-pipe fp32 generation of its own (csrc/gconv4.hip)
    for gen in ("0", "1") + (("4",) if og4 else ()):
        monkeypatch.setenv("MSYNTH_GCONV3", "1" if gen == "4" else gen)
        monkeypatch.setenv("MSYNTH_G4", "1" if gen == "4" else "0")
        names = [lib.ms_conv1d_kernel_name(d, k).decode() for k in range(3)]
        if gen == "4":
            assert names == ["k_g4_fwd", "k_g4_bwd_data", "k_g4_wgrad"], names
        else:
            assert all(("split" in nm) == (gen == "1") or (k == 1 and Cout // groups != 16) for k, nm in enumerate(names)), names
        y, _ = P.conv1d_fwd(x, w, b, d, lo)
        ya = y if gen == "0" else out["0"][0]                 # one activation mask for both generations
        gx = P.conv1d_bwd_data(gy, ya, w, d, gx_add=res)
        gw, gb = P.conv1d_bwd_weight(x, gy, ya, d, w.shape, gw=gw0.clone(), gb=gb0.clone(), accumulate=True)
        out[gen] = (y, gx, gw, gb)
    for a, c in zip(out["1"], out["0"]):
        assert rel_l2(host(a), host(c)) < 2e-6     # (r04: block-scaled fp16 x 2 / three products against fp32 MFMA: fp32 summation-order level)
    if og4:
        for a, c in zip(out["4"], out["0"]):
            assert rel_l2(host(a), host(c)) < 2e-6


@pytest.mark.parametrize("case", [
    # name, B, C, L, dilations (one job per entry), accumulate pattern
    ("stack_c128", 3, 128, 512, (1, 9, 1, 3, 1, 1), (False, True, False, False, True, False)),
    ("stack_c64", 2, 64, 1024, (1, 9, 1, 3, 1, 1), (False,) * 6),
    ("stack_c256_short", 5, 256, 64, (9, 1, 3), (True, False, True)),
    ("pair_c192_ragged_m", 2, 192, 256, (3, 1), (False, False)),
    ("stack_c32", 2, 32, 512, (1, 3, 9, 1), (False, True, False, False)),          # batched per-wave kernel
    ("stack_c32_full", 3, 32, 2048, (1, 9, 1, 3, 1, 1), (False,) * 6),
    ("fallback_c32_odd", 2, 32, 130, (1, 3), (False, False)),                      # entry by entry
    ("fallback_odd_len", 2, 128, 130, (1, 3), (False, False)),                     # unaligned rows
], ids=lambda c: c[0])
def test_wgrad_multi_matches_single_calls(case):
    """ms_conv1d_bwd_weight_multi == the same jobs through ms_conv1d_bwd_weight one by one (the batched
    launch only regroups the split-K slices), including accumulation into existing gradients."""
    from featuresynth._ops import prims as P
    name, B, C, Lg, dils, accs = case
    rng = np.random.default_rng(len(name) + C)
    jobs, ref = [], []
    for dil, acc in zip(dils, accs):
        x = dev(rng.standard_normal((B, C, Lg)).astype(np.float32))
        gy = dev(rng.standard_normal((B, C, Lg)).astype(np.float32))
        ya = dev(rng.standard_normal((B, C, Lg)).astype(np.float32))
        d, _ = P.conv_desc(x.shape, (C, C, 3), pad=dil, dil=dil, act=1)
        gw0 = rng.standard_normal((C, C, 3)).astype(np.float32)
        gb0 = rng.standard_normal((C,)).astype(np.float32)
        gw_a, gb_a = (dev(gw0), dev(gb0)) if acc else (None, None)
        gw_b, gb_b = (dev(gw0), dev(gb0)) if acc else (None, None)
        jobs.append((x, gy, ya, d, (C, C, 3), gw_a, gb_a, acc))
        ref.append(P.conv1d_bwd_weight(x, gy, ya, d, (C, C, 3), gw_b, gb_b, acc))
    got = P.conv1d_bwd_weight_multi(jobs)
    assert len(got) == len(ref)
    for (gw, gb), (rw, rb) in zip(got, ref):
        assert rel_l2(host(gw), host(rw)) < 2e-6
        assert rel_l2(host(gb), host(rb)) < 2e-6


def test_judge_loss_multi_matches_single_terms():
    """The hinge-D / negative-mean terms of all scales in one launch == the sum of the per-scale kernels,
    forward value and gradients (loss/loss.py:9-25 on judgements of 32 / 17 / 9 frames)."""
    from featuresynth._ops import lib as L_, prims as P
    rng = np.random.default_rng(3)
    rs = [dev(rng.standard_normal((8, 1, n)).astype(np.float32) * 2) for n in (32, 17, 9)]
    fs = [dev(rng.standard_normal((8, 1, n)).astype(np.float32) * 2) for n in (32, 17, 9)]
    g = dev(np.array(0.7, dtype=np.float32))
    for kind, use_r in ((L_.JUDGE_HINGE_D, True), (L_.JUDGE_NEG_MEAN, False)):
        out = torch.empty((), dtype=torch.float32, device="cuda")
        P.judge_loss_multi_fwd(kind, rs if use_r else None, fs, out)
        ref = sum(float(P.hinge_d_fwd(r, f)) if use_r else float(P.neg_mean_fwd(f)) for r, f in zip(rs, fs))
        assert abs(float(out) - ref) <= 1e-6 * max(1.0, abs(ref))
        grs = [torch.empty_like(r) for r in rs] if use_r else None
        gfs = [torch.empty_like(f) for f in fs]
        P.judge_loss_multi_bwd(kind, rs if use_r else None, fs, g, 1.0, grs, gfs)
        for i in range(3):
            if use_r:
                er, ef = P.hinge_d_bwd(rs[i], fs[i], g)
                assert np.array_equal(host(grs[i]), host(er)) and np.array_equal(host(gfs[i]), host(ef))
            else:
                assert np.array_equal(host(gfs[i]), host(P.neg_mean_bwd(fs[i], g)))


@pytest.mark.parametrize("rates", [(44100, 22050), (48000, 22050), (11025, 22050)], ids=["down2", "down_147_320", "up2"])
def test_audio_frontend_vs_oracle(rates):
    """feature.audio_from_samples (HIP sinc resampler + peak normalisation: the arithmetic of the reference's
    audio(), feature/feature.py:64-71) against the numpy restatement of resampy 'kaiser_best' (parity unpinned:
    librosa / resampy are absent), plus Audio2Mel on the result (the reference's spectrogram(), :78-85)."""
    from featuresynth.feature import Audio2Mel, audio_from_samples, resample
    from oracle import oracle as O
    orig, target = rates
    rng = np.random.default_rng(stable_seed("audio%s" % (rates,)))
    n = 6000
    t = np.arange(n) / orig
    x = np.stack([0.4 * np.sin(2 * np.pi * 220 * t) + 0.1 * rng.standard_normal(n),
                  0.05 * rng.standard_normal(n), np.zeros(n)]).astype(np.float32)
    y = resample(dev(x), orig, target)
    ref = O.resample_kaiser_best(x, orig, target)
    assert tuple(y.shape) == ref.shape
    assert np.abs(host(y) - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
    a = audio_from_samples(dev(x).view(3, 1, n), orig, target)
    ra = O.audio_from_samples(x, orig, target)
    assert tuple(a.shape) == (3, 1, ref.shape[-1])
    assert np.abs(host(a)[:, 0] - ra).max() < 5e-5
    assert abs(float(a[0].abs().max()) - 0.95) < 1e-5 and float(a[2].abs().max()) == 0.0
    if a.shape[-1] >= 1024 and target == 22050:
        mel = Audio2Mel(n_mel_channels=128).cuda()(a[:1].contiguous())
        assert mel.shape[1] == 128 and torch.isfinite(mel).all()


def test_profile_mode_reports_device_time():
    """ms_profile_kernels / ms_profile_take (bench.py's roofline durations): in profile mode every C-ABI call reports the
    device time of the kernels it launched; the result of the call is unchanged and the mode switches off cleanly."""
    from featuresynth._ops import lib as L
    from featuresynth._ops import prims as P
    x = dev(np.random.default_rng(0).standard_normal((8, 64, 512)))
    w = dev(np.random.default_rng(1).standard_normal((64, 64, 3)) * 0.05)
    d, lo = P.conv_desc(x.shape, w.shape, pad=1, act=L.ACT_LRELU)
    ref, _ = P.conv1d_fwd(x, w, None, d, lo)
    L.profile_begin()
    y, _ = P.conv1d_fwd(x, w, None, d, lo)
    rec = L.profile_end()
    assert len(rec) == 1 and rec[0][1]["kernels"] >= 1
    # the record names what was dispatched and the arithmetic it ran (products per fp32 multiply: 0 / 3 / 6)
    assert rec[0][1]["kernel"].startswith("k_conv_") and rec[0][1]["products"] in (0, 3, 6)
    assert 0.5e-3 < rec[0][2] < 5.0 and rec[0][2] <= rec[0][1]["event_ms"] * 1.05       # milliseconds; device <= event reading
    assert torch.equal(y, ref)
    P.conv1d_fwd(x, w, None, d, lo)
    after = L.ProfileRecord()
    assert L.load().ms_profile_take(ctypes.byref(after)) == 0
    assert after.kernels == 0 and after.kernel == b"" and after.device_us == 0.0     # session closed: launches record nothing
    assert L.load().ms_profile_take(None) < 0                                        # the record is an out-parameter
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):                            # and launches are capturable again
        y2, _ = P.conv1d_fwd(x, w, None, d, lo)
    g.replay(); torch.cuda.synchronize()
    assert torch.equal(y2, ref)


def test_side_stream_tracking_under_capture():
    """graph.join_side_streams / begin_step (the r04 audit of the forked-replay fault, DESIGN_HISTORY.md section 6b): forks are tracked per
    trainer step -- begin_step() drops what earlier steps left -- and under hipGraph capture a stream that is NOT part of the
    capture is neither waited for (recording an event on a non-capturing stream and waiting for it from the capturing one
    is not a captured dependency: it invalidates the capture) nor left in the set; a fork that IS part of the capture is joined."""
    from featuresynth._ops import graph as G
    device = torch.device("cuda", torch.cuda.current_device())
    stale, side, cap = (torch.cuda.Stream(device=device) for _ in range(3))
    x = torch.ones(1024, device=device)
    torch.cuda.synchronize()
    G.begin_step()
    assert not G._FORKED_SINCE_JOIN
    stale.wait_stream(torch.cuda.current_stream(device))
    with G.forked(stale):                                  # an eager step forks to `stale` and never joins explicitly
        w = x * 3
    assert stale in G._FORKED_SINCE_JOIN
    torch.cuda.current_stream(device).wait_stream(stale)
    G.begin_step()                                         # the next step starts clean
    assert not G._FORKED_SINCE_JOIN
    G._FORKED_SINCE_JOIN.add(stale)                        # ... and if a stale stream were still tracked during a capture:
    g = torch.cuda.CUDAGraph()
    cap.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(cap):
        g.capture_begin()
        side.wait_stream(cap)
        with G.forked(side):                               # a fork that belongs to this capture
            y = x * 2
        assert side in G._FORKED_SINCE_JOIN
        G.join_side_streams(device)
        z = y + 1
        g.capture_end()
    assert stale not in G._FORKED_SINCE_JOIN and side not in G._FORKED_SINCE_JOIN
    torch.cuda.current_stream(device).wait_stream(cap)
    g.replay()
    torch.cuda.synchronize()
    assert float(z.min()) == 3.0 and float(z.max()) == 3.0 and float(w.max()) == 3.0
    G.begin_step()


@pytest.mark.parametrize("n", [1, 255, 4096 + 3, 1 << 20])
def test_elementwise_entries_without_a_network_test(n):
    """ms_add (out of place and in place) and ms_l1_mean_bwd (overwrite and accumulate): C-ABI entries of the generic autograd
    path (functional.py) that no network-level test dispatches (tools/trace_dispatch.sh) -- bit-exact against numpy: one fp32
    add, and d/df mean|r - f| = sign(f - r) / n times the upstream gradient and the scale (reference loss/loss.py:44-49:
    F.l1_loss on feature maps)."""
    from featuresynth._ops import prims as P
    rng = np.random.default_rng(n)
    a, b = (rng.standard_normal(n).astype(np.float32) for _ in range(2))
    at, bt = dev(a), dev(b)
    assert np.array_equal(host(P.add(at, bt)), a + b)
    ct = at.clone()
    P.add_(ct, bt)
    assert np.array_equal(host(ct), a + b)
    b[::7] = a[::7]                                   # ties: the gradient of |r - f| at 0 is 0 (torch's sign convention)
    bt = dev(b)
    gout = dev(np.array(0.37, np.float32))
    want = (np.sign(b.astype(np.float64) - a) * (0.37 * 2.5 / n)).astype(np.float32)
    got = host(P.l1_mean_bwd(at, bt, gout, scale=2.5))
    assert np.allclose(got, want, rtol=1e-6, atol=0), float(np.abs(got - want).max())
    assert np.array_equal(got[::7], np.zeros_like(got[::7]))
    acc = dev(np.ones(n, np.float32))
    P.l1_mean_bwd(at, bt, gout, scale=2.5, gf=acc)
    assert np.allclose(host(acc), 1.0 + want, rtol=1e-6, atol=0)
