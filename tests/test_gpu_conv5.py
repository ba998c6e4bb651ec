"""Dense k5 conv on short rows with pre-split weight images (csrc/conv5_img.hip: ms_conv1d_img_*) -- the discriminator's
1024 -> 1024 layer (reference discriminator/full.py:19) at rows of 32 / 17 / 9 samples -- forward and backward data against
the CPU oracle and against the generic ms_conv1d_fwd / ms_conv1d_bwd_data path, over split-K plans (batch sizes), partial
tiles, channel counts that are not powers of two, with and without the fused LeakyReLU / gradient add."""
import numpy as np
import pytest

from conftest import rel_l2, stable_seed

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


# name, B, Cin, Cout, L, act
CASES = [("l32_b3", 3, 256, 320, 32, 1), ("l17_b5", 5, 320, 256, 17, 1), ("l9_b7", 7, 256, 256, 9, 1),
         ("l32_b64_split", 64, 256, 256, 32, 1), ("l17_b33_partial_tile", 33, 256, 256, 17, 1), ("l9_b64", 64, 256, 256, 9, 0),
         ("l20_b4_noact", 4, 256, 256, 20, 0), ("l3_b40", 40, 256, 256, 3, 1), ("l64_b2", 2, 256, 256, 64, 1),
         ("full_l32_b4", 4, 1024, 1024, 32, 1), ("full_l17_b4", 4, 1024, 1024, 17, 1), ("full_l9_b6", 6, 1024, 1024, 9, 1)]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv5_image_kernel_vs_oracle_and_generic(case):
    from featuresynth._ops import lib as L
    from featuresynth._ops import prims as P
    from oracle import oracle as O
    name, B, Cin, Cout, Lg, act = case
    rng = np.random.default_rng(stable_seed(name))
    x = rng.standard_normal((B, Cin, Lg)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 5)) / np.sqrt(5 * Cin)).astype(np.float32)
    b = (rng.standard_normal((Cout,)) * 0.1).astype(np.float32)
    xt, wt, bt = dev(x), dev(w), dev(b)
    d, lo = P.conv_desc(xt.shape, wt.shape, pad=2, act=act)
    assert P.conv_img_bytes(d) > 0, "the image kernel must take this geometry"
    img_f, img_b = P.conv_img_pack(d, wt), P.conv_img_pack(d, wt, backward=True)
    y = P.conv1d_img_fwd(xt, img_f, bt, d, lo)
    y_ref = O.conv1d_fwd(x, w, b, 1, 2, 1, 1, O.PAD_ZERO, act)
    assert rel_l2(host(y), y_ref) < 1e-5
    y2, _ = P.conv1d_fwd(xt, wt, bt, d, lo)
    assert rel_l2(host(y), host(y2)) < 2e-6        # (fp16 x 2 / three products here, bf16 x 3 / six exact products there)
    # backward data, LeakyReLU derivative from the device's own activations, with and without the gradient add
    gy = rng.standard_normal(y_ref.shape).astype(np.float32)
    add = rng.standard_normal(x.shape).astype(np.float32)
    gyt, addt = dev(gy), dev(add)
    ya = y if act else None
    gp = O.act_bwd(host(y), gy, act) if act else gy
    gx_ref = O.conv1d_bwd_data(gp, w, x.shape, 1, 2, 1, 1, O.PAD_ZERO)
    gx = P.conv1d_img_bwd_data(gyt, ya, img_b, d)
    assert rel_l2(host(gx), gx_ref) < 1e-5
    gx_a = P.conv1d_img_bwd_data(gyt, ya, img_b, d, gx_add=addt)
    assert rel_l2(host(gx_a), gx_ref + add) < 1e-5
    gx2 = P.conv1d_bwd_data(gyt, ya, wt, d, gx_add=addt)
    assert rel_l2(host(gx_a), host(gx2)) < 2e-6
    # deterministic (split-K slabs summed in slice order)
    assert torch.equal(P.conv1d_img_fwd(xt, img_f, bt, d, lo), y)
    # block scaling (r04): the relative error does not depend on the magnitude of the data, nor on one channel range being
    # 2^12 louder than the rest (the scale of a row moves between chunks: partial sums are folded under the old scale)
    for scale in (1e-9, 1e4):
        xs = x * scale
        xs[:, Cin // 2:Cin // 2 + 16] *= 4096.0
        ys = P.conv1d_img_fwd(dev(xs), img_f, dev(b * scale), d, lo)
        ys_ref = torch.nn.functional.conv1d(dev(xs).double(), wt.double(), dev(b * scale).double(), padding=2)
        if act:
            ys_ref = torch.nn.functional.leaky_relu(ys_ref, 0.2)
        e = float((ys.double() - ys_ref).norm() / ys_ref.norm())
        assert e < 2e-6, (scale, e)


def test_conv5_image_used_by_the_discriminator_and_switchable(monkeypatch):
    """The discriminator schedules take the image kernel for main.5 at all three scales (one pack per pass and direction);
    MSYNTH_CONV5IMG=0 restores the generic kernels; both give the same judgements / gradients to 1e-6."""
    import featuresynth as fs
    from featuresynth._ops import graph as G
    from featuresynth._ops import lib as L
    from featuresynth._synthetic import module_param_shapes, synthetic_samples, synthetic_state_dict
    d = fs.MelGanDiscriminator()
    d.load_state_dict({k: torch.from_numpy(v) for k, v in
                       synthetic_state_dict(module_param_shapes(d), seed=8, bias_scale=0.02).items()})
    d.cuda()
    params = list(d.parameters())
    x = dev(synthetic_samples(2, 8192))
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MSYNTH_CONV5IMG", mode)
        L.profile_begin()
        with torch.no_grad():
            feats, judges, ctx = G.melgan_forward(x, params, 2)
            gjs = [torch.ones_like(j) for j in judges]
            gx, _ = G.melgan_backward(ctx, params, None, gjs, None, need_gx=True, need_wgrad=False)
        rec = L.profile_end()
        names = [r[0] for r in rec]
        kernels = [r[1].get("kernel", "") for r in rec]          # what the launchers noted (the last kernel of a call)
        if mode == "1":
            # one pack per direction; the layer's three scales travel through ONE parts call each way, which the image
            # kernel serves (part by part at this small batch, as one launch from B = 32 on: tests/test_gpu_parts.py)
            assert names.count("ms_conv1d_img_pack") == 2
            assert sum(k.startswith("k_conv5_img") for k in kernels) == 2, kernels
        else:
            assert not any("img" in n for n in names) and not any(k.startswith("k_conv5_img") for k in kernels)
        out[mode] = ([host(j) for j in judges], host(gx))
    for a, b_ in zip(out["1"][0], out["0"][0]):
        assert rel_l2(a, b_) < 1e-5           # (the judgements are small sums of cancelling terms: measured 1.1e-6)
    assert rel_l2(out["1"][1], out["0"][1]) < 1e-5


def test_image_pair_from_one_maxima_pass():
    """ms_conv1d_img_pack2: the forward and the backward-data image of a layer packed from ONE pass over the weights for their
    common scale -- the kernels give bitwise what they give on the two separately packed images."""
    from featuresynth._ops import lib as L
    from featuresynth._ops import prims as P
    rng = np.random.default_rng(5)
    C, B, Lg = 1024, 32, 32
    x = dev(rng.standard_normal((B, C, Lg)).astype(np.float32))
    w = dev((rng.standard_normal((C, C, 5)) * 0.02).astype(np.float32))
    b = dev(rng.standard_normal(C).astype(np.float32) * 0.1)
    d, lo = P.conv_desc(x.shape, w.shape, pad=2, act=L.ACT_LRELU)
    f1, b1 = P.conv_img_pack(d, w), P.conv_img_pack(d, w, backward=True)
    f2, b2 = P.conv_img_pack2(d, w)
    y1, y2 = P.conv1d_img_fwd(x, f1, b, d, lo), P.conv1d_img_fwd(x, f2, b, d, lo)
    gy = dev(rng.standard_normal(tuple(y1.shape)).astype(np.float32))
    g1, g2 = P.conv1d_img_bwd_data(gy, y1, b1, d), P.conv1d_img_bwd_data(gy, y1, b2, d)
    assert torch.equal(y1, y2) and torch.equal(g1, g2)
