"""ConvTranspose1d forward on pre-split weight images (csrc/convt_img.hip: ms_convt1d_img_*) -- the generator's two
stride-8 upsampling layers (reference generator/full.py:27-32) -- against the CPU oracle and the row-tile path."""
import numpy as np
import pytest

from conftest import rel_l2, stable_seed

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


CASES = [("l1_b32", 32, 512, 32, 256, 1), ("l1_b40_ragged", 40, 512, 28, 256, 1), ("l2_b8", 8, 256, 256, 128, 1),
         ("l2_b5_ragged_noact", 5, 256, 212, 128, 0)]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_convt_image_kernel_vs_oracle_and_row_tile(case, monkeypatch):
    from featuresynth._ops import lib as L
    from featuresynth._ops import prims as P
    from oracle import oracle as O
    name, B, Cin, Lin, Cout, act = case
    rng = np.random.default_rng(stable_seed(name))
    x = rng.standard_normal((B, Cin, Lin)).astype(np.float32)
    w = (rng.standard_normal((Cin, Cout, 16)) / np.sqrt(2 * Cin)).astype(np.float32)
    b = (rng.standard_normal((Cout,)) * 0.1).astype(np.float32)
    xt, wt, bt = dev(x), dev(w), dev(b)
    d, lo = P.convt_desc(xt.shape, wt.shape, 8, 4, act=act)
    assert P.convt_img_bytes(d) > 0, "the image kernel must take the generator's stride-8 layers at training batch sizes"
    L.profile_begin()
    y = P.convt1d_fwd(xt, wt, bt, d, lo)
    names = [r[0] for r in L.profile_end()]
    assert names == ["ms_convt1d_img_pack", "ms_convt1d_img_fwd"], names
    y_ref = O.conv_transpose1d_fwd(x, w, b, 8, 4, act)
    assert rel_l2(host(y), y_ref) < 1e-5
    # the row-tile path (ms_convt1d_fwd)
    y2 = torch.empty_like(y)
    nws = L.load().ms_convt1d_workspace_bytes(d, 0)
    ws = L.workspace(nws, xt.device)
    L.call("ms_convt1d_fwd", None, d, xt.data_ptr(), wt.data_ptr(), bt.data_ptr(), y2.data_ptr(), L.ptr(ws), nws, L.stream())
    assert rel_l2(host(y), host(y2)) < 1e-6
    monkeypatch.setenv("MSYNTH_CONVTIMG", "0")
    assert P.convt_img_bytes(d) == 0


# (name, B, Cin, Lin, Cout, stride, act): the generator's stride-8 layers and the stage-1 generator's line convolutions
BWD_CASES = [("g1_b32", 32, 512, 32, 256, 8, 1), ("g2_b8", 8, 256, 256, 128, 8, 1), ("g1_b40_noact", 40, 512, 32, 256, 8, 0),
             ("s1_w4", 128, 2048, 4, 512, 2, 1), ("s1_w8_ragged", 130, 1024, 8, 256, 2, 1), ("s1_w16", 64, 512, 16, 128, 2, 1),
             ("s1_w32", 40, 256, 32, 128, 2, 1), ("s1_w64_noact", 16, 256, 64, 64, 2, 0), ("s1_w128_m192", 9, 192, 128, 32, 2, 1)]


@pytest.mark.parametrize("case", BWD_CASES, ids=[c[0] for c in BWD_CASES])
def test_convt_backward_data_image_kernel(case, monkeypatch):
    """csrc/convt_bwd_img.hip against float64 torch autograd (1e-5) and the fp32 row-tile path it replaces (1e-6)."""
    import torch.nn.functional as F
    from featuresynth._ops import lib as L
    from featuresynth._ops import prims as P
    name, B, Cin, Lin, Cout, S, act = case
    rng = np.random.default_rng(stable_seed("bwd" + name))
    K, pad = 2 * S, S // 2
    w = (rng.standard_normal((Cin, Cout, K)) / np.sqrt(2 * Cout)).astype(np.float32)
    gy = rng.standard_normal((B, Cout, Lin * S)).astype(np.float32)
    y = rng.standard_normal((B, Cout, Lin * S)).astype(np.float32)         # the saved output: only its sign matters
    wt, gt, yt = dev(w), dev(gy), dev(y)
    d, lo = P.convt_desc((B, Cin, Lin), wt.shape, S, pad, act=act)
    assert lo == Lin * S
    assert P.convt_bwd_img_bytes(d) > 0, "the image kernel must take this layer"
    L.profile_begin()
    gx = P.convt1d_bwd_data(gt, yt if act else None, wt, d)
    names = [r[0] for r in L.profile_end()]
    assert names == ["ms_convt1d_bwd_img_pack", "ms_convt1d_bwd_img_data"], names
    # float64: gx = conv1d(gy * act'(y), w) with the mirrored geometry
    g64 = torch.from_numpy(gy).double()
    if act:
        g64 = torch.where(torch.from_numpy(y) > 0, g64, 0.2 * g64)
    ref = F.conv1d(g64, torch.from_numpy(w).double(), None, S, pad).numpy()
    assert tuple(gx.shape) == ref.shape == (B, Cin, Lin)
    assert rel_l2(host(gx), ref) < 1e-5
    monkeypatch.setenv("MSYNTH_CONVTBWDIMG", "0")
    assert P.convt_bwd_img_bytes(d) == 0
    gx2 = P.convt1d_bwd_data(gt, yt if act else None, wt, d)
    assert rel_l2(host(gx), host(gx2)) < 1e-6
    # deterministic (split-K slabs are summed in slice order)
    monkeypatch.delenv("MSYNTH_CONVTBWDIMG")
    assert torch.equal(gx, P.convt1d_bwd_data(gt, yt if act else None, wt, d))


def test_convt_backward_data_image_kernel_declines():
    from featuresynth._ops import prims as P
    for shape, wshape, S in [((1, 512, 32), (512, 256, 16), 8),        # B = 1 inference
                             ((32, 128, 2048), (128, 64, 4), 2),       # rows longer than 256 positions
                             ((32, 96, 256), (96, 1, 4), 2),           # one gradient channel (the thin stream kernels)
                             ((32, 512, 24), (512, 256, 16), 8)]:      # not a power of two
        d, _ = P.convt_desc(shape, wshape, S, S // 2, act=1)
        assert P.convt_bwd_img_bytes(d) == 0, (shape, wshape)


# (name, rows, Cin, W, Cout, act): the stage-1 generator's first line convolutions (stride 2, short rows, many channels)
SHORT_CASES = [("w4", 128, 2048, 4, 512, 1), ("w8_ragged", 130, 1024, 8, 256, 1), ("w16", 64, 512, 16, 128, 1),
               ("w16_noact", 40, 256, 16, 128, 0), ("w4_cout64", 200, 256, 4, 64, 1)]


@pytest.mark.parametrize("case", SHORT_CASES, ids=[c[0] for c in SHORT_CASES])
def test_convt_forward_short_rows(case, monkeypatch):
    """csrc/convt_fwd_short.hip (through ms_convt1d_img_*) against float64 torch (1e-5) and the generic row kernels (1e-6)."""
    import torch.nn.functional as F
    from featuresynth._ops import lib as L
    from featuresynth._ops import prims as P
    name, B, Cin, W, Cout, act = case
    rng = np.random.default_rng(stable_seed("short" + name))
    x = rng.standard_normal((B, Cin, W)).astype(np.float32)
    w = (rng.standard_normal((Cin, Cout, 4)) / np.sqrt(2 * Cin)).astype(np.float32)
    b = (rng.standard_normal((Cout,)) * 0.1).astype(np.float32)
    xt, wt, bt = dev(x), dev(w), dev(b)
    d, lo = P.convt_desc(xt.shape, wt.shape, 2, 1, act=act)
    assert P.convt_img_bytes(d) > 0
    L.profile_begin()
    y = P.convt1d_fwd(xt, wt, bt, d, lo)
    rec = L.profile_end()
    assert [r[0] for r in rec] == ["ms_convt1d_img_pack", "ms_convt1d_img_fwd"] and rec[1][1].get("kernel") == "k_convt_fwd_short", rec
    ref = F.conv_transpose1d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), 2, 1)
    if act:
        ref = F.leaky_relu(ref, 0.2)
    assert tuple(y.shape) == tuple(ref.shape)
    assert rel_l2(host(y), ref.numpy()) < 1e-5
    assert torch.equal(y, P.convt1d_fwd(xt, wt, bt, d, lo))            # slabs summed in slice order
    monkeypatch.setenv("MSYNTH_CONVTSHORT", "0")
    assert P.convt_img_bytes(d) == 0
    assert rel_l2(host(y), host(P.convt1d_fwd(xt, wt, bt, d, lo))) < 1e-6
