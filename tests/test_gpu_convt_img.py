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
