"""Weight gradient of the stride-2 transposed conv on short rows (csrc/wgrad_convt2s.hip: the stage-1 generator's first line
convolutions) against float64 torch autograd and the generic kernels it replaces."""
import numpy as np
import pytest

from conftest import rel_l2, stable_seed

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

# (name, rows, Cin, W, Cout, act)
CASES = [("w4", 128, 2048, 4, 512, 1), ("w8", 256, 1024, 8, 256, 1), ("w16", 64, 512, 16, 128, 1), ("w32_noact", 40, 256, 32, 128, 0),
         ("w4_odd_tiles", 72, 384, 4, 96, 1)]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_short_row_transposed_weight_gradient(case, monkeypatch):
    import torch.nn.functional as F
    from featuresynth._ops import lib as L
    from featuresynth._ops import prims as P
    name, B, Cin, W, Cout, act = case
    rng = np.random.default_rng(stable_seed("wt2s" + name))
    x = rng.standard_normal((B, Cin, W)).astype(np.float32)
    gy = rng.standard_normal((B, Cout, 2 * W)).astype(np.float32)
    y = rng.standard_normal((B, Cout, 2 * W)).astype(np.float32)          # saved output: only its sign matters
    xt, gt, yt = [torch.from_numpy(a).cuda() for a in (x, gy, y)]
    d, lo = P.convt_desc(xt.shape, (Cin, Cout, 4), 2, 1, act=act)
    assert L.load().ms_convt1d_kernel_name(d, 2).decode() == "k_wgrad_convt2_short"
    gw, gb = P.convt1d_bwd_weight(xt, gt, yt if act else None, d, (Cin, Cout, 4))
    # float64: d/dw of sum(conv_transpose1d(x, w) * g'), g' = gy * act'(y)
    g64 = torch.from_numpy(gy).double()
    if act:
        g64 = torch.where(torch.from_numpy(y) > 0, g64, 0.2 * g64)
    w64 = torch.zeros((Cin, Cout, 4), dtype=torch.float64, requires_grad=True)
    (F.conv_transpose1d(torch.from_numpy(x).double(), w64, None, 2, 1) * g64).sum().backward()
    assert rel_l2(gw.cpu().numpy(), w64.grad.numpy()) < 1e-5
    assert rel_l2(gb.cpu().numpy(), g64.sum(dim=(0, 2)).numpy()) < 1e-5
    # accumulate form (beta = 1) and the generic path
    gw2, _ = P.convt1d_bwd_weight(xt, gt, yt if act else None, d, (Cin, Cout, 4), gw=gw.clone(), gb=gb.clone(), accumulate=True)
    assert rel_l2(gw2.cpu().numpy(), 2 * w64.grad.numpy()) < 1e-5
    again, _ = P.convt1d_bwd_weight(xt, gt, yt if act else None, d, (Cin, Cout, 4))
    assert torch.equal(gw, again)                  # slabs are summed in slice order
    monkeypatch.setenv("MSYNTH_WGRADT2S", "0")
    assert L.load().ms_convt1d_kernel_name(d, 2).decode() != "k_wgrad_convt2_short"
    old, _ = P.convt1d_bwd_weight(xt, gt, yt if act else None, d, (Cin, Cout, 4))
    assert rel_l2(gw.cpu().numpy(), old.cpu().numpy()) < 1e-5
