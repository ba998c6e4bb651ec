"""The contract of the block-scaled two-piece fp16 operand scheme (csrc/atom_fused.hip header), per kernel
family that uses it: a block of operand values is scaled by a power of two so that its largest magnitude sits at 2^12 .. 2^15
and every value is split into two fp16 pieces.  What the comments claim, and what is held here PER ELEMENT against float64:

  * every element within 2^16 of its block's largest magnitude keeps 22 significand bits;
  * smaller elements keep an ABSOLUTE error below 2^-37 of that maximum (fp16's subnormal spacing under the block scale).

So with ONE element of a block 2^20 or 2^30 times larger than its O(1) neighbours, every output must stay within
`bound = 2^-34 * (block maximum) * sum |w|  +  sqrt(K) 2^-23 |y|` of the float64 result (the second term is what an fp32
accumulation over K terms costs when one early term dominates: it is the reference's own arithmetic, not the scheme's) -- outputs whose taps meet the outlier as well as those that
only share its block (whose own inputs lose relative precision: that IS the contract; the reference's fp32 has no such
coupling, and real activations have no such range) -- and outputs of OTHER blocks keep the plain fp32-level accuracy.
Blocks: the tile's window (atoms, stride-8 transposed conv), the wave unit (grouped convs), the batch row (k5 layer)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _check(y, yr, block_max, wsum, what, clean=None, K=256, carried=0.0):
    """y: device result, yr: float64 reference (CPU tensors of equal shape); per-element bound relative to the block maximum.
    K: contraction length (fp32 accumulation term); carried: absolute error the operand itself may already carry."""
    err = (y.double().cpu() - yr).abs()
    bound = 2.0 ** -34 * block_max * wsum + np.sqrt(K) * 2.0 ** -23 * yr.abs() + carried + 1e-30
    worst = float((err / bound).max())
    print("%s: worst per-element error / bound = %.3g (max |err| %.3g, block max %.3g)" % (what, worst, float(err.max()), block_max))
    assert worst <= 1.0, (what, worst)
    if clean is not None:       # rows / blocks without the outlier: fp32-level accuracy
        e = float((y.double().cpu()[clean] - yr[clean]).norm() / yr[clean].norm())
        assert e < 2e-6, (what, "clean block", e)


@pytest.mark.parametrize("p2", [20, 30])
@pytest.mark.parametrize("C,Lg", [(64, 1024), (128, 512)])
def test_atom_in_tile_dynamic_range(C, Lg, p2):
    """Fused ResidualAtom forward (util/modules.py:384-388): one input element of batch row 0 is 2^p2, the rest N(0, 1)."""
    import torch.nn.functional as F
    from featuresynth._ops import graph as G
    from featuresynth._ops import prims as P
    rng = np.random.default_rng(C + p2)
    x = rng.standard_normal((2, C, Lg)).astype(np.float32)
    x[0, 3, 200] = 2.0 ** p2
    w0 = (rng.standard_normal((C, C, 3)) / np.sqrt(3 * C)).astype(np.float32)
    w1 = (rng.standard_normal((C, C, 3)) / np.sqrt(3 * C)).astype(np.float32)
    b0 = (rng.standard_normal(C) * 0.1).astype(np.float32)
    b1 = (rng.standard_normal(C) * 0.1).astype(np.float32)
    xt, w0t, w1t, b0t, b1t = (dev(a) for a in (x, w0, w1, b0, b1))
    img = P.atom_image(C, xt.device)
    P.atom_pack([(w0t, w1t, img)])
    y, rec = G.atom_forward(xt, w0t, b0t, w1t, b1t, 3, True, image=img)
    xd = torch.from_numpy(x).double()
    t = F.leaky_relu(F.conv1d(xd, torch.from_numpy(w0).double(), torch.from_numpy(b0).double(), padding=3, dilation=3), 0.2)
    u = F.leaky_relu(F.conv1d(t, torch.from_numpy(w1).double(), torch.from_numpy(b1).double(), padding=1), 0.2)
    ws0, ws1 = float(np.abs(w0).sum(axis=(1, 2)).max()), float(np.abs(w1).sum(axis=(1, 2)).max())
    clean = (slice(1, 2),)                      # batch row 1 never shares a tile with the outlier
    _check(rec[3], t, 2.0 ** p2, ws0, "atom C=%d t" % C, clean, K=3 * C)
    # second GEMM: its operand block is the t tile, whose maximum is ~ |w| 2^p2
    tmax = float(t.abs().max())
    # (the error t may carry from the first GEMM reaches u through conv1: carried)
    _check(y, xd + u, tmax, ws1, "atom C=%d y" % C, clean, K=3 * C, carried=ws1 * 2.0 ** -34 * 2.0 ** p2 * ws0)


@pytest.mark.parametrize("p2", [20, 30])
def test_k5_layer_in_row_dynamic_range(p2):
    """The 1024 -> 1024 k5 layer over the three scales (parts launch, operands pre-split with one scale per batch row) and its
    per-scale launches (scale per row and 16-channel chunk): batch row 0 holds one element of 2^p2."""
    import torch.nn.functional as F
    from featuresynth._ops import lib as L
    from featuresynth._ops import prims as P
    rng = np.random.default_rng(p2)
    C, B = 1024, 32
    xs = [rng.standard_normal((B, C, l)).astype(np.float32) for l in (32, 17, 9)]
    for x in xs:
        x[0, 5, 3] = 2.0 ** p2
    w = (rng.standard_normal((C, C, 5)) / np.sqrt(5 * C)).astype(np.float32)
    b = (rng.standard_normal(C) * 0.1).astype(np.float32)
    xt, wt, bt = [dev(x) for x in xs], dev(w), dev(b)
    d, lo = P.conv_desc(xt[0].shape, wt.shape, pad=2, act=L.ACT_LRELU)
    img = P.conv_img_pack(d, wt)
    ys = P.conv1d_parts_fwd(xt, wt, bt, d, image=img)
    wsum = float(np.abs(w).sum(axis=(1, 2)).max())
    wd, bd = torch.from_numpy(w).double(), torch.from_numpy(b).double()
    for i, x in enumerate(xs):
        yr = F.leaky_relu(F.conv1d(torch.from_numpy(x).double(), wd, bd, padding=2), 0.2)
        _check(ys[i], yr, 2.0 ** p2, wsum, "k5 parts L=%d" % x.shape[2], (slice(1, None),), K=5 * C)
        di, loi = P.conv_desc(xt[i].shape, wt.shape, pad=2, act=L.ACT_LRELU)
        y1 = P.conv1d_img_fwd(xt[i], img, bt, di, loi)
        _check(y1, yr, 2.0 ** p2, wsum, "k5 single L=%d" % x.shape[2], (slice(1, None),), K=5 * C)


@pytest.mark.parametrize("p2", [20, 30])
def test_grouped_layer_in_unit_dynamic_range(p2):
    """Grouped k41 / stride-4 layer (discriminator/full.py:15-18), forward: block = the 64 outputs' input window of one (batch
    row, group); one input element of batch row 0, group 0 is 2^p2."""
    import torch.nn.functional as F
    from featuresynth._ops import lib as L
    from featuresynth._ops import prims as P
    rng = np.random.default_rng(100 + p2)
    B, Cin, Cout, groups, Lin = 2, 64, 256, 16, 2048
    x = rng.standard_normal((B, Cin, Lin)).astype(np.float32)
    x[0, 1, 700] = 2.0 ** p2
    w = (rng.standard_normal((Cout, Cin // groups, 41)) / np.sqrt(41 * 4)).astype(np.float32)
    b = (rng.standard_normal(Cout) * 0.1).astype(np.float32)
    xt, wt, bt = dev(x), dev(w), dev(b)
    d, lo = P.conv_desc(xt.shape, wt.shape, stride=4, pad=20, groups=groups, act=L.ACT_LRELU)
    y, _ = P.conv1d_fwd(xt, wt, bt, d, lo)
    assert L.load().ms_conv1d_kernel_name(d, 0).decode().startswith("k_gconv_split")
    yr = F.leaky_relu(F.conv1d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(),
                               stride=4, padding=20, groups=groups), 0.2)
    _check(y, yr, 2.0 ** p2, float(np.abs(w).sum(axis=(1, 2)).max()), "grouped fwd", (slice(1, None),), K=164)


def test_grouped_wgrad_scale_swing():
    """Weight gradient of the grouped k41 layer: a wave keeps its sums under STICKY power-of-two operand scales and re-expresses
    them when a scale moves.  Batch rows whose operands are 2^90 times smaller than the rows before them (a product swing of
    2^180, beyond fp32's range) must neither overflow the sums nor disturb them: a scale never rises more than 2^30 above the
    smallest one used so far (gconv_split.hip), operands that much smaller are below the sums' resolution anyway."""
    import torch.nn.functional as F
    from featuresynth._ops import lib as L
    from featuresynth._ops import prims as P
    rng = np.random.default_rng(77)
    B, Cin, Cout, groups, Lin = 32, 16, 64, 4, 8192
    x = rng.standard_normal((B, Cin, Lin)).astype(np.float32)
    Lout = (Lin + 2 * 20 - 41) // 4 + 1
    gy = rng.standard_normal((B, Cout, Lout)).astype(np.float32)
    scale = np.where(np.arange(B) < B // 2, 2.0 ** 45, 2.0 ** -45).astype(np.float32)[:, None, None]
    x *= scale; gy *= scale
    xt, gt = dev(x), dev(gy)
    d, lo = P.conv_desc(xt.shape, (Cout, Cin // groups, 41), stride=4, pad=20, groups=groups)
    assert lo == Lout
    assert L.load().ms_conv1d_kernel_name(d, 2).decode().startswith("k_gconv_split")
    gw, gb = P.conv1d_bwd_weight(xt, gt, None, d, (Cout, Cin // groups, 41))
    xd, gd = torch.from_numpy(x).double(), torch.from_numpy(gy).double()
    wd = torch.zeros(Cout, Cin // groups, 41, dtype=torch.float64, requires_grad=True)
    (F.conv1d(xd, wd, None, stride=4, padding=20, groups=groups) * gd).sum().backward()
    assert bool(torch.isfinite(gw).all()) and bool(torch.isfinite(gb).all())
    e = float((gw.double().cpu() - wd.grad).norm() / wd.grad.norm())
    eb = float((gb.double().cpu() - gd.sum(dim=(0, 2))).norm() / gd.sum(dim=(0, 2)).norm())
    print("grouped wgrad under a 2^180 product swing: rel-L2 %.2e (bias %.2e)" % (e, eb))
    assert e < 2e-6 and eb < 2e-6


@pytest.mark.parametrize("p2", [20, 30])
def test_convt8_in_tile_dynamic_range(p2):
    """Stride-8 transposed conv on its weight image (generator/full.py:27-32), forward: block = the tile's input window."""
    import torch.nn.functional as F
    from featuresynth._ops import lib as L
    from featuresynth._ops import prims as P
    rng = np.random.default_rng(200 + p2)
    B, Cin, Cout, Lin = 8, 256, 128, 256            # (B * Lin >= 1024: the geometry the image kernel takes)
    x = rng.standard_normal((B, Cin, Lin)).astype(np.float32)
    x[0, 7, 100] = 2.0 ** p2
    w = (rng.standard_normal((Cin, Cout, 16)) / np.sqrt(2 * Cin)).astype(np.float32)
    b = (rng.standard_normal(Cout) * 0.1).astype(np.float32)
    xt, wt, bt = dev(x), dev(w), dev(b)
    d, lo = P.convt_desc(xt.shape, wt.shape, 8, 4, act=L.ACT_LRELU)
    if not P.convt_img_bytes(d):
        pytest.skip("the image kernel does not take this geometry")
    y = P.convt1d_fwd(xt, wt, bt, d, lo)
    yr = F.leaky_relu(F.conv_transpose1d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(),
                                         stride=8, padding=4), 0.2)
    wsum = float(np.abs(w).sum(axis=(0, 2)).max())
    _check(y, yr, 2.0 ** p2, wsum, "convT8 fwd", (slice(1, None),), K=2 * Cin)


@pytest.mark.parametrize("wexp", [-18, 12])
def test_weights_of_any_magnitude(wexp):
    """The weight images carry their own power-of-two scale (taken from the tensor's largest magnitude by the pack launches; per
    wave group in the grouped kernels): weights 2^12 times larger than the usual O(0.05) -- r04's fixed 64 w overflowed fp16 to
    inf from |w| >= 2^9 -- and 2^18 times smaller keep the fp32-level accuracy.  Inputs are scaled the other way so that the
    outputs stay O(1).  Families: fused atom (forward, backward data), k5 layer (parts launch), grouped conv (forward, backward
    data), stride-8 transposed conv."""
    import torch.nn.functional as F
    from featuresynth._ops import graph as G
    from featuresynth._ops import lib as L
    from featuresynth._ops import prims as P
    ws = 2.0 ** wexp
    rng = np.random.default_rng(100 + wexp)

    def rel(a, b):
        a, b = a.double().cpu().flatten(), b.double().flatten()
        return float((a - b).norm() / b.norm())

    # ---- atom
    C, Lg = 64, 512
    x = (rng.standard_normal((2, C, Lg)) / ws).astype(np.float32)
    w0 = (rng.standard_normal((C, C, 3)) / np.sqrt(3 * C) * ws).astype(np.float32)
    # the second conv scaled the OTHER way: y - x = u and gx - g stay comparable to x and g (no cancellation in the test itself)
    w1 = (rng.standard_normal((C, C, 3)) / np.sqrt(3 * C) / ws).astype(np.float32)
    b0 = (rng.standard_normal(C) * 0.1).astype(np.float32)
    b1 = (rng.standard_normal(C) * 0.1 / ws).astype(np.float32)
    xt, w0t, w1t, b0t, b1t = (dev(a) for a in (x, w0, w1, b0, b1))
    img, imgb = P.atom_image(C, xt.device), P.atom_image(C, xt.device)
    P.atom_pack([(w0t, w1t, img)]); P.atom_pack([(w0t, w1t, imgb)], backward=True)
    y, rec = G.atom_forward(xt, w0t, b0t, w1t, b1t, 3, True, image=img)
    xd = torch.from_numpy(x).double()
    t = F.leaky_relu(F.conv1d(xd, torch.from_numpy(w0).double(), torch.from_numpy(b0).double(), padding=3, dilation=3), 0.2)
    u = F.leaky_relu(F.conv1d(t, torch.from_numpy(w1).double(), torch.from_numpy(b1).double(), padding=1), 0.2)
    assert torch.isfinite(y).all()
    assert rel(rec[3], t) < 2e-6 and rel(y - xt, u) < 4e-6, (rel(rec[3], t), rel(y - xt, u))
    g = dev(rng.standard_normal((2, C, Lg)).astype(np.float32))
    gt, gx, _ = P.atom_bwd_data(g, rec[4], rec[3], imgb, 3)
    gd = g.double().cpu() * torch.where(rec[4].cpu() > 0, 1.0, 0.2).double()
    gt_r = F.conv_transpose1d(gd, torch.from_numpy(w1).double(), padding=1)
    gx_r = F.conv_transpose1d(gt_r * torch.where(rec[3].cpu() > 0, 1.0, 0.2).double(), torch.from_numpy(w0).double(), padding=3, dilation=3)
    assert rel(gt, gt_r) < 2e-6 and rel(gx - g, gx_r) < 4e-6, (rel(gt, gt_r), rel(gx - g, gx_r))

    # ---- k5 layer over the three scales
    Ck, B = 1024, 32
    xs = [(rng.standard_normal((B, Ck, l)) / ws).astype(np.float32) for l in (32, 17, 9)]
    w = (rng.standard_normal((Ck, Ck, 5)) / np.sqrt(5 * Ck) * ws).astype(np.float32)
    b = (rng.standard_normal(Ck) * 0.1).astype(np.float32)
    xst, wt, bt = [dev(a) for a in xs], dev(w), dev(b)
    d, _ = P.conv_desc(xst[0].shape, wt.shape, pad=2, act=L.ACT_LRELU)
    ys = P.conv1d_parts_fwd(xst, wt, bt, d, image=P.conv_img_pack(d, wt))
    for a, yk in zip(xs, ys):
        yr = F.leaky_relu(F.conv1d(torch.from_numpy(a).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=2), 0.2)
        assert torch.isfinite(yk).all() and rel(yk, yr) < 2e-6, rel(yk, yr)

    # ---- grouped conv, forward and backward data
    Bg, Cin, Cout, groups, Lin = 2, 64, 256, 16, 2048
    xg = (rng.standard_normal((Bg, Cin, Lin)) / ws).astype(np.float32)
    wg = (rng.standard_normal((Cout, Cin // groups, 41)) / np.sqrt(164) * ws).astype(np.float32)
    bg = (rng.standard_normal(Cout) * 0.1).astype(np.float32)
    xgt, wgt, bgt = dev(xg), dev(wg), dev(bg)
    dg, lo = P.conv_desc(xgt.shape, wgt.shape, stride=4, pad=20, groups=groups, act=L.ACT_LRELU)
    yg, _ = P.conv1d_fwd(xgt, wgt, bgt, dg, lo)
    pre = F.conv1d(torch.from_numpy(xg).double(), torch.from_numpy(wg).double(), torch.from_numpy(bg).double(), stride=4, padding=20,
                   groups=groups)
    assert torch.isfinite(yg).all() and rel(yg, F.leaky_relu(pre, 0.2)) < 2e-6
    gy = dev((rng.standard_normal(tuple(yg.shape)) / ws).astype(np.float32))
    gxg = P.conv1d_bwd_data(gy, yg, wgt, dg)
    pg = gy.double().cpu() * torch.where(yg.cpu() > 0, 1.0, 0.2).double()
    gxr = F.conv_transpose1d(pg, torch.from_numpy(wg).double(), stride=4, padding=20, groups=groups, output_padding=Lin - ((lo - 1) * 4 - 40 + 41))
    assert torch.isfinite(gxg).all() and rel(gxg, gxr) < 2e-6, rel(gxg, gxr)

    # ---- stride-8 transposed conv on its weight image
    Bt, Ci, Co, Li = 16, 256, 128, 256
    xc = (rng.standard_normal((Bt, Ci, Li)) / ws).astype(np.float32)
    wc = (rng.standard_normal((Ci, Co, 16)) / np.sqrt(2 * Ci) * ws).astype(np.float32)
    bc = (rng.standard_normal(Co) * 0.1).astype(np.float32)
    xct, wct, bct = dev(xc), dev(wc), dev(bc)
    dc, loc = P.convt_desc(xct.shape, wct.shape, 8, 4, act=L.ACT_LRELU)
    assert P.convt_img_bytes(dc) > 0
    yc = P.convt1d_fwd(xct, wct, bct, dc, loc)
    ycr = F.leaky_relu(F.conv_transpose1d(torch.from_numpy(xc).double(), torch.from_numpy(wc).double(), torch.from_numpy(bc).double(),
                                          stride=8, padding=4), 0.2)
    assert torch.isfinite(yc).all() and rel(yc, ycr) < 2e-6, rel(yc, ycr)
