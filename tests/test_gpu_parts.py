"""ms_conv1d_parts_*: one conv layer with shared weights over the discriminator's three scales in one launch
(reference discriminator/melgan.py:13-27: the same FullDiscriminator on x, pool(x), pool(pool(x))).

Checked against the single-part entry points (the launches they replace) and against float64 torch on the CPU,
at the layer geometries of discriminator/full.py:13-22 and at lengths no parts kernel takes (part-by-part path)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

# (Cin, Cout, K, stride, pad, groups) of FullDiscriminator.main[1..5]
GROUPED = [(16, 64, 41, 4, 20, 4), (64, 256, 41, 4, 20, 16), (256, 1024, 41, 4, 20, 64), (1024, 1024, 41, 4, 20, 256)]
K5 = (1024, 1024, 5, 1, 2, 1)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-300))


def _scale_lengths(L0, depth):
    """Input lengths of layer `depth` (0 = first grouped layer) at the three scales of an L0-sample window."""
    out = []
    L = L0
    for s in range(3):
        if s:
            L = L // 2 + 1                     # avg_pool1d(4, 2, 2)
        l = L
        for _ in range(depth):
            l = (l + 40 - 41) // 4 + 1
        out.append(l)
    return out


def _inputs(geo, B, lens, seed):
    Cin, Cout, K, stride, pad, groups = geo
    rng = np.random.default_rng(seed)
    xs = [dev(rng.standard_normal((B, Cin, l))) for l in lens]
    w = dev(rng.standard_normal((Cout, Cin // groups, K)) * (1.0 / np.sqrt(K * Cin // groups)))
    b = dev(rng.standard_normal((Cout,)) * 0.1)
    return xs, w, b


def _desc(geo, x, w, act=None):
    from featuresynth._ops import lib as L
    from featuresynth._ops import prims as P
    Cin, Cout, K, stride, pad, groups = geo
    return P.conv_desc(x.shape, w.shape, stride=stride, pad=pad, groups=groups, act=L.ACT_LRELU if act is None else act)


def _launches(d, tensors, which, image=False):
    from featuresynth._ops import lib as L
    parts = L.ConvParts()
    parts.count = len(tensors)
    for i, t in enumerate(tensors):
        parts.B[i], parts.Lin[i] = t.shape[0], t.shape[2]
        parts.x[i] = parts.y[i] = parts.gy[i] = parts.y_act[i] = parts.gx[i] = t.data_ptr()      # (16-byte aligned stand-ins)
    return L.load().ms_conv1d_parts_launches(d, parts, which, 1 if image else 0)


@pytest.mark.parametrize("layer", [0, 1, 2, 3])
@pytest.mark.parametrize("B,L0", [(3, 8192), (64, 8192), (2, 3000)], ids=["b3", "b64", "offgrid"])
def test_grouped_layer_over_three_scales(layer, B, L0):
    """Forward, backward data and weight gradient of a grouped k41 layer over the three scales: the parts call equals the
    three single calls it replaces -- bitwise for forward and backward data (same work units, same arithmetic), to
    summation order for the weight gradient (one reduction over all scales' slabs instead of three accumulations) -- and
    float64 torch.  At the window sizes of the reference (8192 samples) it is ONE launch; other lengths run part by part."""
    import torch.nn.functional as F
    from featuresynth._ops import lib as L
    from featuresynth._ops import prims as P
    geo = GROUPED[layer]
    if B == 64 and layer < 2:
        B = 16          # (keeps the float64 CPU reference of the wide maps in seconds)
    lens = _scale_lengths(L0, layer)
    xs, w, b = _inputs(geo, B, lens, 100 + layer)
    d, _ = _desc(geo, xs[0], w)
    one = L0 == 8192
    assert _launches(d, xs, 0) == (1 if one else 3)
    ys = P.conv1d_parts_fwd(xs, w, b, d)
    gys = [dev(np.random.default_rng(7 + i).standard_normal(tuple(y.shape))) for i, y in enumerate(ys)]
    adds = [dev(np.random.default_rng(17 + i).standard_normal(tuple(x.shape))) for i, x in enumerate(xs)]
    gxs = P.conv1d_parts_bwd_data(gys, ys, w, d, [x.shape for x in xs], gx_adds=adds)
    gw, gb = P.conv1d_parts_bwd_weight(xs, gys, ys, d, w.shape)
    gw_sum, gb_sum = None, None
    for i, x in enumerate(xs):
        di, lo = _desc(geo, x, w)
        y1, _ = P.conv1d_fwd(x, w, b, di, lo)
        assert torch.equal(ys[i], y1), ("forward", i)
        gx1 = P.conv1d_bwd_data(gys[i], y1, w, di, gx_add=adds[i])
        assert torch.equal(gxs[i], gx1), ("backward data", i)
        gw_sum, gb_sum = P.conv1d_bwd_weight(x, gys[i], y1, di, w.shape, gw_sum, gb_sum, accumulate=i > 0)
    assert rel(gw, gw_sum) < 2e-6 and rel(gb, gb_sum) < 2e-6, (rel(gw, gw_sum), rel(gb, gb_sum))
    # float64 reference (CPU)
    Cin, Cout, K, stride, pad, groups = geo
    wd, bd = w.double().cpu().requires_grad_(True), b.double().cpu().requires_grad_(True)
    tot = 0
    for i, x in enumerate(xs):
        xd = x.double().cpu().requires_grad_(True)
        yr = F.leaky_relu(F.conv1d(xd, wd, bd, stride=stride, padding=pad, groups=groups), 0.2)
        assert rel(ys[i].cpu(), yr.detach()) < 1e-6, ("forward vs float64", i, rel(ys[i].cpu(), yr.detach()))
        # the device differentiates LeakyReLU by the sign of ITS saved output
        pre_grad = gys[i].double().cpu() * torch.where(ys[i].cpu() > 0, 1.0, 0.2).double()
        pre = F.conv1d(xd, wd, bd, stride=stride, padding=pad, groups=groups)
        (gxr,) = torch.autograd.grad(pre, xd, pre_grad, retain_graph=True)
        assert rel(gxs[i].cpu() - adds[i].cpu(), gxr) < 2e-6, ("backward data vs float64", i)
        tot = tot + (pre * pre_grad).sum()
    gwr, gbr = torch.autograd.grad(tot, (wd, bd))
    assert rel(gw.cpu(), gwr) < 2e-6 and rel(gb.cpu(), gbr) < 2e-6, (rel(gw.cpu(), gwr), rel(gb.cpu(), gbr))
    # accumulate form: beta = 1 adds to what the slots hold
    gw2, gb2 = P.conv1d_parts_bwd_weight(xs, gys, ys, d, w.shape, gw.clone(), gb.clone(), accumulate=True)
    assert rel(gw2, 2 * gw) < 1e-6 and rel(gb2, 2 * gb) < 1e-6


@pytest.mark.parametrize("lens,B", [((29, 256, 100), 5), ((128,), 70), ((33, 65), 1)], ids=["edges", "single", "pair"])
def test_group4_layer_on_the_vector_pipe(lens, B, monkeypatch):
    """The 256-group layer (4 x 4 channels per group) runs in fp32 FMA on the vector pipe (csrc/gconv4.hip) for rows of
    29 .. 256 samples: shortest and longest rows, row counts that leave the last workgroup partly empty, one / two / three
    parts; float64 torch at 1e-6 (plain fp32 accumulation of 164 terms), the recorded kernel names, and the matrix-pipe
    kernels it replaces (MSYNTH_G4=0) at 2e-6."""
    import torch.nn.functional as F
    from featuresynth._ops import lib as L
    from featuresynth._ops import prims as P
    geo = GROUPED[3]
    xs, w, b = _inputs(geo, B, lens, 300 + len(lens))
    d, _ = _desc(geo, xs[0], w)
    L.profile_begin()
    ys = P.conv1d_parts_fwd(xs, w, b, d)
    gys = [dev(np.random.default_rng(27 + i).standard_normal(tuple(y.shape))) for i, y in enumerate(ys)]
    gxs = P.conv1d_parts_bwd_data(gys, ys, w, d, [x.shape for x in xs])
    gw, gb = P.conv1d_parts_bwd_weight(xs, gys, ys, d, w.shape)
    names = [cost.get("kernel", "") for _, cost, _ in L.profile_end()]
    assert names == ["k_g4_fwd", "k_g4_bwd_data", "k_g4_wgrad"], names
    Cin, Cout, K, stride, pad, groups = geo
    wd, bd = w.double().cpu().requires_grad_(True), b.double().cpu().requires_grad_(True)
    tot = 0
    for i, x in enumerate(xs):
        xd = x.double().cpu().requires_grad_(True)
        pre = F.conv1d(xd, wd, bd, stride=stride, padding=pad, groups=groups)
        assert rel(ys[i].cpu(), F.leaky_relu(pre, 0.2).detach()) < 1e-6, ("forward", i)
        pre_grad = gys[i].double().cpu() * torch.where(ys[i].cpu() > 0, 1.0, 0.2).double()
        (gxr,) = torch.autograd.grad(pre, xd, pre_grad, retain_graph=True)
        assert rel(gxs[i].cpu(), gxr) < 1e-6, ("backward data", i, rel(gxs[i].cpu(), gxr))
        tot = tot + (pre * pre_grad).sum()
    gwr, gbr = torch.autograd.grad(tot, (wd, bd))
    assert rel(gw.cpu(), gwr) < 1e-6 and rel(gb.cpu(), gbr) < 1e-6, (rel(gw.cpu(), gwr), rel(gb.cpu(), gbr))
    monkeypatch.setenv("MSYNTH_G4", "0")
    ys0 = P.conv1d_parts_fwd(xs, w, b, d)
    gxs0 = P.conv1d_parts_bwd_data(gys, ys, w, d, [x.shape for x in xs])
    gw0, gb0 = P.conv1d_parts_bwd_weight(xs, gys, ys, d, w.shape)
    for i in range(len(xs)):
        assert rel(ys[i], ys0[i]) < 2e-6 and rel(gxs[i], gxs0[i]) < 2e-6
    assert rel(gw, gw0) < 2e-6 and rel(gb, gb0) < 2e-6


@pytest.mark.parametrize("B,rows", [(64, 64), (64, 32), (6, 6)], ids=["b64", "b64_grad32", "small"])
def test_k5_layer_over_three_scales(B, rows):
    """The 1024 -> 1024 k5 layer on its weight image over rows of 32 / 17 / 9 samples: one launch without split-K slabs at
    the batch sizes of the train step, against the per-scale image launches (1e-6: other slice grouping) and float64; the
    backward pass may cover only the leading rows of what the forward saved (G-step: the fake half of [fake; real])."""
    import torch.nn.functional as F
    from featuresynth._ops import lib as L
    from featuresynth._ops import prims as P
    lens = [32, 17, 9]
    xs, w, b = _inputs(K5, B, lens, 55)
    d, _ = _desc(K5, xs[0], w)
    assert P.conv_img_bytes(d) > 0
    img, imgb = P.conv_img_pack(d, w), P.conv_img_pack(d, w, backward=True)
    assert _launches(d, xs, 0, image=True) == (1 if B >= 32 else 3)
    ys = P.conv1d_parts_fwd(xs, w, b, d, image=img)
    gys = [dev(np.random.default_rng(3 + i).standard_normal((rows,) + tuple(y.shape[1:]))) for i, y in enumerate(ys)]
    adds = [dev(np.random.default_rng(13 + i).standard_normal((rows,) + tuple(x.shape[1:]))) for i, x in enumerate(xs)]
    gxs = P.conv1d_parts_bwd_data(gys, ys, w, d, [x.shape for x in xs], gx_adds=adds, image_bwd=imgb)
    wd, bd = w.double().cpu(), b.double().cpu()
    for i, x in enumerate(xs):
        di, lo = _desc(K5, x, w)
        y1 = P.conv1d_img_fwd(x, img, b, di, lo)
        assert rel(ys[i], y1) < 1e-6, ("forward vs single image launch", i, rel(ys[i], y1))
        yr = F.leaky_relu(F.conv1d(x.double().cpu(), wd, bd, padding=2), 0.2)
        assert rel(ys[i].cpu(), yr) < 1e-6, ("forward vs float64", i, rel(ys[i].cpu(), yr))
        dr, _ = _desc(K5, x[:rows], w)
        gx1 = P.conv1d_img_bwd_data(gys[i], ys[i][:rows].contiguous(), imgb, dr, gx_add=adds[i])
        assert rel(gxs[i], gx1) < 1e-6, ("backward data vs single image launch", i, rel(gxs[i], gx1))
        pre_grad = gys[i].double().cpu() * torch.where(ys[i][:rows].cpu() > 0, 1.0, 0.2).double()
        gxr = F.conv_transpose1d(pre_grad, wd, padding=2)
        assert rel(gxs[i].cpu() - adds[i].cpu(), gxr) < 2e-6, ("backward data vs float64", i)
    # run-to-run determinism of the one-launch form
    ys2 = P.conv1d_parts_fwd(xs, w, b, d, image=img)
    assert all(torch.equal(a, c) for a, c in zip(ys, ys2))
    # weight gradient over the three scales: one launch (the octets of all parts form one contraction), against the three
    # accumulated single calls and float64
    assert _launches(d, gys, 2) == 1
    gw, gb = P.conv1d_parts_bwd_weight(xs, gys, ys, d, w.shape)
    gw_s, gb_s = None, None
    for i, x in enumerate(xs):
        dr, _ = _desc(K5, x[:rows], w)
        gw_s, gb_s = P.conv1d_bwd_weight(x[:rows].contiguous(), gys[i], ys[i][:rows].contiguous(), dr, w.shape, gw_s, gb_s,
                                         accumulate=i > 0)
    assert rel(gw, gw_s) < 2e-6 and rel(gb, gb_s) < 2e-6, (rel(gw, gw_s), rel(gb, gb_s))
    gwr = torch.zeros_like(wd)
    gbr = torch.zeros_like(bd)
    for i, x in enumerate(xs):
        pre_grad = gys[i].double().cpu() * torch.where(ys[i][:rows].cpu() > 0, 1.0, 0.2).double()
        xd = F.pad(x[:rows].double().cpu(), (2, 2))
        # gw[co, ci, k] = sum_{b, t} pre_grad[b, co, t] x[b, ci, t + k - 2]
        for k in range(5):
            gwr[:, :, k] += torch.einsum("bot,bit->oi", pre_grad, xd[:, :, k:k + x.shape[2]])
        gbr += pre_grad.sum((0, 2))
    assert rel(gw.cpu(), gwr) < 2e-6 and rel(gb.cpu(), gbr) < 2e-6, (rel(gw.cpu(), gwr), rel(gb.cpu(), gbr))
    gw2, _ = P.conv1d_parts_bwd_weight(xs, gys, ys, d, w.shape)
    assert torch.equal(gw, gw2)


FIRST = (1, 16, 15, 1, 7, 1)          # discriminator/full.py:14 (+ LeakyReLU)
JUDGE = (1024, 1, 3, 1, 1, 1)        # discriminator/full.py:22 (no activation)


@pytest.mark.parametrize("which", ["first", "judge"])
@pytest.mark.parametrize("B,rows,L0", [(64, 64, 8192), (64, 32, 8192), (3, 3, 8192), (2, 2, 3000)],
                         ids=["b64", "b64_grad32", "b3", "offgrid"])
def test_thin_layers_over_three_scales(which, B, rows, L0):
    """The first conv (1 -> 16, k15, LeakyReLU) and the judge conv (1024 -> 1, k3) over the three scales, every pass as ONE
    launch (csrc/disc_parts.hip) for rows of any length: against the per-scale entry points (1e-6: other summation order)
    and float64 torch; deterministic; the backward passes may cover only the leading batch rows of what was saved."""
    import torch.nn.functional as F
    from featuresynth._ops import lib as L
    from featuresynth._ops import prims as P
    geo = FIRST if which == "first" else JUDGE
    act = L.ACT_LRELU if which == "first" else L.ACT_NONE
    if which == "first":
        lens = _scale_lengths(L0, 0)
    else:
        lens = [l for l in _scale_lengths(L0, 4)]            # rows of 32 / 17 / 9 samples behind the four stride-4 layers
    Cin, Cout, K, stride, pad, groups = geo
    xs, w, b = _inputs(geo, B, lens, 77)
    d, _ = _desc(geo, xs[0], w, act)
    assert _launches(d, xs, 0) == 1 and _launches(d, xs, 1) == 1 and _launches(d, xs, 2) == 1
    ys = P.conv1d_parts_fwd(xs, w, b, d)
    ys_again = P.conv1d_parts_fwd(xs, w, b, d)
    gys = [dev(np.random.default_rng(5 + i).standard_normal((rows,) + tuple(y.shape[1:]))) for i, y in enumerate(ys)]
    adds = [dev(np.random.default_rng(25 + i).standard_normal((rows,) + tuple(x.shape[1:]))) for i, x in enumerate(xs)]
    ya = ys if which == "first" else None
    gxs = P.conv1d_parts_bwd_data(gys, ya, w, d, [x.shape for x in xs], gx_adds=adds)
    gw, gb = P.conv1d_parts_bwd_weight(xs, gys, ya, d, w.shape)
    gw2, gb2 = P.conv1d_parts_bwd_weight(xs, gys, ya, d, w.shape)
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)
    wd, bd = w.double().cpu(), b.double().cpu()
    gw_s = gb_s = None
    gwr, gbr = torch.zeros_like(wd), torch.zeros_like(bd)
    for i, x in enumerate(xs):
        assert torch.equal(ys[i], ys_again[i])
        di, lo = _desc(geo, x, w, act)
        y1, _ = P.conv1d_fwd(x, w, b, di, lo)
        assert rel(ys[i], y1) < 1e-6, ("forward vs per-scale launch", i, rel(ys[i], y1))
        pre = F.conv1d(x.double().cpu(), wd, bd, padding=pad)
        yr = F.leaky_relu(pre, 0.2) if which == "first" else pre
        assert rel(ys[i].cpu(), yr) < 1e-6, ("forward vs float64", i, rel(ys[i].cpu(), yr))
        xr = x[:rows].contiguous()
        dr, _ = _desc(geo, xr, w, act)
        yar = ys[i][:rows].contiguous() if which == "first" else None
        gx1 = P.conv1d_bwd_data(gys[i], yar, w, dr, gx_add=adds[i])
        assert rel(gxs[i], gx1) < 1e-6, ("backward data vs per-scale launch", i, rel(gxs[i], gx1))
        gw_s, gb_s = P.conv1d_bwd_weight(xr, gys[i], yar, dr, w.shape, gw_s, gb_s, accumulate=i > 0)
        pg = gys[i].double().cpu()
        if which == "first":
            pg = pg * torch.where(ys[i][:rows].cpu() > 0, 1.0, 0.2).double()
        gxr = F.conv_transpose1d(pg, wd, padding=pad)
        assert rel(gxs[i].cpu() - adds[i].cpu(), gxr) < 2e-6, ("backward data vs float64", i)
        xp = F.pad(xr.double().cpu(), (pad, pad))
        for k in range(K):
            gwr[:, :, k] += torch.einsum("bot,bit->oi", pg, xp[:, :, k:k + x.shape[2]])
        gbr += pg.sum((0, 2))
    assert rel(gw, gw_s) < 2e-6 and rel(gb, gb_s) < 2e-5, (rel(gw, gw_s), rel(gb, gb_s))
    assert rel(gw.cpu(), gwr) < 2e-6 and rel(gb.cpu(), gbr) < 2e-5, (rel(gw.cpu(), gwr), rel(gb.cpu(), gbr))
    gw3, gb3 = P.conv1d_parts_bwd_weight(xs, gys, ya, d, w.shape, gw.clone(), gb.clone(), accumulate=True)
    assert rel(gw3, 2 * gw) < 1e-6 and rel(gb3, 2 * gb) < 1e-6


def test_three_scale_pass_equals_one_scale_passes():
    """The discriminator pass layer-by-layer over all scales (parts launches) against the SAME layers applied to each scale on
    its own (scales = 0: the parts entry points then run the per-scale kernels): same features, judgements, input gradient
    and parameter gradients to summation order."""
    from featuresynth._ops import graph as G
    from featuresynth._ops import prims as P
    from featuresynth._synthetic import module_param_shapes, synthetic_samples, synthetic_state_dict
    import featuresynth as fs
    dmod = fs.MelGanDiscriminator()
    sd = synthetic_state_dict(module_param_shapes(dmod), seed=8, bias_scale=0.02)
    params = [dev(v) for v in sd.values()]
    x = dev(synthetic_samples(4, 8192, rank=3))
    feats, judges, ctx = G.melgan_forward(x, params)
    g_feats = [[torch.full_like(t, 1e-3) for t in grp] for grp in feats]
    g_judges = [torch.full_like(j, -0.25) for j in judges]
    gx, sink = G.melgan_backward(ctx, params, g_feats, g_judges, None, need_gx=True, need_wgrad=True)
    # one scale at a time
    xs = [x, P.avg_pool_fwd(x)]
    xs.append(P.avg_pool_fwd(xs[1]))
    gws, gxs = None, []
    for s in range(3):
        f1, j1, c1 = G.melgan_forward(xs[s], params, scales=0)
        assert rel(j1[0], judges[s]) < 1e-5, (s, rel(j1[0], judges[s]))          # (small sums of cancelling terms)
        for li in range(6):
            assert rel(f1[0][li], feats[s][li]) < 2e-6, (s, li, rel(f1[0][li], feats[s][li]))
        gx1, sk = G.melgan_backward(c1, params, [g_feats[s]], [g_judges[s]], None, need_gx=True, need_wgrad=True)
        gxs.append(gx1)
        gws = [t.clone() for t in sk.t] if gws is None else [a + t for a, t in zip(gws, sk.t)]
    tot = gxs[0] + P.avg_pool_bwd(gxs[1] + P.avg_pool_bwd(gxs[2], xs[1].shape), xs[0].shape)
    assert rel(gx, tot) < 1e-5, rel(gx, tot)
    for i, (a, c) in enumerate(zip(sink.t, gws)):
        assert rel(a, c) < 1e-5, (i, rel(a, c))
