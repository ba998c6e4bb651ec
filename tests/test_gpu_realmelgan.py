"""GPU parity of the weight-normed MelGAN (reference experiment/realmelgan.py)
and of the ops it adds: activation in front of a conv, 1x1 convs, reflection-pad backward,
AvgPool1d(4,2,1,count_include_pad=False), weight norm."""
import numpy as np
import pytest

from conftest import rel_l2, stable_seed

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def _ref_conv(x, w, b, pad, dil, in_act, act, reflect, residual=None, out_mask=None):
    """out_mask: the LeakyReLU branch each output took ON THE DEVICE (oracle/torch_graph.py:_LReluMasked):
    both backward passes then differentiate the same piecewise-linear function."""
    import torch.nn.functional as F
    from oracle import torch_graph as TG
    h = F.leaky_relu(x, 0.2) if in_act else x
    if reflect and pad:
        h = F.pad(h, (pad, pad), mode="reflect")
        y = F.conv1d(h, w, b, dilation=dil)
    else:
        y = F.conv1d(h, w, b, padding=pad, dilation=dil)
    if act == 1:
        y = F.leaky_relu(y, 0.2) if out_mask is None else TG._LReluMasked.apply(y, out_mask)
    elif act == 2:
        y = torch.tanh(y)
    return y if residual is None else y + residual


CASES = [  # name, B, Cin, L, Cout, K, pad, dil, in_act, act, reflect, residual
    ("res_conv3_c64", 2, 64, 300, 64, 3, 3, 3, 1, 1, True, False),
    ("res_conv3_c256_d9", 2, 256, 70, 256, 3, 9, 9, 1, 1, True, False),
    # reflect-padded dilated convs at 16-byte aligned lengths (row-tile weight-gradient kernel, mirrored vectors)
    ("res_conv3_c128_d9_l2048", 1, 128, 2048, 128, 3, 9, 9, 1, 1, True, False),
    ("res_conv3_c256_d1_l256", 2, 256, 256, 256, 3, 1, 1, 1, 1, True, False),
    ("res_conv3_c64_d9_l64", 3, 64, 64, 64, 3, 9, 9, 1, 1, True, False),
    ("res_conv3_c64_d3_l132", 2, 64, 132, 64, 3, 3, 3, 1, 1, True, False),
    ("res_conv1x1_c128", 2, 128, 257, 128, 1, 0, 1, 0, 0, False, True),
    ("shortcut_1x1_c32", 3, 32, 1031, 32, 1, 0, 1, 0, 0, False, False),
    # pointwise convs on the pipelined row kernel (16-byte aligned rows), incl. packed short rows
    ("res_conv1x1_c128_l256", 2, 128, 256, 128, 1, 0, 1, 0, 0, False, True),
    ("shortcut_1x1_c32_l1024", 3, 32, 1024, 32, 1, 0, 1, 0, 0, False, False),
    ("shortcut_1x1_c256_l64", 5, 256, 64, 256, 1, 0, 1, 0, 0, False, True),
    ("shortcut_1x1_c64_l2048", 2, 64, 2048, 64, 1, 0, 1, 0, 0, False, False),
    ("first_k7_reflect", 2, 128, 9, 512, 7, 3, 1, 0, 0, True, False),
    ("last_k7_reflect_tanh", 2, 32, 515, 1, 7, 3, 1, 1, 2, True, False),
    ("d_first_k15_reflect", 2, 1, 1025, 16, 15, 7, 1, 0, 1, True, False),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_ex_vs_torch_cpu(case):
    from featuresynth._ops import functional as F_
    name, B, Cin, L, Cout, K, pad, dil, in_act, act, reflect, with_res = case
    rng = np.random.default_rng(stable_seed(name))
    x = rng.standard_normal((B, Cin, L)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, K)) * 0.1).astype(np.float32)
    b = (rng.standard_normal((Cout,)) * 0.1).astype(np.float32)
    res = rng.standard_normal((B, Cout, L)).astype(np.float32) if with_res else None
    xt, wt, bt = dev(x).requires_grad_(True), dev(w).requires_grad_(True), dev(b).requires_grad_(True)
    rt = dev(res).requires_grad_(True) if with_res else None
    y = F_.Conv1dExFn.apply(xt, wt, bt, rt, 1, pad, dil, 1, 1 if reflect else 0, act, in_act)
    gy = rng.standard_normal(y.shape).astype(np.float32)
    y.backward(dev(gy))
    # double-precision CPU reference; LeakyReLU masks can only differ at rounding-level activations
    xc, wc, bc = [torch.from_numpy(a).double().requires_grad_(True) for a in (x, w, b)]
    rc = torch.from_numpy(res).double().requires_grad_(True) if with_res else None
    # (an output within fp32 rounding of zero may sit on the other LeakyReLU branch than in the float64
    #  reference: the reference's backward takes the branch the device took; no case combines the
    #  activation behind the conv with a residual, so the device output's sign IS that branch)
    assert not (act == 1 and with_res)
    mask = torch.from_numpy(host(y) > 0) if act == 1 else None
    yc = _ref_conv(xc, wc, bc, pad, dil, in_act, act, reflect, rc, out_mask=mask)
    yc.backward(torch.from_numpy(gy).double())
    assert rel_l2(host(y), yc.detach().numpy()) < 1e-5
    assert rel_l2(host(xt.grad), xc.grad.numpy()) < 1e-4, "gx"
    assert rel_l2(host(wt.grad), wc.grad.numpy()) < 1e-4, "gw"
    assert rel_l2(host(bt.grad), bc.grad.numpy()) < 1e-4, "gb"
    if with_res:
        assert rel_l2(host(rt.grad), rc.grad.numpy()) < 1e-6


CONVT_CASES = [  # name, B, Cin, L, Cout, K, S  (LeakyReLU in front, none behind: realmelgan.py:62-70)
    ("up8_c512_l32", 2, 512, 32, 256, 16, 8),
    ("up8_c256_l64", 2, 256, 64, 128, 16, 8),
    ("up2_c128_l128", 2, 128, 128, 64, 4, 2),
    ("up2_c64_l96", 3, 64, 96, 32, 4, 2),
    ("up8_c64_l7", 2, 64, 7, 32, 16, 8),
]


@pytest.mark.parametrize("case", CONVT_CASES, ids=[c[0] for c in CONVT_CASES])
def test_convt_ex_vs_torch_cpu(case):
    import torch.nn.functional as F
    from featuresynth._ops import functional as F_
    name, B, Cin, L, Cout, K, S = case
    rng = np.random.default_rng(stable_seed(name))
    x = rng.standard_normal((B, Cin, L)).astype(np.float32)
    w = (rng.standard_normal((Cin, Cout, K)) * 0.1).astype(np.float32)
    b = (rng.standard_normal((Cout,)) * 0.1).astype(np.float32)
    xt, wt, bt = dev(x).requires_grad_(True), dev(w).requires_grad_(True), dev(b).requires_grad_(True)
    y = F_.ConvTranspose1dExFn.apply(xt, wt, bt, S, S // 2, 0, 1)
    gy = rng.standard_normal(y.shape).astype(np.float32)
    y.backward(dev(gy))
    xc, wc, bc = [torch.from_numpy(a).double().requires_grad_(True) for a in (x, w, b)]
    yc = F.conv_transpose1d(F.leaky_relu(xc, 0.2), wc, bc, stride=S, padding=S // 2)
    yc.backward(torch.from_numpy(gy).double())
    assert rel_l2(host(y), yc.detach().numpy()) < 1e-5
    assert rel_l2(host(xt.grad), xc.grad.numpy()) < 1e-5, "gx"
    assert rel_l2(host(wt.grad), wc.grad.numpy()) < 1e-5, "gw"
    assert rel_l2(host(bt.grad), bc.grad.numpy()) < 1e-5, "gb"


@pytest.mark.parametrize("L", [64, 67, 5])
def test_avg_pool_421(L):
    import torch.nn.functional as F
    from featuresynth._ops import functional as F_
    rng = np.random.default_rng(L)
    x = rng.standard_normal((2, 3, L)).astype(np.float32)
    xt = dev(x).requires_grad_(True)
    y = F_.AvgPool421Fn.apply(xt)
    xc = torch.from_numpy(x).requires_grad_(True)
    yc = F.avg_pool1d(xc, 4, stride=2, padding=1, count_include_pad=False)
    assert tuple(y.shape) == tuple(yc.shape)
    gy = rng.standard_normal(yc.shape).astype(np.float32)
    y.backward(dev(gy)); yc.backward(torch.from_numpy(gy))
    assert rel_l2(host(y), yc.detach().numpy()) < 1e-6
    assert rel_l2(host(xt.grad), xc.grad.numpy()) < 1e-6


@pytest.mark.parametrize("shape", [(16, 1, 15), (256, 256, 3), (512, 256, 16), (1, 32, 7)])
def test_weight_norm(shape):
    from featuresynth._ops import functional as F_
    rng = np.random.default_rng(sum(shape))
    v = rng.standard_normal(shape).astype(np.float32)
    g = (0.5 + rng.random((shape[0], 1, 1))).astype(np.float32)
    vt, gt = dev(v).requires_grad_(True), dev(g).requires_grad_(True)
    w = F_.WeightNormFn.apply(vt, gt)
    vc, gc = torch.from_numpy(v).double().requires_grad_(True), torch.from_numpy(g).double().requires_grad_(True)
    wc = gc * vc / vc.reshape(shape[0], -1).norm(dim=1).reshape(-1, 1, 1)
    gw = rng.standard_normal(shape).astype(np.float32)
    w.backward(dev(gw)); wc.backward(torch.from_numpy(gw).double())
    assert rel_l2(host(w), wc.detach().numpy()) < 1e-6
    assert rel_l2(host(vt.grad), vc.grad.numpy()) < 1e-5
    assert rel_l2(host(gt.grad), gc.grad.numpy()) < 1e-5


def test_weight_norm_multi():
    """All layers of a network in one launch (70 tensors: two descriptor chunks) vs float64 torch."""
    from featuresynth._ops import functional as F_
    rng = np.random.default_rng(5)
    shapes = [(16, 1, 15), (64, 4, 41), (256, 256, 3), (512, 256, 16), (1, 32, 7), (1024, 4, 41), (3, 5, 1)] * 10
    vs = [rng.standard_normal(s).astype(np.float32) for s in shapes]
    gs = [(0.5 + rng.random((s[0], 1, 1))).astype(np.float32) for s in shapes]
    gws = [rng.standard_normal(s).astype(np.float32) for s in shapes]
    vt = [dev(v).requires_grad_(True) for v in vs]
    gt = [dev(g).requires_grad_(True) for g in gs]
    args = []
    for a, b in zip(vt, gt):
        args += [a, b]
    ws = F_.WeightNormMultiFn.apply(*args)
    torch.autograd.backward(list(ws), [dev(g) for g in gws])
    for i, s in enumerate(shapes):
        vc, gc = torch.from_numpy(vs[i]).double().requires_grad_(True), torch.from_numpy(gs[i]).double().requires_grad_(True)
        wc = gc * vc / vc.reshape(s[0], -1).norm(dim=1).reshape(-1, 1, 1)
        wc.backward(torch.from_numpy(gws[i]).double())
        assert rel_l2(host(ws[i]), wc.detach().numpy()) < 1e-6, i
        assert rel_l2(host(vt[i].grad), vc.grad.numpy()) < 1e-5, i
        assert rel_l2(host(gt[i].grad), gc.grad.numpy()) < 1e-5, i


def _nets():
    from featuresynth._synthetic import module_param_shapes, synthetic_state_dict
    from featuresynth.experiment import realmelgan as R
    g, d = R.Generator(128, 32, 3), R.Discriminator(3, 16, 4, 4)
    gsd = synthetic_state_dict(module_param_shapes(g), seed=21, weight_scale=0.3, bias_scale=0.05)
    dsd = synthetic_state_dict(module_param_shapes(d), seed=22, weight_scale=0.3, bias_scale=0.05)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in gsd.items()})
    d.load_state_dict({k: torch.from_numpy(v) for k, v in dsd.items()})
    return g.cuda(), d.cuda(), gsd, dsd


def test_realmelgan_forward_golden(golden):
    from featuresynth._synthetic import strided_sample, synthetic_samples
    z = golden("realmelgan")
    g, d, _, _ = _nets()
    assert list(g.state_dict().keys()) == list(z["g_param_names"])
    assert list(d.state_dict().keys()) == list(z["d_param_names"])
    feat = np.random.default_rng(5).standard_normal((2, 128, 6)).astype(np.float32)
    with torch.no_grad():
        y = g(dev(feat))
        feats, judges = d(dev(synthetic_samples(2, 2048, rank=9)), None)
    assert tuple(y.shape) == z["g/y_ref32"].shape
    e = rel_l2(host(y), z["g/y_ref32"])
    print("RealMelGan generator rel-L2 vs reference: %.3e" % e)
    assert e < 1e-4
    assert len(feats) == 3 and all(len(f) == 6 for f in feats)
    for s in range(3):
        assert rel_l2(host(judges[s]), z["d/j%d_ref32" % s]) < 1e-4
        for i in range(6):
            assert tuple(feats[s][i].shape) == tuple(z["d/f%d_%d_shape" % (s, i)])
            assert rel_l2(strided_sample(host(feats[s][i])), z["d/f%d_%d_smp_ref32" % (s, i)]) < 1e-4


def test_realmelgan_train_steps_golden(golden):
    """One D-step and one G-step through featuresynth.train (the variant that trains at the
    reference's HEAD) against the reference's trainers: losses and every parameter gradient."""
    from featuresynth import loss as LS
    from featuresynth._synthetic import strided_sample, synthetic_features, synthetic_samples
    from featuresynth.experiment import realmelgan as R
    from featuresynth.train import DiscriminatorTrainer, GeneratorTrainer
    z = golden("realmelgan")
    B, T = 2, 8
    samples, feats = dev(synthetic_samples(B, T * 256, rank=3)), dev(synthetic_features(B, 128, T, rank=3))
    for kind in ("d", "g"):
        g, d, _, _ = _nets()
        go = torch.optim.Adam(g.parameters(), lr=1e-4, betas=(0.5, 0.9))
        do = torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.9))
        if kind == "d":
            r = DiscriminatorTrainer(g, go, d, do, LS.mel_gan_disc_loss).train(samples, feats)
            assert abs(r["d_loss"] - float(z["step/d_loss"][0])) < 1e-4
            net = d
        else:
            r = GeneratorTrainer(g, go, d, do, R.mel_gan_gen_loss).train(samples, feats)
            ref = float(z["step/g_loss"][0])
            assert abs(r["g_loss"] - ref) < 1e-4 * abs(ref)
            assert rel_l2(strided_sample(r["fake"]), z["step/fake_smp"]) < 1e-4
            net = g
        worst = 0.0
        for k, p in net.named_parameters():
            e = rel_l2(strided_sample(host(p.grad)), z["step/%sgrad_smp/%s" % (kind, k)])
            worst = max(worst, e)
            assert e < 1e-2, (kind, k, e)      # deep LeakyReLU-mask flips (the tight gate: the flip-aware test below)
        print("RealMelGan %s-step worst grad rel-L2 %.2e" % (kind, worst))


def _hooked(net, classes):
    """Forward hooks on every submodule of the given classes: {module name: [outputs, one per call]}."""
    outs, handles = {}, []
    for name, m in net.named_modules():
        if isinstance(m, classes):
            handles.append(m.register_forward_hook(
                lambda mod, inp, out, name=name: outs.setdefault(name, []).append(out.detach() if not out.requires_grad else out)))
    return outs, handles


def test_realmelgan_train_steps_vs_flip_aware_oracle():
    """The D-step and the G-step of the weight-normed variant against the float64 restatement (oracle/torch_graph_real.py)
    with every LeakyReLU branch taken from the DEVICE's activations (forward hooks on the product modules): what is left
    is the kernels' rounding, so every parameter gradient is held to 1e-4 rel-L2 (median 1e-5; the headline model's gate in
    test_gpu_networks.py::test_train_steps_vs_oracle is 1e-3 / 1e-4); the fp32 reference fixtures above can only be met to
    1e-2 because rounding-level pre-activations take the other branch there."""
    from featuresynth import loss as LS
    from featuresynth._synthetic import synthetic_features, synthetic_samples
    from featuresynth.experiment import realmelgan as R
    from featuresynth.train import DiscriminatorTrainer, GeneratorTrainer
    from featuresynth.util.modules import Fused
    from oracle import torch_graph_real as RG
    B, T = 2, 8
    samples, feats = synthetic_samples(B, T * 256, rank=3), synthetic_features(B, 128, T, rank=3)
    s64, f64 = torch.from_numpy(samples).double(), torch.from_numpy(feats).double()
    for kind in ("d", "g"):
        g, d, gsd, dsd = _nets()
        go = torch.optim.Adam(g.parameters(), lr=1e-4, betas=(0.5, 0.9))
        do = torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.9))
        g_outs, h1 = _hooked(g, (R.WNConv1d, R.WNConvTranspose1d, R.ResnetBlock))
        d_outs, h2 = _hooked(d, (torch.nn.Sequential,))
        if kind == "d":
            r = DiscriminatorTrainer(g, go, d, do, LS.mel_gan_disc_loss).train(dev(samples), dev(feats))
            loss_dev, net = r["d_loss"], d
        else:
            r = GeneratorTrainer(g, go, d, do, R.mel_gan_gen_loss).train(dev(samples), dev(feats))
            loss_dev, net = r["g_loss"], g
        for h in h1 + h2:
            h.remove()

        def d_masks(rows):
            m = {}
            for name, calls in d_outs.items():
                if ".layer_" not in name:
                    continue
                live = [o for o in calls if o.requires_grad]        # (the G-step's real pass runs under no_grad)
                assert len(live) == 1, (name, len(calls))
                m[name] = (live[0].detach()[rows] > 0).cpu()
            assert len(m) == 18
            return m

        gp = RG.to_params(gsd, requires_grad=kind == "g", dtype=torch.float64)
        dp = RG.to_params(dsd, requires_grad=kind == "d", dtype=torch.float64)
        if kind == "d":
            with torch.no_grad():
                fake = RG.generator(gp, f64)
            _, fj = RG.discriminator(dp, fake, masks=d_masks(slice(0, B)))
            _, rj = RG.discriminator(dp, s64, masks=d_masks(slice(B, 2 * B)))
            loss = RG.disc_loss(rj, fj)
            params = dp
        else:
            gm, prev = {}, None
            for idx, m in g.model.named_children():
                name = "model." + idx
                if isinstance(m, Fused):
                    continue
                if prev is not None:
                    gm["pre." + name] = (g_outs[prev][-1].detach() > 0).cpu()
                if isinstance(m, R.ResnetBlock):
                    gm["mid." + name] = (g_outs[name + ".block.2"][-1].detach() > 0).cpu()
                prev = name
            assert len(gm) == 4 + 12 + 12 + 1
            fake = RG.generator(gp, f64, masks=gm)
            ff, fj = RG.discriminator(dp, fake, masks=d_masks(slice(0, B)))
            with torch.no_grad():
                rf, rj = RG.discriminator(dp, s64)
            loss = RG.gen_loss(rf, ff, fj)
            params = gp
            assert rel_l2(r["fake"], fake.detach().numpy()) < 1e-5
        loss.backward()
        loss = float(loss.detach())
        assert abs(loss_dev - loss) <= 1e-5 * max(1.0, abs(loss)), (kind, loss_dev, loss)
        errs = {}
        for k, p in net.named_parameters():
            ref = params[k].grad.numpy()
            errs[k] = rel_l2(host(p.grad), ref) if np.linalg.norm(ref) > 1e-12 else float(np.abs(host(p.grad)).max())
        worst = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
        med = float(np.median(list(errs.values())))
        print("RealMelGan %s-step vs flip-aware float64 oracle: worst %s, median %.2e" % (kind, worst, med))
        assert worst[0][1] < 1e-4, (kind, worst)      # measured: 4.6e-6 (D-step), 8.2e-7 (G-step)
        assert med < 1e-5, (kind, worst, med)


def test_realmelgan_native_graph_path(monkeypatch):
    """FlatAdam + hipGraph replay of the weight-normed variant walks the eager trajectory."""
    import featuresynth as fs
    from featuresynth import loss as LS
    from featuresynth._synthetic import synthetic_features, synthetic_samples
    from featuresynth.experiment import realmelgan as R
    from featuresynth.train import DiscriminatorTrainer, GeneratorTrainer
    B, T = 2, 4
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MSYNTH_GRAPH", mode)
        g, d, _, _ = _nets()
        go = fs.FlatAdam(g.parameters(), lr=1e-4, betas=(0.5, 0.9))
        do = fs.FlatAdam(d.parameters(), lr=1e-4, betas=(0.5, 0.9))
        dt = DiscriminatorTrainer(g, go, d, do, LS.mel_gan_disc_loss)
        gt = GeneratorTrainer(g, go, d, do, R.mel_gan_gen_loss)
        losses = []
        for step in range(6):
            s = dev(synthetic_samples(B, T * 256, rank=step))
            f = dev(synthetic_features(B, 128, T, rank=step))
            losses.append(dt.train(s, f)["d_loss"] if step % 2 == 0 else gt.train(s, f)["g_loss"])
        if mode == "1":
            assert dt._runner.graphs and gt._runner.graphs and not dt._runner.disabled
        out[mode] = losses
    for a, b in zip(out["0"], out["1"]):
        assert abs(a - b) <= 1e-4 * abs(a) + 1e-6, (out["0"], out["1"])


@pytest.mark.parametrize("d", [1, 9])
def test_weight_normed_residual_atom_golden(golden, d):
    """ResidualAtom(add_weight_norm=True) (reference util/modules.py:350-388 with torch weight_norm on both
    convs): state_dict keys, forward, input gradient and the gradients of bias / weight_g / weight_v."""
    from featuresynth.util.modules import ResidualAtom
    z = golden("partial_rows")
    nm = "atom_wn_d%d" % d
    atom = ResidualAtom(8, d, add_weight_norm=True)
    assert list(atom.state_dict().keys()) == list(z[nm + "/param_names"])
    atom.load_state_dict({k: torch.from_numpy(z[nm + "/sd/" + k]) for k in z[nm + "/param_names"]})
    atom.cuda()
    x = dev(z[nm + "/x"]).requires_grad_(True)
    y = atom(x)
    assert rel_l2(host(y), z[nm + "/y"]) < 1e-5
    y.backward(dev(z[nm + "/gy"]))
    assert rel_l2(host(x.grad), z[nm + "/gx"]) < 1e-4
    for k, p in atom.named_parameters():
        assert rel_l2(host(p.grad), z[nm + "/grad/" + k]) < 1e-4, k


def test_discriminator_conditioning_branch_golden(golden):
    """Discriminator(conditioning_channels=128): the `layer_cond` branch (realmelgan.py:128-136,149-152) --
    mel features average-pooled to the feature map's rate (HIP avg_pool1d(k)), stacked in front of it, three
    weight-normed k3 convs -- against the imported reference: outputs and gradients."""
    from featuresynth._synthetic import (module_param_shapes, strided_sample, synthetic_features, synthetic_samples,
                                         synthetic_state_dict)
    from featuresynth.experiment import realmelgan as R
    z = golden("partial_rows")
    disc = R.Discriminator(3, 16, 4, 4, conditioning_channels=128)
    assert list(disc.state_dict().keys()) == list(z["cond/param_names"])
    disc.load_state_dict({k: torch.from_numpy(v) for k, v in
                          synthetic_state_dict(module_param_shapes(disc), seed=45, weight_scale=0.3, bias_scale=0.05).items()})
    disc.cuda()
    x = dev(synthetic_samples(2, 2048, rank=11)).requires_grad_(True)
    feat = dev(synthetic_features(2, 128, 8, rank=11)).requires_grad_(True)
    feats, judges = disc(x, feat)
    assert len(feats[0]) == int(z["cond/nfeat"][0])
    loss = sum(j.mean() for j in judges) + 0.1 * sum(f.abs().mean() for grp in feats for f in grp)
    assert abs(loss.item() - float(z["cond/loss"][0])) <= 1e-4 * abs(float(z["cond/loss"][0]))
    for s in range(3):
        assert rel_l2(host(judges[s]), z["cond/j%d" % s]) < 1e-4
        for i, f in enumerate(feats[s]):
            assert tuple(f.shape) == tuple(z["cond/f%d_%d_shape" % (s, i)])
            assert rel_l2(strided_sample(host(f)), z["cond/f%d_%d_smp" % (s, i)]) < 1e-4
    loss.backward()
    assert rel_l2(host(feat.grad), z["cond/gfeat"]) < 1e-3
    assert rel_l2(strided_sample(host(x.grad), 1024), z["cond/gx_smp"]) < 1e-2     # (|.| features: sign flips at zero crossings)
    for k, p in disc.named_parameters():
        if "cond" in k or "layer_6" in k:
            s = z["cond/grad_sum/" + k]
            assert abs(float(np.linalg.norm(host(p.grad).astype(np.float64))) - s[0]) <= 1e-3 * s[0] + 1e-9, k


def test_conditioned_discriminator_through_trainers():
    """a discriminator that consumes its conditioning (conditioning_channels = 128) through
    featuresynth.train.  The native D-step runs ONE pass over [fake; real] (batch 2B), so the conditioning must be
    doubled with it; D-step and G-step gradients and losses are compared with the reference's order of operations
    (train/train.py:26-42,63-74) spelled out over the same modules."""
    from featuresynth import loss as LS
    from featuresynth._synthetic import (module_param_shapes, synthetic_features, synthetic_samples,
                                         synthetic_state_dict)
    from featuresynth.experiment import realmelgan as R
    from featuresynth.train import DiscriminatorTrainer, GeneratorTrainer
    from featuresynth.util.modules import zero_grad
    B, T = 2, 8

    def nets():
        g = R.Generator(128, 32, 3)
        d = R.Discriminator(3, 16, 4, 4, conditioning_channels=128)
        g.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(
            module_param_shapes(g), seed=41, weight_scale=0.1, bias_scale=0.05).items()})
        d.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(
            module_param_shapes(d), seed=45, weight_scale=0.3, bias_scale=0.05).items()})
        return g.cuda(), d.cuda()

    samples, feats = dev(synthetic_samples(B, T * 256, rank=5)), dev(synthetic_features(B, 128, T, rank=5))
    for kind in ("d", "g"):
        # reference order of operations
        g, d = nets()
        go = torch.optim.Adam(g.parameters(), lr=1e-4, betas=(0.5, 0.9))
        do = torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.9))
        zero_grad(go, do)
        fake = g(feats)
        f_feat, f_score = d(fake, feats)
        r_feat, r_score = d(samples, feats)
        if kind == "d":
            ref_loss = LS.mel_gan_disc_loss(r_score, f_score, gan_loss=LS.hinge_discriminator_loss)
        else:
            ref_loss = R.mel_gan_gen_loss(r_feat, f_feat, r_score, f_score, gan_loss=LS.hinge_generator_loss)
        ref_loss.backward()
        ref = {k: host(p.grad) for k, p in (d if kind == "d" else g).named_parameters()}
        # trainer (native path: batched [fake; real] pass in the D-step)
        g2, d2 = nets()
        go2 = torch.optim.Adam(g2.parameters(), lr=1e-4, betas=(0.5, 0.9))
        do2 = torch.optim.Adam(d2.parameters(), lr=1e-4, betas=(0.5, 0.9))
        if kind == "d":
            r = DiscriminatorTrainer(g2, go2, d2, do2, LS.mel_gan_disc_loss).train(samples, feats)
            assert abs(r["d_loss"] - ref_loss.item()) <= 1e-5 * abs(ref_loss.item())
        else:
            r = GeneratorTrainer(g2, go2, d2, do2, R.mel_gan_gen_loss).train(samples, feats)
            assert abs(r["g_loss"] - ref_loss.item()) <= 1e-5 * abs(ref_loss.item())
        worst = 0.0
        for k, p in (d2 if kind == "d" else g2).named_parameters():
            e = rel_l2(host(p.grad), ref[k])
            worst = max(worst, e)
            assert e < 1e-4, (kind, k, e)
        print("conditioned discriminator, %s-step: worst grad rel-L2 vs reference order %.2e" % (kind, worst))


def test_realmelgan_replay_gradients_bitwise(monkeypatch):
    """hipGraph capture and replay of the weight-normed variant's train steps reproduce the eager execution bitwise: with
    lr = 0 (parameters fixed) every one of D,G,D,G,D,G -- call 1 eager, call 2 captured, later calls replayed -- yields the
    same loss and the same flat gradient bucket as the eager run.  (r03 also ran the three discriminators on forked streams
    here; that variant was removed in r04, DESIGN_HISTORY.md section 4.)  Trainers are dropped between the runs: their graphs must
    go when they go (no reference cycle through the graphed step), not whenever the cycle collector fires."""
    import gc
    import weakref
    import featuresynth as fs
    from featuresynth import loss as LS
    from featuresynth._synthetic import synthetic_features, synthetic_samples
    from featuresynth.experiment import realmelgan as R
    from featuresynth.train import DiscriminatorTrainer, GeneratorTrainer
    s, f = dev(synthetic_samples(2, 1024, rank=1)), dev(synthetic_features(2, 128, 4, rank=1))
    dead = []

    def run(graph):
        monkeypatch.setenv("MSYNTH_GRAPH", graph)
        g, d, _, _ = _nets()
        go = fs.FlatAdam(g.parameters(), lr=0.0, betas=(0.5, 0.9))
        do = fs.FlatAdam(d.parameters(), lr=0.0, betas=(0.5, 0.9))
        dt = DiscriminatorTrainer(g, go, d, do, LS.mel_gan_disc_loss)
        gt = GeneratorTrainer(g, go, d, do, R.mel_gan_gen_loss)
        out = []
        for i in range(6):
            r = dt.train(s, f) if i % 2 == 0 else gt.train(s, f)
            torch.cuda.synchronize()
            opt = do if i % 2 == 0 else go
            out.append((r.get("d_loss", r.get("g_loss")), host(opt.flat_grads).copy()))
        if graph == "1":
            assert dt._runner.graphs and gt._runner.graphs and not gt._runner.disabled
        dead.append(weakref.ref(dt._runner))
        return out

    gc.disable()
    try:
        ref = run("0")
        got = run("1")
        assert all(r() is None for r in dead), "a dropped trainer's graphed step must be freed by reference counting alone"
    finally:
        gc.enable()
    for i in range(6):
        assert got[i][0] == ref[i][0], (i, got[i][0], ref[i][0])
        assert np.array_equal(got[i][1], ref[i][1]), (i, float(np.abs(got[i][1] - ref[i][1]).max()))
