"""GPU parity tests of the whole networks and the train step (run with -m gpu on an MI355X)."""
import os

import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def load(module, sd):
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return module


def make_nets(gkw, dkw, mels=80):
    import featuresynth as fs
    from featuresynth._synthetic import module_param_shapes, synthetic_state_dict
    g = fs.MelGanGenerator(32, mels)
    d = fs.MelGanDiscriminator()
    gsd = synthetic_state_dict(module_param_shapes(g), **gkw)
    dsd = synthetic_state_dict(module_param_shapes(d), **dkw)
    load(g, gsd); load(d, dsd)
    return g.cuda(), d.cuda(), gsd, dsd


def test_generator_forward_config2_golden(golden):
    """BASELINE config 2: generator output within 1e-4 rel-L2 of the reference (north star)."""
    z = golden("g_fwd")
    g, _, _, _ = make_nets(dict(seed=7), dict(seed=7))
    assert list(g.state_dict().keys()) == list(z["param_names"])
    feat = np.random.default_rng(1).standard_normal((1, 80, 32)).astype(np.float32)
    with torch.no_grad():
        y = g(dev(feat))
    assert tuple(y.shape) == (1, 1, 8192)
    e32, e64 = rel_l2(host(y), z["cfg2/y_ref32"]), rel_l2(host(y), z["cfg2/y_ref64"])
    print("generator cfg2 rel-L2 vs reference fp32 %.3e, fp64 %.3e" % (e32, e64))
    assert e32 < 1e-4 and e64 < 1e-4
    with torch.no_grad():
        y2 = g.forward_layerwise(dev(feat))      # reference-style iteration over self.main
    assert rel_l2(host(y2), host(y)) < 1e-6


def test_generator_forward_variants_golden(golden):
    import featuresynth as fs
    from featuresynth._synthetic import module_param_shapes, synthetic_state_dict
    z = golden("g_fwd")
    g = fs.MelGanGenerator(32, 80)
    load(g, synthetic_state_dict(module_param_shapes(g), seed=11, weight_scale=0.05, bias_scale=0.05)).cuda()
    feat2 = np.random.default_rng(2).standard_normal((2, 80, 5)).astype(np.float32)
    with torch.no_grad():
        assert rel_l2(host(g(dev(feat2))), z["short/y_ref32"]) < 1e-4
    g128 = fs.MelGanGenerator(32, 128)
    load(g128, synthetic_state_dict(module_param_shapes(g128), seed=7, bias_scale=0.02)).cuda()
    feat3 = np.random.default_rng(3).standard_normal((1, 128, 7)).astype(np.float32)
    with torch.no_grad():
        assert rel_l2(host(g128(dev(feat3))), z["mel128/y_ref32"]) < 1e-4
    with pytest.raises(RuntimeError):
        g128(dev(feat2))                          # wrong mel count


def test_discriminator_forward_golden(golden):
    import featuresynth as fs
    from featuresynth._synthetic import (module_param_shapes, strided_sample, synthetic_samples,
                                         synthetic_state_dict)
    z = golden("d_fwd")
    d = fs.MelGanDiscriminator()
    assert list(d.state_dict().keys()) == list(z["param_names"])
    for tag, kw, x in (("cfg", dict(seed=7), synthetic_samples(1)),
                       ("big", dict(seed=13, weight_scale=0.08, bias_scale=0.1),
                        synthetic_samples(2, 3000, rank=5))):
        load(d, synthetic_state_dict(module_param_shapes(d), **kw)).cuda()
        with torch.no_grad():
            feats, judges = d(dev(x))
            feats2, judges2 = d(dev(x), None)      # trainer-style 2-arg call
        assert len(feats) == 3 and all(len(f) == 6 for f in feats) and len(judges) == 3
        for s in range(3):
            assert tuple(judges[s].shape) == z["%s/j%d_ref32" % (tag, s)].shape
            assert rel_l2(host(judges[s]), z["%s/j%d_ref32" % (tag, s)]) < 1e-4
            assert torch.equal(judges[s], judges2[s])
            for i in range(6):
                assert tuple(feats[s][i].shape) == tuple(z["%s/f%d_%d_shape" % (tag, s, i)])
                assert rel_l2(strided_sample(host(feats[s][i])), z["%s/f%d_%d_smp_ref32" % (tag, s, i)]) < 1e-4
                nrm = float(torch.linalg.vector_norm(feats[s][i].double()))
                assert abs(nrm - z["%s/f%d_%d_sum_ref32" % (tag, s, i)][0]) < 1e-4 * nrm
    # single-scale module surface
    fd = d.disc
    with torch.no_grad():
        f1, j1 = fd(dev(synthetic_samples(2, 3000, rank=5)))
    # (the one-scale module runs the per-scale kernels, the three-scale one the parts kernels: summation order)
    assert len(f1) == 6 and rel_l2(host(j1), host(judges[0])) < 1e-5


def _trainers(g, d, optim_kind="flat"):
    import featuresynth as fs
    from featuresynth import loss as LS
    from featuresynth.train import DiscriminatorTrainer, GeneratorTrainer
    if optim_kind == "flat":
        go = fs.FlatAdam(g.parameters(), lr=1e-4, betas=(0.5, 0.9))
        do = fs.FlatAdam(d.parameters(), lr=1e-4, betas=(0.5, 0.9))
    else:
        go = torch.optim.Adam(g.parameters(), lr=1e-4, betas=(0.5, 0.9))
        do = torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.9))
    return (DiscriminatorTrainer(g, go, d, do, LS.mel_gan_disc_loss),
            GeneratorTrainer(g, go, d, do, LS.mel_gan_gen_loss), go, do)


def _masked_oracle_step(kind, gsd, dsd, samples, feats, dbg, dtype=None):
    """The step in float64 on the CPU (oracle/torch_graph.py = the reference's graph as stock torch ops),
    with every LeakyReLU's BACKWARD taking the branch the device took (masks read back from the
    activations the device saved for its own backward): a pre-activation within rounding of zero no longer
    shows up as an O(1) gradient difference, so every parameter gradient can be held to SURVEY 8(d)'s
    1e-3 (reference: train/train.py:26-42,63-74).  -> (loss, {name: grad}, params after Adam, fake)"""
    import torch as th
    from oracle import torch_graph as TG
    dtype = dtype or th.float64
    to_bool = lambda t: (t.detach().cpu() > 0)
    gp, dp = TG.to_params(gsd, dtype=dtype), TG.to_params(dsd, dtype=dtype)
    s, f = th.from_numpy(samples).to(dtype), th.from_numpy(feats).to(dtype)
    B = samples.shape[0]
    if kind == "d":
        with th.no_grad():
            fake = TG.generator(gp, f)
        # the device ran ONE discriminator pass over [fake; real]
        mf = TG.discriminator_masks_from_ctx(dbg["disc_ctx"], to_bool, rows=slice(0, B))
        mr = TG.discriminator_masks_from_ctx(dbg["disc_ctx"], to_bool, rows=slice(B, 2 * B))
        _, fj = TG.discriminator(dp, fake, masks=mf)
        _, rj = TG.discriminator(dp, s, masks=mr)
        loss, net = TG.disc_loss(rj, fj), dp
    else:
        fake = TG.generator(gp, f, masks=TG.generator_masks_from_tape(dbg["gen_tape"], to_bool))
        # (the device may have run ONE pass over [fake; real]: disc_rows then names the fake half of the saved activations)
        ff, fj = TG.discriminator(dp, fake, masks=TG.discriminator_masks_from_ctx(dbg["disc_ctx"], to_bool,
                                                                                   rows=dbg.get("disc_rows")))
        with th.no_grad():
            rf, _ = TG.discriminator(dp, s)
        loss, net = TG.gen_loss(rf, ff, fj), gp
    loss.backward()
    grads = {k: v.grad.numpy().copy() for k, v in net.items()}
    opt = th.optim.Adam(list(net.values()), lr=1e-4, betas=(0.5, 0.9))
    opt.step()
    return loss.item(), grads, {k: v.detach().numpy() for k, v in net.items()}, fake.detach().numpy()


def _check_grads(kind, net, o_grads, max_tol=1e-3, median_tol=1e-4):
    """every parameter gradient within 1e-3 rel-L2 (typical layer 1e-4).  A gradient that
    cancels to rounding level (hinge gradient of the judge bias at init: sum(+1/n) - sum(1/n)) has no
    meaningful relative error and is held to an absolute bound instead: 1e-5 of the largest gradient, or
    1e-6 where that is smaller (the cancelling terms sum to O(1) on either side: half an fp32 ulp of them)."""
    gmax = max(float(np.linalg.norm(r)) for r in o_grads.values())
    errs = {}
    for k, p in net.named_parameters():
        r = o_grads[k]
        if np.linalg.norm(r) > 1e-5 * gmax:
            errs[k] = rel_l2(host(p.grad), r)
        else:
            assert float(np.abs(host(p.grad) - r).max()) < max(1e-5 * gmax, 1e-6), (kind, k)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    assert worst[0][1] < max_tol, (kind, worst)
    assert float(np.median(list(errs.values()))) < median_tol, (kind, worst)
    return worst[0][1], float(np.median(list(errs.values())))


@pytest.mark.parametrize("B,T,mels", [(2, 8, 80), (3, 5, 128)], ids=["b2_t8_mel80", "b3_t5_mel128"])
def test_train_steps_vs_oracle(B, T, mels, monkeypatch):
    """One D-step and one G-step against the flip-aware float64 oracle: loss, every parameter gradient
    (<= 1e-3 rel-L2, median <= 1e-4), the parameters after the Adam update, and the trainer return dicts.
    Then the same steps with stock torch.optim.Adam objects (autograd path, no flat bucket): same numbers."""
    from featuresynth._synthetic import synthetic_features, synthetic_samples
    monkeypatch.setenv("MSYNTH_GRAPH", "0")
    samples, feats = synthetic_samples(B, T * 256, rank=B), synthetic_features(B, mels, T, rank=B)
    for kind in ("d", "g"):
        g, d, gsd, dsd = make_nets(dict(seed=7, bias_scale=0.02), dict(seed=8, bias_scale=0.02), mels=mels)
        dt, gt, _, _ = _trainers(g, d)
        tr, net = (dt, d) if kind == "d" else (gt, g)
        tr.debug = {}
        res = tr.train(dev(samples), dev(feats))
        assert tr.debug, "hand-scheduled step not taken"
        o_loss, o_grads, o_params, o_fake = _masked_oracle_step(kind, gsd, dsd, samples, feats, tr.debug)
        tr.debug = None
        if kind == "d":
            assert set(res) == {"d_loss"} and isinstance(res["d_loss"], float)
            loss = res["d_loss"]
        else:
            assert set(res) == {"g_loss", "fake"} and isinstance(res["fake"], np.ndarray)
            assert res["fake"].shape == (B, 1, T * 256)
            assert rel_l2(res["fake"], o_fake) < 1e-4
            loss = res["g_loss"]
        assert abs(loss - o_loss) <= 1e-4 * abs(o_loss), (kind, loss, o_loss)
        worst, med = _check_grads(kind, net, o_grads)
        # Adam: |update| <= lr whatever the gradient; where the oracle's gradient is clearly above the
        # rounding floor the updates agree closely
        for k, p in net.named_parameters():
            diff = np.abs(host(p) - o_params[k])
            assert diff.max() <= 2.1e-4, (kind, k, diff.max())
            big = np.abs(o_grads[k]) > 1e-3 * np.abs(o_grads[k]).max()
            if big.any():
                assert diff[big].max() < 2e-5, (kind, k, diff[big].max())
        flat_grads = {k: host(p.grad).copy() for k, p in net.named_parameters()}
        print("%s-step B=%d: loss %.8f (oracle %.8f), grad rel-L2 max %.2e median %.2e" % (kind, B, loss, o_loss, worst, med))
        # stock torch optimizers: the autograd-Function path over the same kernels
        g2, d2, _, _ = make_nets(dict(seed=7, bias_scale=0.02), dict(seed=8, bias_scale=0.02), mels=mels)
        dt2, gt2, _, _ = _trainers(g2, d2, "torch")
        tr2, net2 = (dt2, d2) if kind == "d" else (gt2, g2)
        res2 = tr2.train(dev(samples), dev(feats))
        assert abs(res2["d_loss" if kind == "d" else "g_loss"] - loss) <= 1e-6 * abs(loss)
        for k, p in net2.named_parameters():
            assert rel_l2(host(p.grad), flat_grads[k]) < 1e-5 or np.linalg.norm(flat_grads[k]) < 1e-12, (kind, k)


@pytest.mark.parametrize("tag", ["small", "cfg3"])
@pytest.mark.parametrize("optim_kind", ["flat", "torch"])
def test_train_steps_vs_reference_fixture(golden, tag, optim_kind):
    """The HIP trainers against what the REFERENCE's own trainers produced (tests/golden/train.npz, written by
    tools/make_golden.py from /root/reference/featuresynth/train/train.py:26-42,63-74 with stock torch.optim.Adam):
    D-step on fresh parameters (loss, every discriminator gradient), then the G-step behind that D update (loss, `fake`,
    every generator gradient) -- directly, not through the oracle.  `small`: B = 2, 2048-sample windows, D,G,D,G;
    `cfg3`: B = 1 at BASELINE config 3's 8192-sample window, D,G.  Tolerances are those the CPU oracle is held to against
    the same fixture (tests/test_oracle_golden.py::test_train_steps_small): the reference here ran in float32, and its own
    float32 / float64 runs differ by more than this from the second D-step on (DESIGN_HISTORY.md "Adam sensitivity")."""
    from featuresynth._synthetic import strided_sample, synthetic_features, synthetic_samples
    z = golden("train")
    B, T, nsteps = [int(v) for v in z[tag + "/cfg"]]
    gkw, dkw = ((dict(seed=7, bias_scale=0.02), dict(seed=8, bias_scale=0.02)) if tag == "small"
                else (dict(seed=7), dict(seed=7)))
    g, d, _, _ = make_nets(gkw, dkw)
    dt, gt, _, _ = _trainers(g, d, optim_kind)
    ref_losses = z[tag + "/losses"]
    losses = []

    def check_grads(net, kind, tol):
        worst = 0.0
        for k, p in net.named_parameters():
            smp, ref = strided_sample(host(p.grad)), z["%s/%sgrad_smp/%s" % (tag, kind, k)]
            nrm, rsum = float(torch.linalg.vector_norm(p.grad.double())), z["%s/%sgrad_sum/%s" % (tag, kind, k)]
            scale = float(np.linalg.norm(ref.astype(np.float64)))
            # a gradient that cancels to rounding level (the hinge gradient of the judge bias at init: fake and real terms of
            # opposite sign) has no meaningful relative error: absolute bound at the size of the tensor's other entries
            if rsum[0] < 1e-9:
                assert nrm < 1e-6, (k, nrm)
                continue
            e = rel_l2(smp, ref)
            ntol = tol
            if e >= tol and k.endswith("bias"):
                # ONE LeakyReLU branch taken the other way (a pre-activation within rounding of zero: cfg3 has zero biases)
                # shows up as an O(5 %) difference in single bias-gradient entries along that element's backward cone through
                # the grouped layers (seen: main.3 channel 728 -> main.2 180..183 -> main.1 44..47 -> main.0 8..11, every other
                # entry at 1e-9) while the weight gradients stay at 1e-6.  Hold everything but the 4 worst entries to `tol`;
                # the masked comparison of test_train_steps_vs_oracle covers the flipped element itself.
                dlt = np.abs(smp.astype(np.float64) - ref)
                keep = np.ones(smp.size, bool); keep[np.argsort(dlt)[-4:]] = False
                assert e < 10 * tol, (tag, kind, k, e)
                e, ntol = rel_l2(smp[keep], ref[keep]), 10 * tol
            worst = max(worst, e)
            assert e < tol, (tag, kind, k, e)
            assert abs(nrm - rsum[0]) <= ntol * rsum[0] + 1e-12, (tag, kind, k, nrm, rsum[0])
            assert scale > 0
        return worst

    for step in range(nsteps):
        samples, feats = synthetic_samples(B, T * 256, rank=step), synthetic_features(B, 80, T, rank=step)
        if step % 2 == 0:
            r = dt.train(dev(samples), dev(feats))
            assert set(r) == {"d_loss"}
            losses.append(r["d_loss"])
            if step == 0:
                w = check_grads(d, "d", 2e-3)
                print("%s D-step: loss %.8f (reference %.8f), worst gradient rel-L2 %.2e" % (tag, r["d_loss"], ref_losses[0], w))
        else:
            r = gt.train(dev(samples), dev(feats))
            assert set(r) == {"g_loss", "fake"} and r["fake"].shape == (B, 1, T * 256)
            losses.append(r["g_loss"])
            if step == 1:
                assert rel_l2(strided_sample(r["fake"]), z[tag + "/fake_smp"]) < 1e-4
                nrm = float(np.linalg.norm(r["fake"].astype(np.float64)))
                assert abs(nrm - z[tag + "/fake_sum"][0]) < 1e-4 * nrm
                w = check_grads(g, "g", 5e-3)
                print("%s G-step: loss %.8f (reference %.8f), worst gradient rel-L2 %.2e" % (tag, r["g_loss"], ref_losses[1], w))
    assert abs(losses[0] - ref_losses[0]) <= 1e-5 * abs(ref_losses[0]), (losses, ref_losses)
    assert abs(losses[1] - ref_losses[1]) <= 5e-3 * abs(ref_losses[1]), (losses, ref_losses)
    for a, b in zip(losses[2:], ref_losses[2:]):
        assert abs(a - b) <= 0.2 * abs(b), (losses, ref_losses)
    # parameters after the run: every Adam step moves an entry by at most ~lr; weights agree closely
    lr = 1e-4
    for net, kind in ((g, "g"), (d, "d")):
        for k, v in net.state_dict().items():
            got, ref = strided_sample(host(v)), z["%s/%sparam_smp/%s" % (tag, kind, k)]
            assert np.abs(got - ref).max() <= nsteps * lr + 1e-6, (tag, k)
            if k.endswith("weight"):
                assert rel_l2(got, ref) < 5e-3, (tag, k, rel_l2(got, ref))


def test_reference_order_path_matches_native(monkeypatch):
    """The generic (reference order of operations) trainer path gives the same D/G updates as the
    native path that skips the discarded work."""
    import featuresynth as fs
    from featuresynth import loss as LS
    from featuresynth._synthetic import synthetic_features, synthetic_samples
    from featuresynth.train import DiscriminatorTrainer, GeneratorTrainer
    monkeypatch.setenv("MSYNTH_GRAPH", "0")
    B, T = 2, 4
    samples, feats = dev(synthetic_samples(B, T * 256)), dev(synthetic_features(B, 80, T))
    results = {}
    for mode in ("native", "generic"):
        g, d, _, _ = make_nets(dict(seed=7, bias_scale=0.02), dict(seed=8, bias_scale=0.02))
        if mode == "generic":
            g._ms_native = False                      # force the reference-order branch
        go = torch.optim.Adam(g.parameters(), lr=1e-4, betas=(0.5, 0.9))
        do = torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.9))
        dt = DiscriminatorTrainer(g, go, d, do, LS.mel_gan_disc_loss)
        gt = GeneratorTrainer(g, go, d, do, LS.mel_gan_gen_loss)
        r1 = dt.train(samples, feats)
        r2 = gt.train(samples, feats)
        results[mode] = (r1["d_loss"], r2["g_loss"], {k: host(v) for k, v in g.state_dict().items()},
                         {k: host(v) for k, v in d.state_dict().items()})
    a, b = results["native"], results["generic"]
    # the D-step runs on identical parameters: same loss.  The native path sums the shared
    # discriminator's weight grads in a different order (one pass over [fake; real]); after the Adam
    # update, rounding-level gradient entries may step the other way (DESIGN_HISTORY.md "Adam sensitivity"),
    # so post-update quantities are compared at the +-lr level.
    assert abs(a[0] - b[0]) < 1e-6
    assert abs(a[1] - b[1]) < 1e-2 * abs(b[1])
    for k in a[2]:
        assert np.abs(a[2][k] - b[2][k]).max() <= 2.1e-4, k
    for k in a[3]:
        d = np.abs(a[3][k] - b[3][k])
        assert d.max() <= 2.1e-4, k
        if k.endswith("weight"):
            assert rel_l2(a[3][k], b[3][k]) < 5e-3, k


def test_hipgraph_replay_matches_eager(monkeypatch):
    """D,G,D,G,D,G with FlatAdam: the captured-graph path and the eager path walk the same trajectory."""
    import featuresynth as fs
    from featuresynth import loss as LS
    from featuresynth._synthetic import synthetic_features, synthetic_samples
    from featuresynth.train import DiscriminatorTrainer, GeneratorTrainer
    B, T = 2, 4
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MSYNTH_GRAPH", mode)
        g, d, _, _ = make_nets(dict(seed=7, bias_scale=0.02), dict(seed=8, bias_scale=0.02))
        go = fs.FlatAdam(g.parameters(), lr=1e-4, betas=(0.5, 0.9))
        do = fs.FlatAdam(d.parameters(), lr=1e-4, betas=(0.5, 0.9))
        dt = DiscriminatorTrainer(g, go, d, do, LS.mel_gan_disc_loss)
        gt = GeneratorTrainer(g, go, d, do, LS.mel_gan_gen_loss)
        losses = []
        for step in range(6):
            s = dev(synthetic_samples(B, T * 256, rank=step))
            f = dev(synthetic_features(B, 80, T, rank=step))
            losses.append(dt.train(s, f)["d_loss"] if step % 2 == 0 else gt.train(s, f)["g_loss"])
        if mode == "1":
            assert dt._runner.graphs and gt._runner.graphs and not dt._runner.disabled, "graph not captured"
        assert go.step_count() == 3 and do.step_count() == 3
        out[mode] = (losses, {k: host(v) for k, v in g.state_dict().items()})
    for a, b in zip(out["0"][0], out["1"][0]):
        assert abs(a - b) <= 1e-5 * abs(a) + 1e-7, (out["0"][0], out["1"][0])
    for k in out["0"][1]:
        assert np.abs(out["0"][1][k] - out["1"][1][k]).max() < 1e-6, k


def test_checkpoint_roundtrip(tmp_path):
    """state_dicts written as at reference experiment/experiment.py:171-173 round-trip."""
    import featuresynth as fs
    g, d, gsd, dsd = make_nets(dict(seed=7), dict(seed=7))
    torch.save(g.state_dict(), str(tmp_path / "gen.dat"))
    torch.save(d.state_dict(), str(tmp_path / "disc.dat"))
    g2, d2 = fs.MelGanGenerator(32, 80), fs.MelGanDiscriminator()
    g2.load_state_dict(torch.load(str(tmp_path / "gen.dat")))
    d2.load_state_dict(torch.load(str(tmp_path / "disc.dat")))
    for k, v in g2.state_dict().items():
        assert np.array_equal(host(v), gsd[k])
    for k, v in d2.state_dict().items():
        assert np.array_equal(host(v), dsd[k])


def test_full_size_properties():
    """BASELINE config 3 sizes (B=32, 8192 samples): size-independent properties -- batch
    independence (no cross-sample coupling), shape contract, finite outputs, d_loss ~ 6 at init
    (3 scales x hinge 2.0)."""
    import featuresynth as fs
    from featuresynth import loss as LS
    from featuresynth._synthetic import synthetic_features, synthetic_samples
    g, d, _, _ = make_nets(dict(seed=7), dict(seed=7))
    feats, samples = dev(synthetic_features(32)), dev(synthetic_samples(32))
    with torch.no_grad():
        fake = g(feats)
        assert tuple(fake.shape) == (32, 1, 8192) and torch.isfinite(fake).all()
        sub = g(feats[5:7].contiguous())
        # (the split-K factor, and with it the fp32 summation order, depends on the batch size)
        assert rel_l2(host(sub), host(fake[5:7])) < 1e-5
        f_all, j_all = d(samples)
        f_sub, j_sub = d(samples[30:32].contiguous())
        assert [tuple(j.shape) for j in j_all] == [(32, 1, 32), (32, 1, 17), (32, 1, 9)]
        for s in range(3):
            assert rel_l2(host(j_sub[s]), host(j_all[s][30:32])) < 1e-5
        _, fj = d(fake)
        dl = LS.mel_gan_disc_loss(j_all, fj)
        assert abs(dl.item() - 6.0) < 1e-2


@pytest.mark.parametrize("B,T", [(1, 4), (3, 5), (2, 128)])
def test_generator_lengths_vs_oracle(B, T):
    """Fully-convolutional generator at the shortest input the reflection pad allows, an odd
    batch/length, and the 4x inference length (experiment/experiment.py:223-229)."""
    import torch as th
    from featuresynth._synthetic import synthetic_features
    from oracle import torch_graph as TG
    g, _, gsd, _ = make_nets(dict(seed=7, weight_scale=0.05, bias_scale=0.05), dict(seed=7))
    feats = synthetic_features(B, 80, T, rank=T)
    with th.no_grad():
        y = g(dev(feats))
        ref = TG.generator(TG.to_params(gsd, False), th.from_numpy(feats)).numpy()
    assert tuple(y.shape) == (B, 1, 256 * T)
    assert rel_l2(host(y), ref) < 1e-4


@pytest.mark.parametrize("B,L", [(1, 64), (3, 300), (2, 1000)])
def test_discriminator_short_and_odd_inputs_vs_oracle(B, L):
    """Very short / odd windows: pooled scales go down to a handful of samples (K=41 > L)."""
    import torch as th
    from featuresynth._synthetic import synthetic_samples
    from oracle import torch_graph as TG
    _, d, _, dsd = make_nets(dict(seed=7), dict(seed=13, weight_scale=0.08, bias_scale=0.1))
    x = synthetic_samples(B, L, rank=L)
    with th.no_grad():
        feats, judges = d(dev(x))
        rf, rj = TG.discriminator(TG.to_params(dsd, False), th.from_numpy(x))
    for s in range(3):
        assert tuple(judges[s].shape) == tuple(rj[s].shape)
        assert rel_l2(host(judges[s]), rj[s].numpy()) < 1e-4
        for i in range(6):
            assert tuple(feats[s][i].shape) == tuple(rf[s][i].shape)
            assert rel_l2(host(feats[s][i]), rf[s][i].numpy()) < 1e-4


def test_full_size_train_step_vs_oracle():
    """BASELINE config 3 sizes (B=32, 8192-sample windows, 80 mels): one D-step and one G-step -- the
    kernels, tile shapes and split-K plans the benchmark runs -- against the flip-aware float64 oracle:
    loss 1e-4, every parameter gradient 1e-3 rel-L2, median 1e-4."""
    import torch as th
    from featuresynth._synthetic import synthetic_features, synthetic_samples
    B, T = 32, 32
    samples, feats = synthetic_samples(B, T * 256, rank=2), synthetic_features(B, 80, T, rank=2)
    th.set_num_threads(16)
    for kind in ("d", "g"):
        g, d, gsd, dsd = make_nets(dict(seed=7, bias_scale=0.02), dict(seed=8, bias_scale=0.02))
        dt, gt, _, _ = _trainers(g, d)
        tr, net = (dt, d) if kind == "d" else (gt, g)
        tr.debug = {}
        res = tr.train(dev(samples), dev(feats))        # first call: eager
        o_loss, o_grads, _, o_fake = _masked_oracle_step(kind, gsd, dsd, samples, feats, tr.debug)
        tr.debug = None
        loss = res["d_loss" if kind == "d" else "g_loss"]
        assert abs(loss - o_loss) <= 1e-4 * abs(o_loss), (kind, loss, o_loss)
        if kind == "g":
            assert rel_l2(res["fake"], o_fake) < 1e-4
        worst, med = _check_grads(kind, net, o_grads)
        print("%s-step full size: loss %.6f (oracle %.6f), grad rel-L2 max %.1e median %.1e" % (
            kind, loss, o_loss, worst, med))


def test_full_size_graph_replay_matches_eager(monkeypatch):
    """The mode bench.py times -- hipGraph replay with forked streams at B=32 -- against the eager
    execution of the same calls: D,G,D,G,D,G on different batches (calls 3+ replay).  Same kernels in the
    same per-stream order, so losses, both gradient buckets and every updated parameter must agree
    BITWISE.  The stream-serialised schedule (MSYNTH_STREAMS=0) adds the three scales' weight gradients
    in another order: its first D-step and G-step gradients are held to 1e-6 rel-L2 of the forked ones."""
    from featuresynth._synthetic import synthetic_features, synthetic_samples
    B, T = 32, 32
    out = {}
    for mode, env, ncalls in (("eager", {"MSYNTH_GRAPH": "0", "MSYNTH_STREAMS": "1"}, 6),
                              ("replay", {"MSYNTH_GRAPH": "1", "MSYNTH_STREAMS": "1"}, 6),
                              ("serial", {"MSYNTH_GRAPH": "0", "MSYNTH_STREAMS": "0"}, 2)):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g, d, _, _ = make_nets(dict(seed=7, bias_scale=0.02), dict(seed=8, bias_scale=0.02))
        dt, gt, go, do = _trainers(g, d)
        losses, first = [], {}
        for step in range(ncalls):
            s = dev(synthetic_samples(B, T * 256, rank=step))
            f = dev(synthetic_features(B, 80, T, rank=step))
            losses.append(dt.train(s, f)["d_loss"] if step % 2 == 0 else gt.train(s, f)["g_loss"])
            if step == 0:
                first["d"] = host(do.flat_grads).copy()
            if step == 1:
                first["g"] = host(go.flat_grads).copy()
        if mode == "replay":
            assert dt._runner.graphs and gt._runner.graphs and not dt._runner.disabled and not gt._runner.disabled
        out[mode] = (losses, {k: host(v) for k, v in list(g.state_dict().items()) + list(d.state_dict().items())},
                     host(go.flat_grads), host(do.flat_grads), first)
    assert out["eager"][0] == out["replay"][0], (out["eager"][0], out["replay"][0])
    for k in out["eager"][1]:
        assert np.array_equal(out["eager"][1][k], out["replay"][1][k]), k
    assert np.array_equal(out["eager"][2], out["replay"][2]) and np.array_equal(out["eager"][3], out["replay"][3])
    assert abs(out["serial"][0][0] - out["eager"][0][0]) <= 1e-6 * abs(out["eager"][0][0])
    assert rel_l2(out["serial"][4]["d"], out["eager"][4]["d"]) < 1e-6
    # (the G-step follows a D update whose rounding-level entries may step the other way: DESIGN.md
    #  "Adam sensitivity"; its gradient is compared at the level that leaves)
    assert rel_l2(out["serial"][4]["g"], out["eager"][4]["g"]) < 1e-3


def test_flat_adam_with_module_zero_grad():
    """callers that clear gradients with net.zero_grad() (torch sets .grad = None) instead of
    optim.zero_grad() -- evaluate_pair.py:193 in the reference -- must still step on the real gradient,
    and a generic optimizer checkpoint must carry the moments."""
    import featuresynth as fs
    from featuresynth import loss as LS
    from featuresynth._synthetic import synthetic_features, synthetic_samples
    B, T = 2, 4
    s, f = dev(synthetic_samples(B, T * 256)), dev(synthetic_features(B, 80, T))
    res = {}
    for kind in ("flat", "torch"):
        g, d, _, _ = make_nets(dict(seed=7, bias_scale=0.02), dict(seed=8, bias_scale=0.02))
        opt = (fs.FlatAdam if kind == "flat" else torch.optim.Adam)(d.parameters(), lr=1e-4, betas=(0.5, 0.9))
        with torch.no_grad():
            fake = g(f)

        def backward():
            _, fj = d(fake)
            _, rj = d(s)
            loss = LS.mel_gan_disc_loss(rj, fj)
            loss.backward()
            return loss.item()
        opt.zero_grad()
        backward()                                     # flat: the bucket is bound and now holds this gradient
        d.zero_grad()                                  # NOT opt.zero_grad(): .grad = None, the bucket is stale
        loss = backward()
        grads = {k: host(p.grad).copy() for k, p in d.named_parameters()}
        opt.step()
        if kind == "flat":                             # the step consumed THIS gradient, not stale + new
            for (k, p), gv in zip(d.named_parameters(), opt.grad_views()):
                assert np.array_equal(host(gv), grads[k]), k
        res[kind] = (loss, grads, {k: host(v) for k, v in d.state_dict().items()}, opt)
    assert abs(res["flat"][0] - res["torch"][0]) <= 1e-6 * abs(res["torch"][0])
    for k, gt_ in res["torch"][1].items():
        assert rel_l2(res["flat"][1][k], gt_) < 1e-5 or np.linalg.norm(gt_) < 1e-12, k
    for k, v in res["torch"][2].items():
        assert np.abs(res["flat"][2][k] - v).max() <= 2.1e-4, k
        big = np.abs(res["torch"][1][k]) > 1e-3 * np.abs(res["torch"][1][k]).max()
        if big.any():
            assert np.abs(res["flat"][2][k] - v)[big].max() < 2e-5, k
    # optimizer checkpoint through the generic interface
    sd = res["flat"][3].state_dict()
    assert "flat" in sd and int(sd["flat"]["step"]) == 1 and float(sd["flat"]["exp_avg"].abs().sum()) > 0
    g, d, _, _ = make_nets(dict(seed=7, bias_scale=0.02), dict(seed=8, bias_scale=0.02))
    opt2 = fs.FlatAdam(d.parameters(), lr=1e-4, betas=(0.5, 0.9))
    opt2.load_state_dict(sd)
    assert opt2.step_count() == 1 and torch.equal(opt2._flat[2].cpu(), sd["flat"]["exp_avg"])


def test_generator_feature_gradient_vs_oracle():
    """ADVICE r1 / BASELINE config 5: d loss / d mel features through the reflection-padded first conv
    (a stage-1 feature generator in front of the vocoder needs it), with and without parameter grads."""
    import torch as th
    from featuresynth._synthetic import synthetic_features
    from oracle import torch_graph as TG
    g, _, gsd, _ = make_nets(dict(seed=7, bias_scale=0.02), dict(seed=8))
    feats = synthetic_features(2, 80, 6, rank=3)
    gy = np.random.default_rng(5).standard_normal((2, 1, 6 * 256)).astype(np.float32)
    gp = TG.to_params(gsd, dtype=th.float64)
    xr = th.from_numpy(feats).double().requires_grad_(True)
    TG.generator(gp, xr).backward(th.from_numpy(gy).double())
    for frozen in (False, True):
        for p in g.parameters():
            p.requires_grad_(not frozen)
            p.grad = None
        x = dev(feats).requires_grad_(True)
        g(x).backward(dev(gy))
        assert x.grad is not None and rel_l2(host(x.grad), xr.grad.numpy()) < 1e-3, frozen
        if not frozen:
            assert rel_l2(host(g.main[1].weight.grad), gp["main.1.weight"].grad.numpy()) < 1e-3
    for p in g.parameters():
        p.requires_grad_(True)


def test_discriminator_captured_on_a_side_stream():
    """The fork guard (graph.forked / _may_fork): a discriminator pass issued while the current stream is
    already a forked one must not fork again under hipGraph capture (nested forks crashed graph
    instantiation on ROCm 7.2), and gives the same numbers."""
    from featuresynth._ops import graph as G
    from featuresynth._synthetic import synthetic_samples
    _, d, _, _ = make_nets(dict(seed=7), dict(seed=8, bias_scale=0.02))
    x = dev(synthetic_samples(2, 2048))
    with torch.no_grad():
        ref = d(x)[1]
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with G.forked(side):
                assert not G._may_fork(x.device)
                out = d(x)[1]
            main.wait_stream(side)
        gr.replay()
        torch.cuda.synchronize()
    for a, b in zip(ref, out):
        assert torch.equal(a, b)


def test_training_loop_checkpoint_resume(tmp_path, monkeypatch):
    """featuresynth.train.training_loop over an Experiment: alternating D/G
    steps, logger plumbing, checkpoint + resume including the optimizer state."""
    import torch as th
    import featuresynth.experiment as E
    from featuresynth.train import training_loop
    monkeypatch.chdir(tmp_path)
    device = th.device("cuda", 0)
    th.manual_seed(0)
    exp = E.MultiScaleMelGanExperiment(n_mels=80).to(device)
    seen = []

    def logger(experiment, batch, result, i, elapsed):
        seen.append(sorted(result))
        return {"iter": i}

    logs = list(training_loop(exp.synthetic_batch_stream(2, n_batches=6), exp, device, [logger]))
    assert [l[0] for l in logs] == list(range(6)) and logs[-1][2] == {"iter": 5}
    assert seen == [["d_loss"], ["fake", "g_loss"]] * 3
    exp.checkpoint("ck_", with_optimizers=True)
    th.manual_seed(1)
    exp2 = E.MultiScaleMelGanExperiment(n_mels=80).to(device)
    exp2.resume("ck_", with_optimizers=True)
    assert exp2._d_optim.step_count() == 3 and exp2._g_optim.step_count() == 3
    s, f = next(exp.synthetic_batch_stream(2))
    s, f = dev(s), dev(f)
    a = exp.discriminator_trainer(s, f)["d_loss"]
    b = exp2.discriminator_trainer(s, f)["d_loss"]
    assert abs(a - b) < 1e-6
    for (k, p), (_, q) in zip(exp.discriminator.state_dict().items(), exp2.discriminator.state_dict().items()):
        assert np.abs(host(p) - host(q)).max() < 1e-7, k


def test_flat_adam_state_reload_keeps_captured_graphs_valid(monkeypatch):
    """FlatAdam.load_state_dict after the first captured step.  The captured hipGraphs hold the
    bucket addresses, so a reload must not re-home the buckets (and if one moves, the trainers re-plan):
    3 calls, save + load both optimizer states and both modules, 3 more calls == 6 uninterrupted calls, bitwise."""
    from featuresynth._synthetic import synthetic_features, synthetic_samples
    B, T = 2, 4
    batches = [(dev(synthetic_samples(B, T * 256, rank=i)), dev(synthetic_features(B, 80, T, rank=i))) for i in range(6)]

    def run(reload_at):
        g, d, _, _ = make_nets(dict(seed=7, bias_scale=0.02), dict(seed=8, bias_scale=0.02))
        dt, gt, go, do = _trainers(g, d)
        losses = []
        for i, (s, f) in enumerate(batches):
            if i == reload_at:
                ptrs = (go.flat_params.data_ptr(), do.flat_params.data_ptr())
                st = (go.state_dict(), do.state_dict(), {k: v.clone() for k, v in g.state_dict().items()},
                      {k: v.clone() for k, v in d.state_dict().items()})
                go.load_state_dict(st[0]); do.load_state_dict(st[1])
                g.load_state_dict(st[2]); d.load_state_dict(st[3])
                assert (go.flat_params.data_ptr(), do.flat_params.data_ptr()) == ptrs, "intact buckets must stay put"
            losses.append(dt.train(s, f)["d_loss"] if i % 2 == 0 else gt.train(s, f)["g_loss"])
        assert dt._runner.graphs and gt._runner.graphs and dt.graph_status()["mode"] == "graph"
        return losses, {k: host(v) for k, v in list(g.state_dict().items()) + list(d.state_dict().items())}

    l0, sd0 = run(None)
    l1, sd1 = run(4)                 # both trainers have captured by call 4 (calls 3 and 4 capture); 5-6 replay
    assert l0 == l1, (l0, l1)
    for k in sd0:
        assert np.array_equal(sd0[k], sd1[k]), k
    # a bucket that DOES move (flatten() called by hand) makes the trainers drop their plan and re-capture
    g, d, _, _ = make_nets(dict(seed=7, bias_scale=0.02), dict(seed=8, bias_scale=0.02))
    dt, gt, go, do = _trainers(g, d)
    for i in range(4):
        s, f = batches[i]
        (dt if i % 2 == 0 else gt).train(s, f)
    old = dt._runner
    do.flatten()
    l_d = dt.train(*batches[4])["d_loss"]
    assert dt._runner is not old, "the trainer must re-plan when the bucket moved"
    assert abs(l_d - l0[4]) <= 1e-5 * abs(l0[4])


def test_sign_words_train_step_equals_fp32_activations(monkeypatch):
    """r04: the G-step saves one sign bit per element of the atoms' u (and reads t's signs) where the backward pass only
    needs the LeakyReLU derivative.  Nothing but bookkeeping changes: after D, G, D, G at B = 4 the losses, both gradient
    buckets and every parameter are BITWISE what MSYNTH_ATOM_SIGNS=0 (fp32 activations saved) gives; and the sign-word path
    is really the one that runs (the library notes the instantiation)."""
    from featuresynth._ops import lib as L
    from featuresynth._synthetic import synthetic_features, synthetic_samples
    B, T = 4, 8
    out = {}
    for mode in ("signs", "fp32"):
        monkeypatch.setenv("MSYNTH_GRAPH", "0")
        if mode == "fp32":
            monkeypatch.setenv("MSYNTH_ATOM_SIGNS", "0")
        g, d, _, _ = make_nets(dict(seed=7, bias_scale=0.02), dict(seed=8, bias_scale=0.02))
        dt, gt, go, do = _trainers(g, d)
        losses = []
        for step in range(4):
            s = dev(synthetic_samples(B, T * 256, rank=step))
            f = dev(synthetic_features(B, 80, T, rank=step))
            if step == 1:
                L.profile_begin()
            losses.append(dt.train(s, f)["d_loss"] if step % 2 == 0 else gt.train(s, f)["g_loss"])
            if step == 1:
                names = [c.get("kernel") or n for n, c, _ in L.profile_end()]
                masked = sum(n.startswith("k_atom_fwd") and n.endswith("true>") for n in names)
                assert masked == (24 if mode == "signs" else 0), (mode, masked)      # 12 training forwards + 12 backward datas
        out[mode] = (losses, {k: host(v) for k, v in list(g.state_dict().items()) + list(d.state_dict().items())},
                     host(go.flat_grads), host(do.flat_grads))
        monkeypatch.delenv("MSYNTH_ATOM_SIGNS", raising=False)
    assert out["signs"][0] == out["fp32"][0]
    assert np.array_equal(out["signs"][2], out["fp32"][2]) and np.array_equal(out["signs"][3], out["fp32"][3])
    for k in out["signs"][1]:
        assert np.array_equal(out["signs"][1][k], out["fp32"][1][k]), k
