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
    assert len(f1) == 6 and torch.equal(j1, judges[0])


def _oracle_step(kind, gsd, dsd, samples, feats):
    from oracle import oracle as O
    gw = {k: v.copy() for k, v in gsd.items()}
    dw = {k: v.copy() for k, v in dsd.items()}
    if kind == "d":
        adam = O.AdamState(dw)
        loss, grads = O.d_step(gw, dw, adam, samples, feats)
        return loss, grads, dw, None
    adam = O.AdamState(gw)
    loss, fake, grads = O.g_step(gw, dw, adam, samples, feats)
    return loss, grads, gw, fake


@pytest.mark.parametrize("optim_kind", ["flat", "torch"])
def test_train_steps_vs_oracle(optim_kind, monkeypatch):
    """One D-step and one G-step (B=2, 2048-sample windows) against the oracle: loss, every
    parameter gradient, and the parameters after the Adam update."""
    import featuresynth as fs
    from featuresynth import loss as LS
    from featuresynth._synthetic import synthetic_features, synthetic_samples
    from featuresynth.train import DiscriminatorTrainer, GeneratorTrainer
    monkeypatch.setenv("MSYNTH_GRAPH", "0")
    B, T = 2, 8
    samples, feats = synthetic_samples(B, T * 256), synthetic_features(B, 80, T)
    for kind in ("d", "g"):
        g, d, gsd, dsd = make_nets(dict(seed=7, bias_scale=0.02), dict(seed=8, bias_scale=0.02))
        if optim_kind == "flat":
            go = fs.FlatAdam(g.parameters(), lr=1e-4, betas=(0.5, 0.9))
            do = fs.FlatAdam(d.parameters(), lr=1e-4, betas=(0.5, 0.9))
        else:
            go = torch.optim.Adam(g.parameters(), lr=1e-4, betas=(0.5, 0.9))
            do = torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.9))
        o_loss, o_grads, o_params, o_fake = _oracle_step(kind, gsd, dsd, samples, feats)
        if kind == "d":
            tr = DiscriminatorTrainer(g, go, d, do, LS.mel_gan_disc_loss)
            res = tr.train(dev(samples), dev(feats))
            assert set(res) == {"d_loss"} and isinstance(res["d_loss"], float)
            loss, net = res["d_loss"], d
        else:
            tr = GeneratorTrainer(g, go, d, do, LS.mel_gan_gen_loss)
            res = tr.train(dev(samples), dev(feats))
            assert set(res) == {"g_loss", "fake"} and isinstance(res["fake"], np.ndarray)
            assert res["fake"].shape == (B, 1, T * 256)
            assert rel_l2(res["fake"], o_fake) < 1e-4
            loss, net = res["g_loss"], g
        assert abs(loss - o_loss) <= 1e-4 * abs(o_loss), (kind, loss, o_loss)
        # A LeakyReLU input within rounding of 0 can take the other slope on the device than in the
        # float64 oracle; at B=2 one such element moves the gradients of its layer by up to ~1e-2 and
        # of the layers behind it by ~1e-3 (measured: one flip at +1.3e-8 / -1.9e-9 between two
        # summation orders).  So: every layer within 3e-2, and the typical layer within 1e-4.
        errs = {}
        for k, p in net.named_parameters():
            errs[k] = rel_l2(host(p.grad), o_grads[k]) if np.linalg.norm(o_grads[k]) > 0 else float(p.grad.abs().max())
            assert errs[k] < 3e-2, (kind, k, errs[k])
        worst = max(errs.values())
        assert float(np.median(list(errs.values()))) < 1e-4, (kind, sorted(errs.items(), key=lambda kv: -kv[1])[:5])
        # Adam: identical gradients up to 1e-3 => every entry moves by at most ~lr; compare where the
        # oracle's gradient is clearly above the rounding floor
        for k, p in net.named_parameters():
            diff = np.abs(host(p) - o_params[k])
            assert diff.max() <= 2.1e-4, (kind, k, diff.max())
            big = np.abs(o_grads[k]) > 1e-3 * np.abs(o_grads[k]).max()
            if big.any() and errs[k] < 1e-3:     # (a layer behind a flipped mask only meets the +-lr bound)
                assert diff[big].max() < 2e-5, (kind, k, diff[big].max())
        print("%s-step [%s]: loss %.8f (oracle %.8f), worst grad rel-L2 %.2e" % (kind, optim_kind, loss, o_loss, worst))


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
    # update, rounding-level gradient entries may step the other way (DESIGN.md "Adam sensitivity"),
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


def test_train_step_mel128_odd_batch_vs_torch_graph():
    """A D-step and a G-step with 128 mel channels (experiment/melgan.py:23) and B=3 against the
    torch-functional oracle's autograd."""
    import torch as th
    import featuresynth as fs
    from featuresynth import loss as LS
    from featuresynth._synthetic import synthetic_features, synthetic_samples
    from featuresynth.train import DiscriminatorTrainer, GeneratorTrainer
    from oracle import torch_graph as TG
    B, T = 3, 5
    samples, feats = synthetic_samples(B, T * 256, rank=1), synthetic_features(B, 128, T, rank=1)
    for kind in ("d", "g"):
        g, d, gsd, dsd = make_nets(dict(seed=7, bias_scale=0.02), dict(seed=8, bias_scale=0.02), mels=128)
        go = fs.FlatAdam(g.parameters(), lr=1e-4, betas=(0.5, 0.9))
        do = fs.FlatAdam(d.parameters(), lr=1e-4, betas=(0.5, 0.9))
        gp, dp = TG.to_params(gsd), TG.to_params(dsd)
        fake = TG.generator(gp, th.from_numpy(feats))
        ff, fj = TG.discriminator(dp, fake)
        rf, rj = TG.discriminator(dp, th.from_numpy(samples))
        if kind == "d":
            ref = TG.disc_loss(rj, fj); ref.backward()
            loss = DiscriminatorTrainer(g, go, d, do, LS.mel_gan_disc_loss).train(dev(samples), dev(feats))["d_loss"]
            net, refp = d, dp
        else:
            ref = TG.gen_loss(rf, ff, fj); ref.backward()
            loss = GeneratorTrainer(g, go, d, do, LS.mel_gan_gen_loss).train(dev(samples), dev(feats))["g_loss"]
            net, refp = g, gp
        assert abs(loss - ref.item()) <= 1e-4 * abs(ref.item())
        for k, p in net.named_parameters():
            r = refp[k].grad.numpy()
            assert rel_l2(host(p.grad), r) < 2e-3 or np.linalg.norm(r) < 1e-12, (kind, k)


def test_full_size_train_step_vs_torch_graph():
    """BASELINE config 3 sizes (B=32, 8192-sample windows, 80 mels): one D-step and one G-step
    against the torch-functional oracle's autograd -- the kernels, tile shapes and split-K plans
    the benchmark actually runs.  fp32 torch is the comparison here (the float64 C oracle would
    take minutes at this size), so the bounds are those of two fp32 summation orders: loss 1e-4,
    typical layer gradient 1e-3, every layer 5e-2 (LeakyReLU mask flips)."""
    import torch as th
    import featuresynth as fs
    from featuresynth import loss as LS
    from featuresynth._synthetic import synthetic_features, synthetic_samples
    from featuresynth.train import DiscriminatorTrainer, GeneratorTrainer
    from oracle import torch_graph as TG
    B, T = 32, 32
    samples, feats = synthetic_samples(B, T * 256, rank=2), synthetic_features(B, 80, T, rank=2)
    th.set_num_threads(16)
    for kind in ("d", "g"):
        g, d, gsd, dsd = make_nets(dict(seed=7, bias_scale=0.02), dict(seed=8, bias_scale=0.02))
        go = fs.FlatAdam(g.parameters(), lr=1e-4, betas=(0.5, 0.9))
        do = fs.FlatAdam(d.parameters(), lr=1e-4, betas=(0.5, 0.9))
        gp, dp = TG.to_params(gsd), TG.to_params(dsd)
        fake = TG.generator(gp, th.from_numpy(feats))
        ff, fj = TG.discriminator(dp, fake)
        rf, rj = TG.discriminator(dp, th.from_numpy(samples))
        if kind == "d":
            ref = TG.disc_loss(rj, fj); ref.backward()
            loss = DiscriminatorTrainer(g, go, d, do, LS.mel_gan_disc_loss).train(dev(samples), dev(feats))["d_loss"]
            net, refp = d, dp
        else:
            ref = TG.gen_loss(rf, ff, fj); ref.backward()
            res = GeneratorTrainer(g, go, d, do, LS.mel_gan_gen_loss).train(dev(samples), dev(feats))
            loss, net, refp = res["g_loss"], g, gp
            assert rel_l2(res["fake"], fake.detach().numpy()) < 1e-4
        assert abs(loss - ref.item()) <= 1e-4 * abs(ref.item()), (kind, loss, ref.item())
        # (the hinge gradient of the judge bias is sum(+1/n over fake) - sum(1/n over real) = 0 up to
        # rounding at this init: a relative error is meaningless for such fully cancelling entries,
        # they are held to an absolute bound instead)
        refs = {k: refp[k].grad.numpy() for k, _ in net.named_parameters()}
        gmax = max(float(np.linalg.norm(r)) for r in refs.values())
        errs = {}
        for k, p in net.named_parameters():
            r = refs[k]
            if np.linalg.norm(r) > 1e-4 * gmax:
                errs[k] = rel_l2(host(p.grad), r)
            else:      # e.g. judge bias: exactly 0 here, -2^-22 (rounding residue) in fp32 torch
                assert float(np.abs(host(p.grad) - r).max()) < 1e-4 * gmax, (kind, k)
        wk = max(errs, key=errs.get)
        assert max(errs.values()) < 5e-2, (kind, sorted(errs.items(), key=lambda kv: -kv[1])[:4], wk, host(dict(net.named_parameters())[wk].grad).ravel()[:4], refs[wk].ravel()[:4], gmax)
        assert float(np.median(list(errs.values()))) < 1e-3, (kind, sorted(errs.items(), key=lambda kv: -kv[1])[:4])
        print("%s-step full size: loss %.6f (oracle %.6f), grad rel-L2 median %.1e max %.1e" % (
            kind, loss, ref.item(), float(np.median(list(errs.values()))), max(errs.values())))


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
