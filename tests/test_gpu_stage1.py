"""GPU parity tests of the stage-1 2-D conv mel GAN (SURVEY.md 8(f) row 2 / BASELINE config 5): the
SpectrogramFeatureGenerator / SpectrogramFeatureDiscriminator on the gfx950 kernels against fixtures made by
the imported reference classes (tests/golden/stage1.npz) and against the torch-functional oracle."""
import numpy as np
import pytest

from conftest import rel_l2, stable_seed

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def _nets():
    import featuresynth as fs
    from featuresynth._synthetic import module_param_shapes, synthetic_state_dict
    g = fs.featuregenerator.SpectrogramFeatureGenerator(out_channels=128, noise_dim=128)
    d = fs.featurediscriminator.SpectrogramFeatureDiscriminator(feature_channels=128, channels=256)
    gsd = synthetic_state_dict(module_param_shapes(g), seed=31, weight_scale=0.03, bias_scale=0.02)
    dsd = synthetic_state_dict(module_param_shapes(d), seed=32, weight_scale=0.03, bias_scale=0.02)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in gsd.items()})
    d.load_state_dict({k: torch.from_numpy(v) for k, v in dsd.items()})
    return g.cuda(), d.cuda(), gsd, dsd


def _inputs():
    noise = np.random.default_rng(6).standard_normal((2, 128, 1)).astype(np.float32)
    real = (np.random.default_rng(7).standard_normal((2, 128, 512)) * 0.5).astype(np.float32)
    return noise, real


@pytest.mark.parametrize("geom", [((4, 4), (2, 2)), ((3, 4), (1, 2))], ids=["k44_s22", "k34_s12"])
@pytest.mark.parametrize("shape", [(2, 24, 5, 8, 16), (1, 64, 4, 4, 32), (3, 8, 9, 12, 1)], ids=["c24", "c64", "cout1"])
def test_conv_transpose2d_vs_torch(geom, shape):
    """HipConvTranspose2d (lines through the 1-D transposed-conv kernels) against F.conv_transpose2d in
    float64: forward, input gradient, weight and bias gradients."""
    import torch.nn.functional as F
    from featuresynth.util.modules import HipConvTranspose2d
    (k, s), (B, Cin, H, W, Cout) = geom, shape
    rng = np.random.default_rng(stable_seed("ct2d%s%s" % (geom, shape)))
    m = HipConvTranspose2d(Cin, Cout, k, s, (1, 1), activation=None).cuda()
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cin, Cout) + k) * 0.1).astype(np.float32)
    b = (rng.standard_normal((Cout,)) * 0.1).astype(np.float32)
    with torch.no_grad():
        m.weight.copy_(dev(w)); m.bias.copy_(dev(b))
    xt = dev(x).requires_grad_(True)
    y = m(xt)
    xr, wr, br = [torch.from_numpy(a).double().requires_grad_(True) for a in (x, w, b)]
    yr = F.conv_transpose2d(xr, wr, br, s, (1, 1))
    assert tuple(y.shape) == tuple(yr.shape)
    assert rel_l2(host(y), yr.detach().numpy()) < 1e-5
    gy = rng.standard_normal(tuple(yr.shape)).astype(np.float32)
    y.backward(dev(gy)); yr.backward(torch.from_numpy(gy).double())
    assert rel_l2(host(xt.grad), xr.grad.numpy()) < 1e-4
    assert rel_l2(host(m.weight.grad), wr.grad.numpy()) < 1e-4
    assert rel_l2(host(m.bias.grad), br.grad.numpy()) < 1e-4


def test_stage1_forward_golden(golden):
    from featuresynth._synthetic import strided_sample
    z = golden("stage1")
    g, d, _, _ = _nets()
    assert list(g.state_dict().keys()) == list(z["g_param_names"])
    assert [str(tuple(v.shape)) for v in g.state_dict().values()] == list(z["g_param_shapes"])
    assert list(d.state_dict().keys()) == list(z["d_param_names"])
    assert [str(tuple(v.shape)) for v in d.state_dict().values()] == list(z["d_param_shapes"])
    noise, real = _inputs()
    with torch.no_grad():
        y = g(dev(noise))
        feats, judge = d(dev(real), None)
    assert tuple(y.shape) == tuple(z["g/shape"])
    e = rel_l2(strided_sample(host(y), 8192), z["g/y_smp_ref32"])
    print("stage-1 generator rel-L2 vs reference %.3e" % e)
    assert e < 1e-4
    assert abs(float(np.linalg.norm(host(y).astype(np.float64))) - z["g/y_sum_ref64"][0]) < 1e-4 * z["g/y_sum_ref64"][0]
    assert rel_l2(host(judge), z["d/j_ref32"]) < 1e-4
    assert len(feats) == 7
    for i, f in enumerate(feats):
        assert tuple(f.shape) == tuple(z["d/f%d_shape" % i])
        assert rel_l2(strided_sample(host(f), 2048), z["d/f%d_smp_ref32" % i]) < 1e-4


@pytest.mark.parametrize("optim_kind", ["flat", "torch"])
def test_stage1_train_steps_golden(golden, optim_kind, monkeypatch):
    """One D-step and one G-step through featuresynth.train with the least-squares losses of
    experiment/featureexperiment.py:289-293, against the reference's own trainers: loss, the generated batch,
    every parameter gradient (<= 1e-3 rel-L2 on the reference's sample of it and on its norm)."""
    import featuresynth as fs
    from featuresynth._synthetic import strided_sample
    from featuresynth.experiment.featureexperiment import _disc_loss, _gen_loss
    from featuresynth.train import DiscriminatorTrainer, GeneratorTrainer
    monkeypatch.setenv("MSYNTH_GRAPH", "0")
    z = golden("stage1")
    noise, real = _inputs()
    for kind in ("d", "g"):
        g, d, _, _ = _nets()
        make = (lambda ps: fs.FlatAdam(ps, lr=1e-4, betas=(0.5, 0.9))) if optim_kind == "flat" else \
            (lambda ps: torch.optim.Adam(ps, lr=1e-4, betas=(0.5, 0.9)))
        go, do = make(g.parameters()), make(d.parameters())
        if kind == "d":
            res = DiscriminatorTrainer(g, go, d, do, _disc_loss, sub_loss=None).train(dev(real), dev(noise))
            loss, net = res["d_loss"], d
        else:
            res = GeneratorTrainer(g, go, d, do, _gen_loss, sub_loss=None).train(dev(real), dev(noise))
            loss, net = res["g_loss"], g
            assert res["fake"].shape == (2, 128, 512)
            assert rel_l2(strided_sample(res["fake"], 8192), z["step/fake_smp"]) < 1e-4
        ref = float(z["step/%s_loss" % kind][0])
        assert abs(loss - ref) <= 1e-4 * abs(ref), (kind, loss, ref)
        errs = {}
        for k, p in net.named_parameters():
            smp, s = z["step/%sgrad_smp/%s" % (kind, k)], z["step/%sgrad_sum/%s" % (kind, k)]
            errs[k] = max(rel_l2(strided_sample(host(p.grad)), smp),
                          abs(float(np.linalg.norm(host(p.grad).astype(np.float64))) - s[0]) / (s[0] + 1e-30))
        worst = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
        print("stage-1 %s-step [%s]: loss %.6f (reference %.6f), worst grads %s" % (kind, optim_kind, loss, ref, worst))
        assert worst[0][1] < 1e-3, (kind, worst)


def test_stage1_experiment_loop_and_vocoder(tmp_path, monkeypatch):
    """TwoDimGeneratorFeatureExperiment through featuresynth.train.training_loop (alternating D/G steps, the
    graphs captured on the second call of each trainer), checkpoint / resume, and generated spectrograms
    through a stage-2 vocoder (the two stages end to end)."""
    import featuresynth as fs
    import featuresynth.experiment as E
    from featuresynth.train import training_loop
    monkeypatch.chdir(tmp_path)
    device = torch.device("cuda", 0)
    torch.manual_seed(0)
    vocoder = fs.MelGanGenerator(32, 128).to(device)
    exp = E.TwoDimGeneratorFeatureExperiment(vocoder_network=vocoder).to(device)
    seen = []
    logs = list(training_loop(exp.synthetic_batch_stream(2, n_batches=6), exp, device,
                              [lambda e, b, r, i, t: seen.append(sorted(r))]))
    assert len(logs) == 6 and seen == [["d_loss"], ["fake", "g_loss"]] * 3
    assert exp.d_trainer._runner.graphs and exp.g_trainer._runner.graphs
    exp.checkpoint()
    exp2 = E.TwoDimGeneratorFeatureExperiment().to(device)
    exp2.resume()
    for (k, p), (_, q) in zip(exp.feature_generator.state_dict().items(), exp2.feature_generator.state_dict().items()):
        assert torch.equal(p, q), k
    noise = np.random.default_rng(0).standard_normal((1, 128, 1)).astype(np.float32)
    with torch.no_grad():
        spec = exp.feature_generator(dev(noise))
    audio = exp.features_to_audio(host(spec)[:, :, :8])
    assert audio.shape == (1, 1, 8 * 256) and np.isfinite(audio).all()


def test_two_stage_pair_full_size(monkeypatch):
    """BASELINE configs[4] at its per-GPU size (B = 32): the two-stage D,G pair of bench.py --model twostage -- per step one
    stage-1 trainer call (128 x 512 spectrograms, least-squares losses) plus one stage-2 trainer call (the headline vocoder at
    128 mels).  (a) hipGraph replay (calls 3+) against eager execution of the same schedule: every loss and both stages'
    flat gradient buckets agree BITWISE with lr = 0; (b) the stage-1 D-step and G-step gradients against the float64
    torch-functional oracle (oracle/torch_graph_stage1.py on the device, LeakyReLU branches as they fall in float64) at the
    SURVEY 8(d) gates: losses 1e-4, every parameter gradient <= 1e-3 rel-L2."""
    import featuresynth as fs
    import featuresynth.experiment as E
    from featuresynth import loss as LS
    from featuresynth._synthetic import (module_param_shapes, synthetic_features, synthetic_samples,
                                         synthetic_state_dict)
    from featuresynth.train import DiscriminatorTrainer, GeneratorTrainer
    from oracle import torch_graph_stage1 as S1
    B, T = 32, 32
    device = torch.device("cuda", 0)
    rng = np.random.default_rng(77)
    spec = dev((rng.standard_normal((B, 128, 512)) * 0.5).astype(np.float32))
    noise = dev(rng.standard_normal((B, 128, 1)).astype(np.float32))
    samples, feats = dev(synthetic_samples(B, 8192, rank=3)), dev(synthetic_features(B, 128, T, rank=3))

    def build():
        torch.manual_seed(5)
        g = fs.MelGanGenerator(T, 128)
        d = fs.MelGanDiscriminator()
        g.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(module_param_shapes(g), seed=7).items()})
        d.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(module_param_shapes(d), seed=7).items()})
        g.to(device); d.to(device)
        s1 = E.TwoDimGeneratorFeatureExperiment(vocoder_network=g)
        s1g = synthetic_state_dict(module_param_shapes(s1.feature_generator), seed=31, weight_scale=0.03, bias_scale=0.02)
        s1d = synthetic_state_dict(module_param_shapes(s1.feature_disc), seed=32, weight_scale=0.03, bias_scale=0.02)
        s1.feature_generator.load_state_dict({k: torch.from_numpy(v) for k, v in s1g.items()})
        s1.feature_disc.load_state_dict({k: torch.from_numpy(v) for k, v in s1d.items()})
        s1.to(device)
        for opt in (s1.g_optim, s1.d_optim):
            for grp in opt.param_groups:
                grp["lr"] = 0.0
        go = fs.FlatAdam(g.parameters(), lr=0.0, betas=(0.5, 0.9))
        do = fs.FlatAdam(d.parameters(), lr=0.0, betas=(0.5, 0.9))
        return (s1, DiscriminatorTrainer(g, go, d, do, LS.mel_gan_disc_loss), GeneratorTrainer(g, go, d, do, LS.mel_gan_gen_loss),
                go, do, s1g, s1d)

    def run(graph):
        monkeypatch.setenv("MSYNTH_GRAPH", graph)
        s1, dt, gt, go, do, s1g, s1d = build()
        out = []
        for i in range(6):
            if i % 2 == 0:
                r1 = s1.d_trainer.train(spec, noise); r2 = dt.train(samples, feats)
                rec = (r1["d_loss"], r2["d_loss"], host(s1.d_optim.flat_grads).copy(), host(do.flat_grads).copy())
            else:
                r1 = s1.g_trainer.train(spec, noise); r2 = gt.train(samples, feats)
                rec = (r1["g_loss"], r2["g_loss"], host(s1.g_optim.flat_grads).copy(), host(go.flat_grads).copy())
            out.append(rec)
        if graph == "1":
            assert s1.d_trainer._runner.graphs and s1.g_trainer._runner.graphs and dt._runner.graphs and gt._runner.graphs
        return out, s1, s1g, s1d

    eager, _, _, _ = run("0")
    replay, s1, s1g, s1d = run("1")
    for i in range(6):
        assert eager[i][0] == replay[i][0] and eager[i][1] == replay[i][1], (i, eager[i][:2], replay[i][:2])
        assert np.array_equal(eager[i][2], replay[i][2]), "stage-1 bucket, call %d" % i
        assert np.array_equal(eager[i][3], replay[i][3]), "stage-2 bucket, call %d" % i

    # (b) stage-1 gradients of the last REPLAYED D / G calls against the float64 oracle (parameters never moved: lr = 0)
    pg = {k: torch.from_numpy(v).to(device).double().requires_grad_(True) for k, v in s1g.items()}
    pd = {k: torch.from_numpy(v).to(device).double().requires_grad_(True) for k, v in s1d.items()}
    fake = S1.generator(pg, noise.double())
    _, fj = S1.discriminator(pd, fake)
    _, rj = S1.discriminator(pd, spec.double())
    d_loss = S1.ls_disc_loss(rj, fj)
    d_grads = torch.autograd.grad(d_loss, list(pd.values()), retain_graph=True)
    g_loss = S1.ls_gen_loss(fj)
    g_grads = torch.autograd.grad(g_loss, list(pg.values()))
    assert abs(replay[4][0] - float(d_loss)) <= 1e-4 * abs(float(d_loss)), (replay[4][0], float(d_loss))
    assert abs(replay[5][0] - float(g_loss)) <= 1e-4 * abs(float(g_loss)), (replay[5][0], float(g_loss))
    worst = {}
    for (net, opt, grads, call, sd) in ((s1.feature_disc, s1.d_optim, d_grads, 4, s1d), (s1.feature_generator, s1.g_optim, g_grads, 5, s1g)):
        flat = replay[call][2]                   # the bucket as the call left it (the next call's zero_grad clears both buckets)
        params = opt.param_groups[0]["params"]
        names = {id(p): k for k, p in net.named_parameters()}
        by_name = {}
        for p, (off, n) in zip(params, opt._flat[5]):
            by_name[names[id(p)]] = flat[off:off + n].reshape(tuple(p.shape))
        for k, ref in zip(sd.keys(), grads):
            worst[k] = rel_l2(by_name[k], ref.cpu().numpy())
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    print("two-stage B=32: stage-1 gradients vs float64, worst %s" % top)
    assert top[0][1] < 1e-3, top
