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
