"""Host-side logic of the featuresynth surface (CPU only, no kernels launched)."""
import numpy as np
import pytest
import torch


def test_state_dict_contract_matches_reference_shapes():
    """Keys and shapes as probed from the reference; the oracle's shape tables
    were checked against the imported reference when the fixtures were made."""
    import featuresynth as fs
    from oracle import oracle as O
    for mels in (80, 128):
        g = fs.MelGanGenerator(32, mels)
        assert [(k, tuple(v.shape)) for k, v in g.state_dict().items()] == O.generator_param_shapes(mels)
    d = fs.MelGanDiscriminator()
    assert [(k, tuple(v.shape)) for k, v in d.state_dict().items()] == O.discriminator_param_shapes()
    assert sum(p.numel() for p in fs.MelGanGenerator(32, 80).parameters()) == 4519937
    assert sum(p.numel() for p in d.parameters()) == 5637953
    assert len(list(g.main)) == 17 and d.scales == 2
    fd = fs.FullDiscriminator()
    assert list(fd.state_dict().keys())[0] == "main.0.weight" and "judge.bias" in fd.state_dict()


def test_golden_param_names(golden):
    import featuresynth as fs
    assert list(fs.MelGanGenerator(32, 80).state_dict().keys()) == list(golden("g_fwd")["param_names"])
    assert list(fs.MelGanDiscriminator().state_dict().keys()) == list(golden("d_fwd")["param_names"])


def test_weights_init_semantics():
    import featuresynth as fs
    from featuresynth.experiment.init import weights_init
    torch.manual_seed(0)
    g = fs.MelGanGenerator(32, 80).apply(weights_init)
    for name, p in g.named_parameters():
        if name.endswith("bias"):
            assert float(p.abs().max()) == 0.0
        elif p.numel() > 10000:
            assert abs(float(p.std()) - 0.02) < 2e-3 and abs(float(p.mean())) < 1e-3
    lin = torch.nn.Linear(4, 4)
    before = lin.weight.clone()
    weights_init(lin)                      # class name without "Conv": untouched
    assert torch.equal(before, lin.weight)


def test_no_cpu_fallback():
    import featuresynth as fs
    from featuresynth import loss as LS
    g, d = fs.MelGanGenerator(32, 80), fs.MelGanDiscriminator()
    with pytest.raises(RuntimeError, match="HIP device"):
        g(torch.zeros(1, 80, 32))
    with pytest.raises(RuntimeError, match="HIP device"):
        d(torch.zeros(1, 1, 8192))
    with pytest.raises(RuntimeError, match="HIP device"):
        LS.hinge_generator_loss(torch.zeros(2, 1, 9))
    with pytest.raises(RuntimeError):
        fs.FlatAdam(g.parameters(), lr=1e-4, betas=(0.5, 0.9)).step()
    from featuresynth.feature.feature import Audio2Mel
    with pytest.raises(RuntimeError, match="HIP device"):
        Audio2Mel()(np.zeros(22050, np.float32))


def test_audio2mel_buffers_and_basis(golden):
    from featuresynth.feature.feature import Audio2Mel, slaney_mel_basis
    z = golden("audio2mel")
    a = Audio2Mel()
    assert set(dict(a.named_buffers())) == {"mel_basis", "window"}
    assert tuple(a.mel_basis.shape) == (80, 513) and tuple(a.window.shape) == (1024,)
    assert np.abs(a.window.numpy() - z["hann1024"]).max() < 1e-6
    for n in (80, 128):
        assert np.abs(slaney_mel_basis(22050, 1024, n) - z["basis%d" % n]).max() < 1e-6
    with pytest.raises(NotImplementedError):
        Audio2Mel(win_length=512)


def test_workload_spec_matches_survey():
    """10.56 GFLOP per element per call from the layer spec."""
    from featuresynth import _workload as W
    d1, g1 = W.totals(W.d_step_launches(1)), W.totals(W.g_step_launches(1))
    assert abs(d1["flops"] / 1e9 - 8.51) < 0.15 and abs(g1["flops"] / 1e9 - 12.62) < 0.15
    d32 = W.totals(W.d_step_launches(32))
    assert abs(d32["flops"] / 32 / d1["flops"] - 1.0) < 0.01
    assert W.feature_elems(1, 8192) == 1039008             # loss.py feature-matching elements
    assert W.roofline_seconds(W.g_step_launches(32), 8e12, 157.3e12) > 0


def test_stage1_workload_spec():
    """Layer spec of the stage-1 networks (bench.py --model twostage: step_roofline) against the modules themselves: parameter
    counts, output geometry, and the forward MACs of the transposed-conv stack counted from the layer shapes."""
    import featuresynth as fs
    from featuresynth import _workload as W
    g = fs.featuregenerator.SpectrogramFeatureGenerator(out_channels=128, noise_dim=128)
    d = fs.featurediscriminator.SpectrogramFeatureDiscriminator(feature_channels=128, channels=256)
    assert sum(p.numel() for p in g.parameters()) == W.S1_NPARAM_G
    assert sum(p.numel() for p in d.parameters()) == W.S1_NPARAM_D
    H = Wd = 4
    macs = 128 * 16384
    for layer, (cin, cout, kh, sh) in zip(g.stack, W.S1_G):
        assert tuple(layer.weight.shape) == (cin, cout, kh, 4) and layer.stride == (sh, 2)
        macs += H * Wd * cin * cout * kh * 4
        H, Wd = H * sh, 2 * Wd
    assert (H, Wd) == (128, 512)
    fwd = W.totals(W.stage1_generator_launches(1, "fwd"))
    assert fwd["flops"] == 2 * macs
    dsteps, gsteps = W.totals(W.stage1_d_step_launches(32)), W.totals(W.stage1_g_step_launches(32))
    assert 300e9 < dsteps["flops"] < 400e9 and 350e9 < gsteps["flops"] < 450e9
    assert [m.dilation[0] for m in d.stack.main] == list(W.S1_D_DIL)


def test_realmelgan_workload_spec():
    """Layer spec of the weight-normed MelGAN (bench.py --model realmelgan: step_roofline) against the modules: parameter
    counts, the ResnetBlock's three convs per block, output length."""
    from featuresynth import _workload as W
    from featuresynth.experiment import realmelgan as R
    g, d = R.Generator(128, 32, 3), R.Discriminator(3, 16, 4, 4)
    assert sum(p.numel() for p in g.parameters()) == W.real_generator_nparam(128)
    assert sum(p.numel() for p in d.parameters()) == W.REAL_NPARAM_D
    blocks = [m for m in g.model if isinstance(m, R.ResnetBlock)]
    assert len(blocks) == 12 and [b.block[2].cfg[2] for b in blocks[:3]] == [1, 3, 9]
    fwd = W.real_generator_launches(1, 128, 32, "fwd")
    assert sum(1 for n, _ in fwd if n.startswith("res")) == 36
    macs = 128 * 512 * 7 * 32
    L = 32
    for cin, cout, k, s, p in W.REAL_UPS:
        macs += L * cin * cout * k              # every input sample meets every tap once
        L = (L - 1) * s - 2 * p + k
        macs += 3 * L * cout * cout * (3 + 1 + 1)
    assert L == 8192
    macs += L * 32 * 7
    assert abs(W.totals(fwd)["flops"] / (2 * macs) - 1.0) < 0.01
    assert W.totals(W.real_d_step_launches(32))["flops"] > 200e9


def test_synthetic_inputs_are_deterministic():
    from featuresynth._synthetic import synthetic_features, synthetic_samples, synthetic_state_dict
    a, b = synthetic_samples(2, 64, rank=3), synthetic_samples(2, 64, rank=3)
    assert np.array_equal(a, b) and np.abs(a).max() <= 0.95
    assert not np.array_equal(a, synthetic_samples(2, 64, rank=4))
    assert synthetic_features(2, 80, 4).shape == (2, 80, 4)
    sd = synthetic_state_dict([("w.weight", (4, 3, 3)), ("w.bias", (4,))])
    assert float(np.abs(sd["w.bias"]).max()) == 0.0 and sd["w.weight"].dtype == np.float32


def test_losses_generic_composition_signature():
    import inspect
    from featuresynth import loss as LS
    assert "gan_loss" in inspect.signature(LS.mel_gan_disc_loss).parameters
    sig = inspect.signature(LS.mel_gan_gen_loss).parameters
    assert list(sig)[:4] == ["real_features", "fake_features", "real_judgements", "fake_judgements"]
    assert sig["feature_loss_weight"].default == 10


def test_trainer_surface():
    import inspect
    from featuresynth.train import DiscriminatorTrainer, GeneratorTrainer, training_loop
    for cls in (GeneratorTrainer, DiscriminatorTrainer):
        assert list(inspect.signature(cls.__init__).parameters)[1:] == [
            "generator", "g_optim", "discriminator", "d_optim", "loss", "sub_loss"]
    assert inspect.isgeneratorfunction(training_loop)


def test_experiment_surface(tmp_path, monkeypatch):
    """the experiment object evaluate.py instantiates by name
    (evaluate.py:52 `getattr(featuresynth.experiment, name)()`)."""
    import featuresynth.experiment as E
    from featuresynth.optim import FlatAdam
    exp = getattr(E, "MultiScaleMelGanExperiment")()
    assert exp._name() == "multiscalemelgan"
    assert exp._gen_name("p_") == "trained_models/p_multiscalemelgan_gen.dat"
    assert exp.feature_spec == {"audio": (8192, 1), "spectrogram": (32, 128)}
    assert exp.inference_spec == {"audio": (32768, 1), "spectrogram": (128, 128)}
    assert isinstance(exp._g_optim, FlatAdam) and exp._g_optim.param_groups[0]["betas"] == (0.5, 0.9)
    assert exp._g_optim.param_groups[0]["lr"] == 1e-4
    steps = [next(exp.training_steps) for _ in range(4)]
    assert steps[0] == exp.discriminator_trainer and steps[1] == exp.generator_trainer   # D first
    assert steps[2] == steps[0]
    w = exp.generator.main[1].weight
    assert abs(float(w.std()) - 0.02) < 2e-3                     # weights_init applied
    with pytest.raises(NotImplementedError):
        exp.batch_stream("/data", "*.wav", 4)
    s, f = next(exp.synthetic_batch_stream(3))
    assert s.shape == (3, 1, 8192) and f.shape == (3, 128, 32) and s.dtype == np.float32
    ps, pf = exp.preprocess_batch((s, f))
    assert ps.shape == s.shape and pf.shape == f.shape
    monkeypatch.chdir(tmp_path)
    exp.checkpoint("t_")                                           # CPU state_dicts round-trip
    exp2 = E.MultiScaleMelGanExperiment()
    exp2.resume("t_")
    for (k, a), (_, b) in zip(exp.generator.state_dict().items(), exp2.generator.state_dict().items()):
        assert torch.equal(a, b), k
    real = E.RealMelGanExperiment(optimizer="torch")
    assert real._name() == "realmelgan" and isinstance(real._d_optim, torch.optim.Adam)


def test_stage1_module_surface(golden):
    """class names, constructor signatures and state_dict keys / shapes of the stage-1 modules
    against the imported reference classes (tests/golden/stage1.npz); weights_init touches the conv layers only."""
    import featuresynth as fs
    from featuresynth.experiment.init import weights_init
    z = golden("stage1")
    g = fs.featuregenerator.SpectrogramFeatureGenerator(out_channels=128, noise_dim=128)
    d = fs.featurediscriminator.SpectrogramFeatureDiscriminator(feature_channels=128, channels=256)
    assert list(g.state_dict().keys()) == list(z["g_param_names"])
    assert [str(tuple(v.shape)) for v in g.state_dict().values()] == list(z["g_param_shapes"])
    assert list(d.state_dict().keys()) == list(z["d_param_names"])
    assert [str(tuple(v.shape)) for v in d.state_dict().values()] == list(z["d_param_shapes"])
    lin_before = g.initial.weight.detach().clone()
    g.apply(weights_init)
    assert torch.equal(g.initial.weight, lin_before)                     # nn.Linear is not a "Conv" (init.py:4)
    assert abs(float(g.stack[0].weight.detach().std()) - 0.02) < 2e-3 and float(g.stack[0].bias.detach().abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        d(torch.zeros(1, 128, 512))                                      # CPU tensor: no fallback
    from featuresynth.util.modules import HipConvTranspose2d
    with pytest.raises(NotImplementedError):
        HipConvTranspose2d(8, 8, (3, 3), (1, 1), (1, 1))


def test_stage1_experiment_surface():
    import featuresynth.experiment as E
    exp = E.TwoDimGeneratorFeatureExperiment()
    assert exp.condition_shape == (128, 1) and exp.feature_spec == {"spectrogram": (512, 128)}
    spec, cond = exp.preprocess_batch((np.zeros((3, 128, 512), np.float32),))
    assert spec.shape == (3, 128, 512) and cond.shape == (3, 128, 1) and cond.dtype == np.float32
    (batch,) = next(exp.synthetic_batch_stream(2))
    assert batch.shape == (2, 128, 512)
    assert exp._gen_name() == "trained_models/twodimgeneratorfeature_gen.dat"
    d_step, g_step = next(exp.training_steps), next(exp.training_steps)
    assert d_step.__self__ is exp.d_trainer and g_step.__self__ is exp.g_trainer
    with pytest.raises(NotImplementedError):
        exp.batch_stream("/x", "*.wav", 2)


def test_kaiser_best_filter_table():
    """feature.sinc_window with resampy's published kaiser_best parameters: table size, unit DC gain after the
    rolloff scaling, monotone main lobe."""
    from featuresynth.feature.feature import KAISER_BEST, sinc_window
    win, num_table = sinc_window(**KAISER_BEST)
    assert num_table == 512 and win.shape == (512 * 64 + 1,)
    assert abs(win[0] - KAISER_BEST["rolloff"]) < 1e-12
    assert np.all(np.diff(win[:512]) < 0)
    full = np.concatenate([win[:0:-1], win])          # symmetric filter sampled at 512 per zero crossing
    assert abs(full.sum() / 512 - 1.0) < 1e-3
