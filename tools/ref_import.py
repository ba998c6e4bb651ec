"""Import the reference's hot-path leaf modules from /root/reference (this container only).

Test infrastructure: used by tools/make_golden.py to pin the oracle.  The
reference's package __init__ files pull in lmdb / zounds / librosa (absent, no
network), so we register empty parent packages whose __path__ points into the
reference tree and import the *unmodified* leaf files below them
(SURVEY.md section 8(c)).  Nothing from the reference is copied into this repo and
this module is never imported by the product path, the -m gpu tests, smoke() or
bench.py.
"""
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("MSYNTH_REFERENCE_ROOT", "/root/reference")


def _pkg(name, rel):
    mod = types.ModuleType(name)
    mod.__path__ = [os.path.join(REF_ROOT, rel)]
    mod.__package__ = name
    sys.modules[name] = mod
    return mod


def load_reference():
    """Returns a namespace with the reference's hot-path classes/functions."""
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree %s not present" % REF_ROOT)
    sys.dont_write_bytecode = True  # never leave __pycache__ in the reference tree
    import scipy.signal
    import scipy.signal.windows
    if not hasattr(scipy.signal, "hann"):  # generator/ddsp.py:62 imports it (SciPy < 1.13 name)
        scipy.signal.hann = scipy.signal.windows.hann
    if "zounds" not in sys.modules:  # train/train.py:5 imports zounds, never uses it
        sys.modules["zounds"] = types.ModuleType("zounds")

    fs = _pkg("featuresynth", "featuresynth")
    util = _pkg("featuresynth.util", "featuresynth/util")
    _pkg("featuresynth.generator", "featuresynth/generator")
    _pkg("featuresynth.discriminator", "featuresynth/discriminator")
    loss_pkg = _pkg("featuresynth.loss", "featuresynth/loss")
    _pkg("featuresynth.train", "featuresynth/train")
    _pkg("featuresynth.experiment", "featuresynth/experiment")

    dev = importlib.import_module("featuresynth.util.device")
    util.device = dev.device
    modules = importlib.import_module("featuresynth.util.modules")
    gen_full = importlib.import_module("featuresynth.generator.full")
    disc_full = importlib.import_module("featuresynth.discriminator.full")
    disc_melgan = importlib.import_module("featuresynth.discriminator.melgan")
    loss = importlib.import_module("featuresynth.loss.loss")
    for k in ("mel_gan_disc_loss", "mel_gan_gen_loss", "hinge_generator_loss",
              "hinge_discriminator_loss", "least_squares_disc_loss",
              "least_squares_generator_loss"):
        setattr(loss_pkg, k, getattr(loss, k))
    train = importlib.import_module("featuresynth.train.train")
    init = importlib.import_module("featuresynth.experiment.init")

    ns = types.SimpleNamespace(
        MelGanGenerator=gen_full.MelGanGenerator,
        FullDiscriminator=disc_full.FullDiscriminator,
        MelGanDiscriminator=disc_melgan.MelGanDiscriminator,
        ResidualAtom=modules.ResidualAtom,
        ResidualStack=modules.ResidualStack,
        zero_grad=modules.zero_grad,
        loss=loss,
        train=train,
        weights_init=init.weights_init,
    )
    return ns


def load_realmelgan():
    """The reference's experiment/realmelgan.py.  Its module-level imports of
    the data layer (..audio, .experiment, ..feature, zounds.SR22050) are satisfied with inert stubs;
    the model classes and losses are the unmodified reference code."""
    load_reference()
    for name, attrs in (("featuresynth.audio", {"RawAudio": object}),
                        ("featuresynth.experiment.experiment", {"Experiment": object}),
                        ("featuresynth.feature", {"audio": None, "spectrogram": None})):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for k, v in attrs.items():
                setattr(m, k, v)
            sys.modules[name] = m
    sys.modules["zounds"].SR22050 = lambda: 22050
    return importlib.import_module("featuresynth.experiment.realmelgan")


def load_stage1():
    """The reference's stage-1 spectrogram GAN classes: the unmodified leaf files
    featuregenerator/upscale.py and featurediscriminator/upscale.py (their only package-relative import is
    util/modules.py:DilatedStack)."""
    ns = load_reference()
    _pkg("featuresynth.featuregenerator", "featuresynth/featuregenerator")
    _pkg("featuresynth.featurediscriminator", "featuresynth/featurediscriminator")
    gen = importlib.import_module("featuresynth.featuregenerator.upscale")
    disc = importlib.import_module("featuresynth.featurediscriminator.upscale")
    ns.SpectrogramFeatureGenerator = gen.SpectrogramFeatureGenerator
    ns.SpectrogramFeatureDiscriminator = disc.SpectrogramFeatureDiscriminator
    return ns


def unload_reference():
    for k in [k for k in sys.modules if k == "featuresynth" or k.startswith("featuresynth.")]:
        del sys.modules[k]
