"""Experiment glue with the surface of the reference's featuresynth/experiment/experiment.py:67-232
(SURVEY.md 8(f) row 3): builds the two optimizers (Adam 1e-4, betas (0.5, 0.9), :111-117), the two
trainers (:119-133), `training_steps = cycle([D, G])` (:141-144), `checkpoint` / `resume`
(:171-177), `to(device)`, `feature_spec` / `inference_spec` (:217-229) and `preprocess_batch`.

Differences, all additive:
  * optimizers are FlatAdam by default (same hyper-parameters and .step()/.zero_grad() interface;
    pass optimizer="torch" for stock torch.optim.Adam);
  * `checkpoint` / `resume` can also carry the optimizer state (the reference drops it);
  * under torch.distributed, `to(device)` broadcasts rank 0's parameters so replicas start equal,
    and `synthetic_batch_stream` seeds each rank differently;
  * the file / LMDB data layer (`batch_stream`, `from_audio` on zounds objects, reports) is out
    of scope: `batch_stream` raises and points at `synthetic_batch_stream`.
"""
import os
from itertools import cycle

import numpy as np
import torch

from .. import _dist
from ..loss import hinge_discriminator_loss, hinge_generator_loss
from ..optim import FlatAdam
from ..train import DiscriminatorTrainer, GeneratorTrainer
from .init import weights_init


class BaseGanExperiment(object):
    def preprocess_batch(self, batch):
        return batch

    def to(self, device):
        self.generator.to(device)
        self.discriminator.to(device)
        return self


class Experiment(BaseGanExperiment):
    def __init__(self, generator, discriminator, learning_rate, feature_size, audio_repr_class=None,
                 generator_loss=None, discriminator_loss=None, g_init=weights_init,
                 d_init=weights_init, feature_funcs=None, total_samples=16384, feature_channels=256,
                 inference_sequence_factor=4, samplerate=11025,
                 sub_disc_loss=hinge_discriminator_loss, sub_gen_loss=hinge_generator_loss,
                 optimizer="flat"):
        super().__init__()
        self.sub_gen_loss = sub_gen_loss
        self.sub_disc_loss = sub_disc_loss
        self.inference_sequence_factor = inference_sequence_factor
        self.discriminator_init = d_init
        self.generator_init = g_init
        for net, what in ((generator, "generators"), (discriminator, "discriminators")):
            if hasattr(net, "initialize_weights") and not getattr(net, "_ms_native", False):
                raise ValueError("initialize_weights() method on %s is deprecated" % what)
        make = FlatAdam if optimizer == "flat" else torch.optim.Adam
        self._g = generator
        self._g.apply(g_init)
        self._g_optim = make(self._g.parameters(), lr=learning_rate, betas=(0.5, 0.9))
        self._d = discriminator
        self._d.apply(d_init)
        self._d_optim = make(self._d.parameters(), lr=learning_rate, betas=(0.5, 0.9))
        self._g_trainer = GeneratorTrainer(self._g, self._g_optim, self._d, self._d_optim,
                                           generator_loss, self.sub_gen_loss)
        self._d_trainer = DiscriminatorTrainer(self._g, self._g_optim, self._d, self._d_optim,
                                               discriminator_loss, self.sub_disc_loss)
        self._feature_size = feature_size
        self._audio_repr_class = audio_repr_class
        self._feature_funcs = feature_funcs
        self.training_steps = cycle([self.discriminator_trainer, self.generator_trainer])
        self.samplerate = samplerate
        self.total_samples = total_samples
        self.feature_channels = feature_channels

    # ---- names / checkpoints (reference :150-177)
    @classmethod
    def _name(cls):
        return cls.__name__.lower().replace("experiment", "")

    @classmethod
    def _gen_name(cls, prefix=""):
        return "trained_models/%s%s_gen.dat" % (prefix, cls._name())

    @classmethod
    def _disc_name(cls, prefix=""):
        return "trained_models/%s%s_disc.dat" % (prefix, cls._name())

    @classmethod
    def _optim_name(cls, prefix=""):
        return "trained_models/%s%s_optim.dat" % (prefix, cls._name())

    @classmethod
    def load_generator_weights(cls, generator, prefix=""):
        generator.load_state_dict(torch.load(cls._gen_name(prefix)))
        return generator

    def checkpoint(self, prefix="", with_optimizers=False):
        """Under data parallelism the replicas are identical: rank 0 writes, everyone waits for it."""
        if _dist.rank() == 0:
            os.makedirs(os.path.dirname(self._gen_name(prefix)), exist_ok=True)
            torch.save(self.generator.state_dict(), self._gen_name(prefix))
            torch.save(self.discriminator.state_dict(), self._disc_name(prefix))
            if with_optimizers:
                torch.save({"g": _optim_state(self._g_optim), "d": _optim_state(self._d_optim)},
                           self._optim_name(prefix))
        _dist.barrier()

    def resume(self, prefix="", with_optimizers=False):
        self.generator.load_state_dict(torch.load(self._gen_name(prefix)))
        self.discriminator.load_state_dict(torch.load(self._disc_name(prefix)))
        if with_optimizers:
            st = torch.load(self._optim_name(prefix))
            _load_optim_state(self._g_optim, st["g"])
            _load_optim_state(self._d_optim, st["d"])

    # ---- properties (reference :179-193)
    @property
    def generator(self):
        return self._g

    @property
    def discriminator(self):
        return self._d

    @property
    def generator_trainer(self):
        return self._g_trainer.train

    @property
    def discriminator_trainer(self):
        return self._d_trainer.train

    def to(self, device):
        super().to(device)
        if _dist.is_distributed():          # replicas start from rank 0's parameters
            for net in (self._g, self._d):
                for p in net.parameters():
                    _dist.broadcast_(p.data, 0)
        return self

    # ---- batches
    def preprocess_batch(self, batch):
        """RawAudio is the identity representation (audio/representation.py:38-54): samples stay
        (B, 1, N) float32, features (B, channels, frames)."""
        samples, features = batch
        return np.asarray(samples, dtype=np.float32), np.asarray(features, dtype=np.float32)

    def batch_stream(self, path, pattern, batch_size, feature_spec=None):
        raise NotImplementedError(
            "the sound-file / LMDB data layer (featuresynth.data) is outside this build's hot path; "
            "feed training_loop from synthetic_batch_stream() or any iterator of "
            "(samples (B,1,N), features (B,C,T)) float32 numpy pairs")

    def synthetic_batch_stream(self, batch_size, n_batches=None):
        """Endless (or n_batches long) stream of synthetic batches with the layout the reference's
        data layer emits (data/datastore.py:35-37,77-79); every rank draws a different sequence."""
        from .._synthetic import synthetic_features, synthetic_samples
        i = 0
        base = 1000 * _dist.rank()
        while n_batches is None or i < n_batches:
            yield (synthetic_samples(batch_size, self.total_samples, rank=base + i),
                   synthetic_features(batch_size, self.feature_channels, self._feature_size, rank=base + i))
            i += 1

    @property
    def feature_spec(self):
        return {"audio": (self.total_samples, 1),
                "spectrogram": (self._feature_size, self.feature_channels)}

    @property
    def inference_spec(self):
        return {k: (size * self.inference_sequence_factor, channels)
                for k, (size, channels) in self.feature_spec.items()}


def _optim_state(opt):
    if isinstance(opt, FlatAdam):
        fp, fg, m, v, step, _ = opt._flat if opt._flat is not None else (None,) * 6
        if m is None:
            return {"kind": "flat", "empty": True}
        return {"kind": "flat", "empty": False, "m": m.cpu(), "v": v.cpu(), "step": step.cpu()}
    return {"kind": "torch", "state": opt.state_dict()}


def _load_optim_state(opt, st):
    if isinstance(opt, FlatAdam):
        if st.get("kind") != "flat":
            raise ValueError("checkpoint holds a torch.optim.Adam state, the experiment uses FlatAdam")
        if st["empty"]:
            return
        if not opt._bucket_ok():          # an intact bucket stays where it is (captured hipGraphs hold its addresses)
            opt.flatten()
        _, _, m, v, step, _ = opt._flat
        m.copy_(st["m"]); v.copy_(st["v"]); step.copy_(st["step"])
    else:
        opt.load_state_dict(st["state"])
