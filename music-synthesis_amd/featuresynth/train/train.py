"""The alternating D / G train step, drop-in for the reference's featuresynth/train/train.py
(GeneratorTrainer :8-42, DiscriminatorTrainer :45-74, training_loop :77-106): same constructor
signatures, same `train(samples, features)` return dicts.

Two execution paths, same numbers:
  * reference order of operations (any generator / discriminator / optimizer objects);
  * native path, taken when both networks are this package's HIP modules: work that cannot
    change the result is skipped (D-step: no backward through the generator, which the reference
    performs only because `fake` is not detached, train.py:66-71; G-step: no real-path backward
    and no discriminator weight grads, train.py:36), and when both optimizers are FlatAdam the
    whole step (zero_grad, forwards, backward, Adam) is captured once into a hipGraph and replayed;
    for the headline model with its stock losses the step is hand-scheduled (_ops/step.py: no
    autograd engine).  Under torch.distributed the stepped network's flat gradient bucket is
    all-reduced (RCCL) in two slices: the slice that is final early in the backward travels while
    the rest of the backward runs, the other one just before the Adam graph.
"""
import gc
import os
import sys
import warnings
import weakref
from datetime import datetime

import numpy as np
import torch

from .. import _dist
from .._ops import graph as _graph
from .._ops import functional as _F
from .._ops import step as _step
from ..loss import hinge_discriminator_loss, hinge_generator_loss
from ..loss import mel_gan_disc_loss as _mel_gan_disc_loss
from ..loss import mel_gan_gen_loss as _mel_gan_gen_loss
from ..optim import FlatAdam
from ..util.modules import zero_grad


def _native(*modules):
    return all(getattr(m, "_ms_native", False) for m in modules)


def _use_graph():
    return os.environ.get("MSYNTH_GRAPH", "1") != "0"


class _GraphedStep:
    """Runs `body(samples, features, cut) -> dict of device/host tensors`: first call eager (loads the
    code objects, sizes the buckets), second call captures into hipGraphs, later calls copy the
    inputs into the static buffers and replay.

    `between` are eager host actions (the data-parallel all-reduces): the body calls `cut()` exactly
    len(between) times, which ends one graph segment and starts the next; between[k]() runs after segment
    k -- directly from cut() on an eager call, between two graph launches on a replay.  All segments
    share one memory pool and are replayed in capture order."""

    def __init__(self, body, between=()):
        self.body = body
        self.between = list(between)
        self.seen = {}
        self.graphs = {}
        self.disabled = False
        self.capture_error = None     # the exception that demoted this step to eager execution, if any
        self.segment = None           # index of the segment being issued (eager call or capture), else None

    def in_final_segment(self):
        """True unless a body is running in front of a cut that is still to come."""
        return self.segment is None or self.segment == len(self.between)

    def _eager(self, samples, features):
        k = [0]

        def cut():
            self.between[k[0]]()
            k[0] += 1
            self.segment = k[0]
        self.segment = 0
        try:
            out = self.body(samples, features, cut)
        finally:
            self.segment = None
        assert k[0] == len(self.between), "train step: body made %d cuts, %d expected" % (k[0], len(self.between))
        return out

    @staticmethod
    def _new_graph():
        g = torch.cuda.CUDAGraph()
        if os.environ.get("MSYNTH_GRAPH_DOT_DIR"):      # debugging aid: keep the captured hipGraph for a DOT dump
            g.enable_debug_mode()
        return g

    _dumped = [0]

    @classmethod
    def _dump(cls, graphs):
        """MSYNTH_GRAPH_DOT_DIR=<dir>: every captured segment is written as <dir>/step<N>_seg<k>.dot
        (hipGraphDebugDotPrint) -- the topology the runtime replays: branches, cross-stream edges, node kinds."""
        d = os.environ.get("MSYNTH_GRAPH_DOT_DIR")
        if not d:
            return
        os.makedirs(d, exist_ok=True)
        for k, g in enumerate(graphs):
            g.debug_dump(os.path.join(d, "step%d_seg%d.dot" % (cls._dumped[0], k)))
        cls._dumped[0] += 1

    def _capture(self, s_in, f_in):
        # Dead reference cycles may hold hipGraph / tensor objects of earlier trainers: collect them NOW, not at whatever
        # allocation inside the capture trips the collector (torch >= 2.9 no longer does this in torch.cuda.graph.__enter__;
        # destroying graph execs and returning their pools belongs in front of a capture, not inside one)
        gc.collect()
        torch.cuda.synchronize()
        graphs = []
        if not self.between:
            g = self._new_graph()
            # (thread-local error mode: other threads -- the RCCL watchdog polling its events -- stay legal)
            self.segment = 0
            try:
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    out = self.body(s_in, f_in, None)
            finally:
                self.segment = None
            self._dump([g])
            return [g], out
        pool = torch.cuda.graph_pool_handle()
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        cur = [None]

        def begin():
            cur[0] = self._new_graph()
            cur[0].capture_begin(pool=pool, capture_error_mode="thread_local")

        def end():
            cur[0].capture_end()
            graphs.append(cur[0])
            cur[0] = None

        def cut():
            end()
            begin()
            self.segment += 1
        with torch.cuda.stream(stream):
            begin()
            self.segment = 0
            try:
                out = self.body(s_in, f_in, cut)
            finally:
                self.segment = None
                if cur[0] is not None:
                    end()
        torch.cuda.current_stream().wait_stream(stream)
        if len(graphs) != len(self.between) + 1:
            raise RuntimeError("train step: %d graph segments captured, %d expected"
                               % (len(graphs), len(self.between) + 1))
        self._dump(graphs)
        return graphs, out

    def __call__(self, samples, features):
        key = (tuple(samples.shape), tuple(features.shape), samples.device)
        entry = self.graphs.get(key)
        if entry is None:
            n = self.seen.get(key, 0)
            self.seen[key] = n + 1
            if n == 0 or self.disabled or not _use_graph():
                return self._eager(samples, features)
            try:
                s_in, f_in = samples.clone(), features.clone()
                graphs, out = self._capture(s_in, f_in)
                entry = (graphs, s_in, f_in, out)
                self.graphs[key] = entry
            except Exception as e:  # perf-only fallback, reported loudly: a warning once + an attribute
                self.disabled = True            # that loggers can read (trainer.graph_status())
                self.capture_error = e
                torch.cuda.synchronize()
                warnings.warn("featuresynth: hipGraph capture of the train step failed (%s: %s); this trainer "
                              "runs eagerly from now on" % (type(e).__name__, e), RuntimeWarning, stacklevel=2)
                return self._eager(samples, features)
        graphs, s_in, f_in, out = entry
        s_in.copy_(samples)
        f_in.copy_(features)
        for k, g in enumerate(graphs):
            g.replay()
            if k < len(self.between):
                self.between[k]()
        return out


class _TrainerBase(object):
    def __init__(self, generator, g_optim, discriminator, d_optim, loss, sub_loss):
        super().__init__()
        self.sub_loss = sub_loss
        self.loss = loss
        self.d_optim = d_optim
        self.discriminator = discriminator
        self.g_optim = g_optim
        self.generator = generator
        self._runner = None
        self._runner_world = None
        self.debug = None             # tests: a dict here receives the step's intermediate tensors (eager calls)

    def _native_ok(self, samples, features):
        return (_native(self.generator, self.discriminator) and isinstance(samples, torch.Tensor)
                and samples.is_cuda and isinstance(features, torch.Tensor))

    def _stepped_optim(self):
        raise NotImplementedError

    def _stock_losses(self):
        raise NotImplementedError

    def _fwd_bwd(self, samples, features, cut=None, loss_slot=None):
        raise NotImplementedError

    def _direct_ok(self):
        """The hand-scheduled step (_ops/step.py) applies: the headline model with its stock losses."""
        from ..discriminator.melgan import MelGanDiscriminator
        from ..generator.full import MelGanGenerator
        return (os.environ.get("MSYNTH_DIRECT", "1") != "0"
                and type(self.generator) is MelGanGenerator
                and type(self.discriminator) is MelGanDiscriminator
                and isinstance(self.g_optim, FlatAdam) and isinstance(self.d_optim, FlatAdam)
                and self._stock_losses()
                and all(p.requires_grad for p in self.generator.parameters())
                and all(p.requires_grad for p in self.discriminator.parameters()))

    def _result(self, out, device, loss_key, with_fake):
        if out["loss"].is_cuda:                 # torch-optimizer path: plain device tensors
            res = {loss_key: out["loss"].item()}
            if with_fake:
                res['fake'] = out["fake"].cpu().numpy()
            return res
        fake = None
        if with_fake:
            # the generated batch leaves in a pinned block of its own (torch's caching host allocator: no new pinning
            # after the first steps), issued behind the step on its stream; the array handed out IS that block -- no
            # second pass over 1 MB (and its page faults) on the host while the device waits for the next call
            fake = torch.empty(out["fake"].shape, dtype=out["fake"].dtype, pin_memory=True)
            fake.copy_(out["fake"], non_blocking=True)
        torch.cuda.current_stream(device).synchronize()
        loss = float(out["loss"])
        if self._runner is not None and self._runner.between:     # data-parallel plan: the slot holds the rank SUM
            loss /= _dist.world_size()
        res = {loss_key: loss}
        if with_fake:
            res['fake'] = fake.numpy()      # (keeps the block alive; it returns to the cache with the array)
        return res

    def graph_status(self):
        """{'mode': 'graph' | 'eager' | 'eager (capture failed)' | 'unplanned', 'segments': [...], 'error': str | None}
        -- what `training_loop` loggers can print: a failed hipGraph capture demotes the trainer to eager
        execution for the life of the process (a RuntimeWarning is raised once when it happens)."""
        r = self._runner
        if r is None:
            return {"mode": "unplanned", "segments": [], "error": None}
        if r.disabled:
            return {"mode": "eager (capture failed)", "segments": [],
                    "error": "%s: %s" % (type(r.capture_error).__name__, r.capture_error)}
        return {"mode": "graph" if r.graphs else "eager", "segments": [len(e[0]) for e in r.graphs.values()],
                "error": None}

    def _to_host(self, out):
        """Device results -> pinned host buffers with asynchronous copies issued on the step's stream
        (inside the captured graph they become memcpy nodes): the caller then needs ONE stream
        synchronisation instead of a blocking .item() plus a blocking pageable copy of `fake`.

        Only in the FINAL segment of a multi-segment step: the host reads the pinned buffers (and the next
        trainer call overwrites the static inputs) as soon as the stream has drained, which must mean the whole
        step has -- a device-to-host copy node in an earlier segment of a pool-sharing capture sequence is the
        one topology that crashed hipGraphLaunch on ROCm 7.2 (DESIGN_HISTORY.md section 6, "graph faults")."""
        if self._runner is not None and not self._runner.in_final_segment():
            raise RuntimeError("train step: host copies belong to the last graph segment (segment %d of %d)"
                               % (self._runner.segment, len(self._runner.between) + 1))
        if not hasattr(self, "_pins"):
            self._pins = {}
        host = {}
        for k, v in out.items():
            if k == "fake":         # leaves in _result, outside the graph: a fresh pinned block per call
                host[k] = v
                continue
            key = (k, tuple(v.shape))
            if key not in self._pins:
                self._pins[key] = torch.empty(tuple(v.shape), dtype=v.dtype, pin_memory=True)
            self._pins[key].copy_(v, non_blocking=True)
            host[k] = self._pins[key]
        return host

    def _split_point(self, opt):
        """Element offset in the stepped network's flat gradient bucket that separates the slice finished
        LATE in the backward ([0, off): the layers nearest the input) from the slice finished EARLY
        ([off, end)); 0 when the step has no cut point (one all-reduce of the whole bucket)."""
        return 0

    @staticmethod
    def _bucket_in_module_order(opt, net):
        """The cut points of graph.py are defined in MODULE parameter order: the sliced exchange is only
        valid when the optimizer's bucket holds exactly list(net.parameters()), same objects, same order."""
        ps = opt.param_groups[0]["params"]
        ms = list(net.parameters())
        return len(ps) == len(ms) and all(a is b for a, b in zip(ps, ms))

    def _native_step(self, samples, features):
        """world 1: ONE graph (zero_grad, forwards, backward, Adam).
        Data parallel: [graph: zero_grad + forwards + backward up to the cut] -> all-reduce of the early
        slice, asynchronous (RCCL) -> [graph: rest of the backward, overlapping it] -> all-reduce of the
        late slice -> [graph: Adam with grad_scale = 1/world + the copies to the host].
        The loss value rides in the spare float behind the gradient bucket (FlatAdam.loss_slot), i.e. inside the
        early slice's all-reduce: the value returned under data parallelism is the mean over ranks (what a
        single process on the global batch prints, the reference's loggers: evaluate.py:164-165)."""
        opt = self._stepped_optim()
        flat = isinstance(opt, FlatAdam) and isinstance(self.g_optim, FlatAdam) and \
            isinstance(self.d_optim, FlatAdam)
        world = _dist.world_size()
        force_dp = os.environ.get("MSYNTH_DP_FORCE") == "1"      # one-rank rehearsal of the N > 1 control flow
        if not flat:
            out = self._fwd_bwd(samples, features)
            if world > 1:
                for p in opt.param_groups[0]["params"]:
                    if p.grad is not None:
                        _dist.allreduce_sum_(p.grad)
                        p.grad.div_(world)
                out["loss"] = _dist.allreduce_mean_scalar(out["loss"])
            opt.step()
            return out
        opt.grad_scale = 1.0 / world
        # (builds the buckets: _split_point reads their layout.)  A captured graph holds the buckets' addresses: when
        # an optimizer re-homes them (flatten() after .to(device), a checkpoint load into a fresh bucket) the plan is
        # dropped and re-captured
        plan = (world, force_dp, self.g_optim.bucket_version(), self.d_optim.bucket_version())
        if self._runner is not None and self._runner_world != plan:
            self._runner = None                   # the process group appeared / went away, or a bucket moved: re-plan
        if self._runner is None:
            self._runner_world = plan
            # the runner's bodies reach the trainer through a weak reference: trainer -> runner -> closure -> trainer would be
            # a reference cycle, and a dropped trainer's hipGraphs would then be destroyed whenever the cycle collector
            # happens to run (possibly inside another trainer's capture) instead of when the trainer goes
            me = weakref.ref(self)
            if world == 1 and not force_dp:
                def body(s, f, cut):
                    out = me()._fwd_bwd(s, f)
                    opt.step()
                    return me()._to_host(out)
                self._runner = _GraphedStep(body)
            else:
                direct = self._direct_ok()
                off = self._split_point(opt) if direct else 0
                pending = []

                def start_early():
                    # [off, end) + the loss slot; the whole store when the step has no cut point
                    st = opt.grad_store
                    pending.append(_dist.allreduce_sum_async(st[off:] if off else st, force=force_dp))

                def finish():
                    if off:
                        pending.append(_dist.allreduce_sum_async(opt.flat_grads[:off], force=force_dp))
                    for w in pending:
                        w.wait()
                    del pending[:]

                def publish(out):
                    out = dict(out)
                    out["loss"] = opt.loss_slot       # summed over ranks by now; _result divides by world
                    return me()._to_host(out)

                if off:
                    def body(s, f, cut):
                        # _fwd_bwd writes the loss into the slot and calls cut() once, where the early slice is final
                        out = me()._fwd_bwd(s, f, cut, loss_slot=opt.loss_slot)
                        cut()
                        opt.step()
                        return publish(out)
                    self._runner = _GraphedStep(body, [start_early, finish])
                else:
                    def body(s, f, cut):
                        out = me()._fwd_bwd(s, f, loss_slot=opt.loss_slot)
                        cut()
                        opt.step()
                        return publish(out)
                    self._runner = _GraphedStep(body, [lambda: (start_early(), finish())])
        return self._runner(samples, features)


class GeneratorTrainer(_TrainerBase):
    def __init__(self, generator, g_optim, discriminator, d_optim, loss,
                 sub_loss=hinge_generator_loss):
        super().__init__(generator, g_optim, discriminator, d_optim, loss, sub_loss)

    def _stepped_optim(self):
        return self.g_optim

    def _stock_losses(self):
        return self.loss is _mel_gan_gen_loss and self.sub_loss is hinge_generator_loss

    def _split_point(self, opt):
        if opt._flat is None or not self._bucket_in_module_order(opt, self.generator):
            return 0
        return opt._flat[5][_graph.G_TAIL_PARAM][0]

    def _fwd_bwd(self, samples, features, cut=None, loss_slot=None):
        zero_grad(self.g_optim, self.d_optim)
        if self._direct_ok():
            loss, fake = _step.g_step(list(self.generator.parameters()), list(self.discriminator.parameters()),
                                      samples, features, self.discriminator.scales, cut=cut, debug=self.debug,
                                      loss_slot=loss_slot)
            return {"loss": loss, "fake": fake}
        _graph.begin_step()
        d_params = [p for p in self.discriminator.parameters() if p.requires_grad]
        for p in d_params:          # discriminator weight grads are never used by a G-step
            p.requires_grad_(False)
        try:
            # the real path neither depends on the generator nor needs gradients: it runs on a
            # forked stream (a parallel branch of the captured graph) beside G and D(fake)
            main = torch.cuda.current_stream(samples.device)
            side = _graph.aux_stream(samples.device)
            side.wait_stream(main)
            with _graph.forked(side), torch.no_grad():
                r_features, r_score = self.discriminator(samples, features)
            fake = self.generator(features)
            f_features, f_score = self.discriminator(fake, features)
            main.wait_stream(side)
            loss = self.loss(r_features, f_features, r_score, f_score, gan_loss=self.sub_loss)
            loss.backward()
            _graph.join_side_streams(samples.device)      # (modules that fork streams inside their forward: realmelgan)
        finally:
            for p in d_params:
                p.requires_grad_(True)
        if loss_slot is not None:
            loss_slot.copy_(loss.detach())
        return {"loss": loss.detach(), "fake": fake.detach()}

    def train(self, samples, features):
        if self._native_ok(samples, features):
            out = self._native_step(samples, features)
            return self._result(out, samples.device, 'g_loss', True)
        # reference order of operations (train.py:26-42)
        zero_grad(self.g_optim, self.d_optim)
        fake = self.generator(features)
        f_features, f_score = self.discriminator(fake, features)
        r_features, r_score = self.discriminator(samples, features)
        loss = self.loss(r_features, f_features, r_score, f_score, gan_loss=self.sub_loss)
        loss.backward()
        self.g_optim.step()
        if isinstance(fake, dict):
            fake = {k: v.data.cpu().numpy() for k, v in fake.items()}
        else:
            fake = fake.data.cpu().numpy()
        return {'g_loss': loss.item(), 'fake': fake}


class DiscriminatorTrainer(_TrainerBase):
    def __init__(self, generator, g_optim, discriminator, d_optim, loss,
                 sub_loss=hinge_discriminator_loss):
        super().__init__(generator, g_optim, discriminator, d_optim, loss, sub_loss)

    def _stepped_optim(self):
        return self.d_optim

    def _stock_losses(self):
        return self.loss is _mel_gan_disc_loss and self.sub_loss is hinge_discriminator_loss

    def _split_point(self, opt):
        if opt._flat is None or not self._bucket_in_module_order(opt, self.discriminator):
            return 0
        return opt._flat[5][_graph.D_HEAD_PARAM][0]

    def _fwd_bwd(self, samples, features, cut=None, loss_slot=None):
        zero_grad(self.g_optim, self.d_optim)
        if self._direct_ok():
            loss = _step.d_step(list(self.generator.parameters()), list(self.discriminator.parameters()),
                                samples, features, self.discriminator.scales, cut=cut, debug=self.debug,
                                loss_slot=loss_slot)
            return {"loss": loss}
        _graph.begin_step()
        with torch.no_grad():       # generator grads of a D-step are discarded by the reference
            fake = self.generator(features)
        # one discriminator pass over [fake; real]: samples are independent (no batch coupling),
        # so the judgements are the same and the shared weights' gradients are summed in-kernel
        B = fake.shape[0]
        both = torch.cat([fake, samples], 0)
        # a discriminator that consumes its conditioning (experiment/realmelgan.py:128-136,149-152) needs one
        # conditioning row per row of the doubled batch
        cond = features
        if getattr(self.discriminator, "conditioning_channels", 0) > 0:
            cond = torch.cat([features, features], 0)
        _, scores = self.discriminator(both, cond)
        if self._stock_losses():
            loss = _F.MelGanDiscLossCatFn.apply(len(scores), B, *scores)   # no per-slice autograd nodes
        elif isinstance(scores, torch.Tensor):      # single-judgement discriminators (stage 1)
            loss = self.loss(scores[B:], scores[:B], gan_loss=self.sub_loss)
        else:
            f_score = [j[:B] for j in scores]
            r_score = [j[B:] for j in scores]
            loss = self.loss(r_score, f_score, gan_loss=self.sub_loss)
        loss.backward()
        _graph.join_side_streams(samples.device)
        if loss_slot is not None:
            loss_slot.copy_(loss.detach())
        return {"loss": loss.detach()}

    def train(self, samples, features):
        if self._native_ok(samples, features):
            out = self._native_step(samples, features)
            return self._result(out, samples.device, 'd_loss', False)
        # reference order of operations (train.py:63-74)
        zero_grad(self.g_optim, self.d_optim)
        fake = self.generator(features)
        _, f_score = self.discriminator(fake, features)
        _, r_score = self.discriminator(samples, features)
        loss = self.loss(r_score, f_score, gan_loss=self.sub_loss)
        loss.backward()
        self.d_optim.step()
        return {'d_loss': loss.item()}


class _Prefetcher:
    """H2D prefetch (replaces the blocking copy of the reference's
    train.py:85): the next batch is preprocessed, staged in pinned host memory and copied on a side
    stream while the current step runs; the consumer's stream waits on the copy's event only."""

    def __init__(self, batch_stream, experiment, device):
        self.it = iter(batch_stream)
        self.exp = experiment
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.pending = None
        self._fetch()

    def _to_device(self, x):
        if isinstance(x, dict):
            return {k: self._to_device(v) for k, v in x.items()}
        t = torch.from_numpy(np.ascontiguousarray(x))
        if not self.cuda:
            return t.to(self.device).float()
        t = t.float().pin_memory()
        return t.to(self.device, non_blocking=True)

    def _fetch(self):
        try:
            batch = next(self.it)
        except StopIteration:
            self.pending = None
            return
        pre = self.exp.preprocess_batch(batch)
        if self.cuda:
            with torch.cuda.stream(self.stream):
                tensors = [self._to_device(x) for x in pre]
                ev = torch.cuda.Event()
                ev.record(self.stream)
        else:
            tensors, ev = [self._to_device(x) for x in pre], None
        self.pending = (pre, tensors, ev)

    def __iter__(self):
        return self

    def __next__(self):
        if self.pending is None:
            raise StopIteration
        pre, tensors, ev = self.pending
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
            for t in tensors:                     # allocated on the copy stream, consumed on this one
                for u in (t.values() if isinstance(t, dict) else (t,)):
                    u.record_stream(torch.cuda.current_stream(self.device))
        self._fetch()                             # the next copy overlaps the step the caller now runs
        return pre, tensors


def training_loop(batch_stream, experiment, device, loggers):
    """Driver with the contract of the reference's train.py:77-106: numpy batch -> float tensors
    on `device` -> next training step -> loggers; yields (i, elapsed, log_results).  The host-to-device
    copy of batch i+1 is prefetched while step i runs."""
    started = datetime.utcnow()
    for i, (preprocessed, tensors) in enumerate(_Prefetcher(batch_stream, experiment, device)):
        step = next(experiment.training_steps)
        step_result = step(*tensors)
        elapsed = datetime.utcnow() - started
        log_results = {}
        for logger in loggers:
            result = logger(experiment, preprocessed, step_result, i, elapsed)
            if result is not None:
                log_results.update(result)
        yield i, elapsed, log_results
