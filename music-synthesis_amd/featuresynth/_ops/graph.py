"""Whole-network forward / backward schedules over the C-ABI kernels (no autograd).

The layer geometry mirrors the reference exactly:
  generator      /root/reference/featuresynth/generator/full.py:22-45
  residual atom  /root/reference/featuresynth/util/modules.py:350-405 (dilations 1,3,9 hard-coded :397-399)
  discriminator  /root/reference/featuresynth/discriminator/full.py:13-22, melgan.py:13-27
Parameters travel as flat lists in state_dict order.  Every backward fuses the activation
derivative into the kernels' operand loaders (y_act), the skip/feature-gradient adds into the
backward-data epilogues (gx_add) and the cross-scale weight-grad sums into the weight-grad
epilogue (accumulate), so no stand-alone elementwise pass remains.
"""
import contextlib
import os

import torch

from . import lib as L
from . import prims as P

G_UPSAMPLE = ((8, 4), (8, 4), (2, 1), (2, 1))     # (stride, padding) of the 4 ConvTranspose1d
DILATIONS = (1, 3, 9)
G_NPARAMS = 2 + 4 * (2 + 12) + 2
# (stride, padding, groups) of FullDiscriminator.main, then judge
D_LAYERS = ((1, 7, 1), (4, 20, 4), (4, 20, 16), (4, 20, 64), (4, 20, 256), (1, 2, 1))
D_NPARAMS = 14


class GradSink:
    """Destination slots for parameter gradients (state_dict order).  A slot may be preset to a
    view of an optimizer's flat gradient bucket; `acc[i]` says whether the next write must add
    to what is there (a second pass over shared weights) or overwrite it."""

    def __init__(self, n, dests=None, acc=None):
        self.t = list(dests) if dests is not None else [None] * n
        self.acc = list(acc) if acc is not None else [False] * n

    def pair(self, i):
        return self.t[i], self.t[i + 1], (self.acc[i] and self.t[i] is not None)

    def put(self, i, gw, gb):
        self.t[i], self.t[i + 1] = gw, gb
        self.acc[i] = self.acc[i + 1] = True


_FORK_DEPTH = 0     # > 0 while work is being issued on a forked stream (forks stay one level deep)
_FORKED_SINCE_JOIN = set()     # side streams work was issued on since the last join_side_streams()


@contextlib.contextmanager
def forked(stream):
    """`with torch.cuda.stream(stream)` that also counts the fork level: code issued inside must not
    fork again (_WgradFork / _PackAside / fork_aux check `_may_fork`).  Forks are kept ONE
    level deep under hipGraph capture: a fork inside a fork crashed graph instantiation on ROCm 7.2
    (DESIGN_HISTORY.md section 4, "Streams inside the graph")."""
    global _FORK_DEPTH
    _FORK_DEPTH += 1
    _FORKED_SINCE_JOIN.add(stream)
    try:
        with torch.cuda.stream(stream):
            yield
    finally:
        _FORK_DEPTH -= 1


def _may_fork(device):
    if not _concurrent_scales() or _on_aux(device):
        return False
    return _FORK_DEPTH == 0


class _WgradFork:
    """Weight gradients on a side stream (idx: which of the two; MSYNTH_WGSTREAM=0 disables): each layer's
    weight-grad only needs the incoming gradient, so it can run beside the backward-data chain.
    Tensors it reads are kept alive until the join (the caching allocator is per stream)."""

    def __init__(self, device, idx=0):
        self.on = os.environ.get("MSYNTH_WGSTREAM", "2") != "0" and _may_fork(device)
        self.keep = []
        if self.on:
            self.main = torch.cuda.current_stream(device)
            self.side = _side_streams(device, 2)[idx]

    def run(self, fn, *tensors):
        if not self.on:
            return fn()
        self.keep.extend(t for t in tensors if t is not None)
        self.side.wait_stream(self.main)
        with forked(self.side):
            return fn()

    def join(self):
        if self.on:
            self.main.wait_stream(self.side)
            self.keep.clear()


def atom_fused_ok(x_shape, w0, b0, b1, dil):
    """The fused one-launch atom (csrc/atom_fused.hip) takes this geometry and these (16-byte aligned) biases."""
    B, C, Lg = x_shape
    return (tuple(w0.shape) == (C, C, 3) and b0 is not None and b1 is not None and
            b0.data_ptr() % 16 == 0 and b1.data_ptr() % 16 == 0 and P.atom_supported(B, C, Lg, dil))


def atom_forward(h, w0, b0, w1, b1, dil, save, image=None, signs=False):
    """image: the atom's pre-split weight image (P.atom_pack) -> ONE fused launch, the intermediate stays on chip;
    None -> the two row-tile conv launches.
    signs (fused, training): the record holds u as SIGN WORDS and t's in its AtomAux: the caller has checked that the
    whole backward pass of the stack takes them (P.stack_signs_ok).  Record: (d0, d1, h, t, u, aux)."""
    d0, lo = P.conv_desc(h.shape, w0.shape, pad=dil, dil=dil, act=L.ACT_LRELU)
    d1, _ = P.conv_desc(h.shape, w1.shape, pad=1, act=L.ACT_LRELU)
    if image is not None:
        out, t, u, aux = P.atom_fwd(h, image, b0, b1, dil, save, signs=signs and save)
        return out, (d0, d1, h, t, u, aux)
    t, _ = P.conv1d_fwd(h, w0, b0, d0, lo)
    out, u = P.conv1d_fwd(t, w1, b1, d1, lo, residual=h, want_y_act=save)
    return out, (d0, d1, h, t, u, None)


def pack_convt_images(x_shape, params, backward=False):
    """Weight images of the generator's transposed convs that an image kernel takes (forward: csrc/convt_img.hip; backward
    data: csrc/convt_bwd_img.hip) -> {parameter index of the layer's weight: image}."""
    B, _, Lg = x_shape
    images = {}
    i = 2
    for stride, pad in G_UPSAMPLE:
        w = params[i]
        dt, lo = P.convt_desc((B, w.shape[0], Lg), w.shape, stride, pad, act=L.ACT_LRELU)
        if backward and P.convt_bwd_img_bytes(dt):
            images[i] = P.convt_bwd_img_pack(dt, w)
        elif not backward and P.convt_img_bytes(dt):
            images[i] = P.convt_img_pack(dt, w)
        Lg = lo
        i += 2 + 4 * len(DILATIONS)
    return images


def pack_atom_images(x_shape, params, backward=False):
    """Pre-splits the weights of every generator atom the fused kernel takes -- ONE launch per forward (backward) pass
    (the images are only valid for the weights as they are now: they are rebuilt on every pass, 9 MB).
    -> {parameter index of the atom's first weight: image}"""
    B, _, Lg = x_shape
    images, jobs = {}, []
    i = 2
    for stride, pad in G_UPSAMPLE:
        K = params[i].shape[2]
        Lg = (Lg - 1) * stride - 2 * pad + K
        i += 2
        for dil in DILATIONS:
            w0, b0, w1, b1 = params[i], params[i + 1], params[i + 2], params[i + 3]
            if atom_fused_ok((B, w0.shape[0], Lg), w0, b0, b1, dil) and \
                    (not backward or P.atom_bwd_supported(B, w0.shape[0], Lg, dil)):
                images[i] = P.atom_image(w0.shape[0], w0.device)
                jobs.append((w0, w1, images[i]))
            i += 4
    if jobs:
        P.atom_pack(jobs, backward=backward)
    return images


class _PackAside:
    """The weight images of a generator pass, packed on a side stream of its own beside the caller's chain.  The side stream
    forks where this object is made; hipGraphLaunch feeds the device in capture order (step.py), so the caller makes it first,
    issues the head of its chain (conv0), then asks for forward() -- transposed-conv images (the first is needed right behind
    conv0), then the atoms' -- and, once the whole forward chain is issued, for backward(): the backward pass of the same step
    multiplies by the same weights, and its images are thereby off the critical path for good.
    Keys of the image dicts: parameter index of the layer's (first) weight.  Events are None when the packs ran on the
    caller's stream."""
    # (a fork + join costs ~15 us of dependency latency inside a replayed graph: only where the pass is long enough to hide a
    #  pack launch behind -- measured at B = 1: 275 us on one stream, 311 us forked)

    def __init__(self, x_shape, params, device):
        self.x_shape, self.params, self.device = x_shape, params, device
        self.side = None
        if _may_fork(device) and x_shape[0] * x_shape[2] >= 256 and os.environ.get("MSYNTH_PACKASIDE", "1") != "0":
            key = (device.index, "pack")
            if key not in _SIDE_STREAMS:
                _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
            self.side = _SIDE_STREAMS[key]
            self.side.wait_stream(torch.cuda.current_stream(device))

    def _event(self):
        ev = torch.cuda.Event()
        ev.record(self.side)
        return ev

    def _publish(self, images):
        main = torch.cuda.current_stream(self.device)
        for t in images.values():
            t.record_stream(main)       # (allocated on the side stream, read on the caller's)

    def forward(self):
        """-> (images, event behind the transposed convs' images, event behind all of them)"""
        if self.side is None:
            images = pack_convt_images(self.x_shape, self.params)
            images.update(pack_atom_images(self.x_shape, self.params))
            return images, None, None
        with forked(self.side):
            images = pack_convt_images(self.x_shape, self.params)
            ev_t = self._event()
            images.update(pack_atom_images(self.x_shape, self.params))
            ev = self._event()
        self._publish(images)
        return images, ev_t, ev

    def backward(self):
        """-> images of the backward pass; the caller's stream waits for them here (the side stream has had the whole
        forward pass to finish them)."""
        if self.side is None:
            images = pack_atom_images(self.x_shape, self.params, True)
            images.update(pack_convt_images(self.x_shape, self.params, True))
            return images
        with forked(self.side):
            images = pack_atom_images(self.x_shape, self.params, True)
            images.update(pack_convt_images(self.x_shape, self.params, True))
            ev = self._event()
        self._publish(images)
        torch.cuda.current_stream(self.device).wait_event(ev)
        return images


def atom_backward(rec, w0, w1, g, sink, i, need_wgrad=True, need_gx=True, fork=None, batch=None, image_bwd=None):
    """g = d loss / d atom output; parameter grads go to sink slots i..i+3 (w0, b0, w1, b1).
    batch: list collecting the weight-grad jobs (slot, x, gy, y_act, desc, w_shape) instead of running them
    (the caller issues a stack's six jobs as one launch, flush_wgrad_batch).
    image_bwd: the atom's backward weight image -> both backward-data convs in ONE fused launch (atom_fused.hip)."""
    d0, d1, h, t, u, aux = rec
    run = fork.run if fork is not None else (lambda fn, *ts: fn())
    if image_bwd is not None and need_gx:
        gt, gx, ab = P.atom_bwd_data(g, u, t, image_bwd, d0.dil, t_signs=aux.t_signs if aux is not None else None)
        if need_wgrad and batch is not None:
            # operand bounds published by the fused launches: forward [0] = |x|, [1] = |t|; backward [0] = |g|, [1] = |gt lrelu'(t)|
            af = aux.amax if aux is not None else None
            both = af is not None and ab is not None
            # (sign words: the derivative operand of the dilated conv's gradient is t's sign words, not t)
            ta = aux.t_signs if P.is_signs(u) else t
            batch.append((i + 2, t, g, u, d1, w1.shape) + ((af[1], ab[0]) if both else ()))
            batch.append((i, h, gt, ta, d0, w0.shape) + ((af[0], ab[1]) if both else ()))
        elif need_wgrad:
            gw, gb, acc = sink.pair(i + 2)
            run(lambda: sink.put(i + 2, *P.conv1d_bwd_weight(t, g, u, d1, w1.shape, gw, gb, acc)), t, g, u)
            gw, gb, acc = sink.pair(i)
            run(lambda: sink.put(i, *P.conv1d_bwd_weight(h, gt, t, d0, w0.shape, gw, gb, acc)), h, gt, t)
        return gx
    if P.is_signs(u):
        raise RuntimeError("residual atom backward: the forward pass saved sign words, which only the fused backward takes")
    if need_wgrad and batch is not None:
        batch.append((i + 2, t, g, u, d1, w1.shape))
    elif need_wgrad:
        gw, gb, acc = sink.pair(i + 2)
        run(lambda: sink.put(i + 2, *P.conv1d_bwd_weight(t, g, u, d1, w1.shape, gw, gb, acc)), t, g, u)
    gt = P.conv1d_bwd_data(g, u, w1, d1)
    if need_wgrad and batch is not None:
        batch.append((i, h, gt, t, d0, w0.shape))
    elif need_wgrad:
        gw, gb, acc = sink.pair(i)
        run(lambda: sink.put(i, *P.conv1d_bwd_weight(h, gt, t, d0, w0.shape, gw, gb, acc)), h, gt, t)
    if not need_gx:
        return None
    return P.conv1d_bwd_data(gt, t, w0, d0, gx_add=g)   # skip connection: + g


def flush_wgrad_batch(batch, sink, fork):
    """Issues the collected weight-grad jobs of one ResidualStack through the batched C-ABI entry."""
    if not batch:
        return
    jobs, slots, keep = [], [], []
    for job in batch:
        (slot, x, gy, ya, d, w_shape), bounds = job[:6], job[6:]
        gw, gb, acc = sink.pair(slot)
        jobs.append((x, gy, ya, d, w_shape, gw, gb, acc) + tuple(bounds))
        slots.append(slot)
        keep += [x, gy, ya] + list(bounds)

    def go():
        for slot, (gw, gb) in zip(slots, P.conv1d_bwd_weight_multi(jobs)):
            sink.put(slot, gw, gb)
    fork.run(go, *keep)
    batch.clear()


def gen_forward(x, params, save, out=None):
    """out: optional preallocated (B, 1, 256 T) buffer for the waveform (e.g. the leading half of a [fake; real] batch)."""
    if len(params) != G_NPARAMS:
        raise RuntimeError("generator expects %d parameter tensors, got %d" % (G_NPARAMS, len(params)))
    L.require(x, "generator input")
    if x.dim() != 3 or x.shape[1] != params[0].shape[1]:
        raise RuntimeError("generator input must be (B, %d, T), got %s" %
                           (params[0].shape[1], tuple(x.shape)))
    i = 0
    tape = []
    aside = _PackAside(x.shape, params, x.device)
    w, b = params[i], params[i + 1]; i += 2
    d, lo = P.conv_desc(x.shape, w.shape, pad=3, pad_mode=L.PAD_REFLECT, act=L.ACT_LRELU)
    h, _ = P.conv1d_fwd(x, w, b, d, lo)
    tape.append(("conv0", d, x, h))
    images, packed_t, packed = aside.forward()
    for stride, pad in G_UPSAMPLE:
        w, b = params[i], params[i + 1]; i += 2
        dt, lo = P.convt_desc(h.shape, w.shape, stride, pad, act=L.ACT_LRELU)
        hin = h
        if packed_t is not None:
            torch.cuda.current_stream(x.device).wait_event(packed_t)
            packed_t = None
        h = P.convt1d_fwd(hin, w, b, dt, lo, img=images.get(i - 2))
        tape.append(("convT", dt, hin, h))
        if packed is not None:
            torch.cuda.current_stream(x.device).wait_event(packed)
            packed = None
        n = len(DILATIONS)
        if not save and all(images.get(i + 4 * k) is not None for k in range(n)) and P.stack_supported(h, DILATIONS):
            # nothing to save: the whole ResidualStack in one launch, the values between its atoms stay on chip
            h = P.stack_fwd(h, [images[i + 4 * k] for k in range(n)], [params[i + 4 * k + 1] for k in range(n)],
                            [params[i + 4 * k + 3] for k in range(n)], DILATIONS)
            i += 4 * n
            continue
        # training: where the stack's whole backward pass (fused backward data, batched weight gradients) takes them, u is
        # saved as one sign bit per element (and t's signs beside t): a third less traffic over forward + backward
        signs = save and all(images.get(i + 4 * k) is not None for k in range(n)) and P.stack_signs_ok(h, DILATIONS)
        for dil in DILATIONS:
            h, rec = atom_forward(h, params[i], params[i + 1], params[i + 2], params[i + 3], dil, save,
                                  image=images.get(i), signs=signs)
            i += 4
            tape.append(("atom", rec))
    w, b = params[i], params[i + 1]
    d, lo = P.conv_desc(h.shape, w.shape, pad=3, act=L.ACT_TANH)
    y, _ = P.conv1d_fwd(h, w, b, d, lo, out=out)
    tape.append(("last", d, h, y))
    if save:
        # (gen_backward: no pack launch in front of the backward chain.)  The images hold the weights as they were at THIS
        # moment and bypass save_for_backward: the parameters' version counters travel with them
        tape.append(("images_bwd", aside.backward(), _param_versions(params)))
    return y, (tape if save else None)


def _param_versions(params):
    """In-place modification counters of the tensors a weight image was packed from (what autograd's saved-tensor check
    looks at): an image is only used by a backward pass that sees the same versions."""
    return tuple(int(getattr(p, "_version", 0)) for p in params)


G_TAIL_PARAM = 4        # first parameter behind the generator's "tail" (conv0 + the first transposed conv)


def gen_backward(tape, params, gy, sink=None, need_gx=False, cut=None):
    """Fills a GradSink with the parameter grads in state_dict order; returns (sink, gx) where gx =
    d loss / d mel features when need_gx (a stage-1 feature generator in front, BASELINE config 5), else None.

    cut: optional callback, called once every gradient of parameters G_TAIL_PARAM.. is final in the sink
    (all stacks and transposed convs 2-4; 53 % of the bytes) and before the tail (first transposed conv,
    conv0) is processed: the data-parallel trainer starts that slice's all-reduce there (train.py)."""
    L.require(gy, "generator grad_output")
    sink = sink if sink is not None else GradSink(G_NPARAMS)
    i = G_NPARAMS
    g = gy
    gx = None
    fork = _WgradFork(gy.device)
    # (r05: TWO side streams -- the stacks' batched weight gradients on one, the transposed convs' and the first / last conv's on
    #  the other: -0.85 % per trainer call over six interleaved A/B pairs.  MSYNTH_WGSTREAM=1: one side stream, 0: none.
    #  Not under the data-parallel schedule (cut given): there the pass is cut into graph segments at a join of all weight
    #  gradients so far, and the second stream measured +2 % on that path.)
    fork_b = _WgradFork(gy.device, 1) if (fork.on and cut is None and os.environ.get("MSYNTH_WGSTREAM", "2") == "2") else fork
    batch = [] if os.environ.get("MSYNTH_WMULTI", "1") == "1" else None
    deferred = []
    conv0 = tape[0]
    stash = [rec for rec in tape if rec[0] == "images_bwd"]
    if stash and stash[0][2] == _param_versions(params):
        images_bwd = stash[0][1]
    else:           # no stash, or a parameter was modified in place between forward and backward: pack from the weights as they are
        images_bwd = pack_atom_images(conv0[2].shape, params, backward=True)
        images_bwd.update(pack_convt_images(conv0[2].shape, params, True))
    for rec in reversed(tape):
        kind = rec[0]
        if kind == "images_bwd":
            continue
        if kind != "atom" and batch:
            flush_wgrad_batch(batch, sink, fork)     # the stack's six weight gradients: one launch
        if kind == "last":
            _, d, h, y = rec
            i -= 2
            gw, gb, acc = sink.pair(i)
            fork_b.run(lambda: sink.put(i, *P.conv1d_bwd_weight(h, g, y, d, params[i].shape, gw, gb, acc)), h, g, y)
            g = P.conv1d_bwd_data(g, y, params[i], d)
        elif kind == "atom":
            i -= 4
            g = atom_backward(rec[1], params[i], params[i + 2], g, sink, i, fork=fork, batch=batch,
                              image_bwd=images_bwd.get(i))
        elif kind == "convT":
            _, dt, hin, h = rec
            i -= 2
            gw, gb, acc = sink.pair(i)
            if i == 2 and cut is not None:
                fork.join()                          # slots G_TAIL_PARAM.. are complete on the main stream
                if fork_b is not fork:
                    fork_b.join()
                cut()
            if i == 2 and fork.on:
                # the FIRST transposed conv is processed last: by then the side stream still holds the last
                # stack's batched weight gradients while the main stream runs dry, so this one's weight
                # gradient goes to the main stream, issued after conv0's (which forks off first, below)
                def late(i=i, dt=dt, hin=hin, h=h, g=g, gw=gw, gb=gb, acc=acc):
                    sink.put(i, *P.convt1d_bwd_weight(hin, g, h, dt, params[i].shape, gw, gb, acc))
                deferred.append(late)
            else:
                fork_b.run(lambda: sink.put(i, *P.convt1d_bwd_weight(hin, g, h, dt, params[i].shape, gw, gb, acc)), hin, g, h)
            g = P.convt1d_bwd_data(g, h, params[i], dt, img=images_bwd.get(i))
        else:  # conv0 (reflection-padded): the gradient reaches the mel features only on request
            _, d, x, h = rec
            i -= 2
            gw, gb, acc = sink.pair(i)
            fork_b.run(lambda: sink.put(i, *P.conv1d_bwd_weight(x, g, h, d, params[i].shape, gw, gb, acc)), x, g, h)
            if need_gx:
                gx = P.conv1d_bwd_data(g, h, params[i], d)
    if batch:
        flush_wgrad_batch(batch, sink, fork)
    for fn in deferred:
        fn()
    fork.join()
    if fork_b is not fork:
        fork_b.join()
    assert i == 0
    return sink, gx



def pack_k5_image(x_shape, params, backward=False):
    """Weight image of the discriminator's 1024 -> 1024 k5 conv (main.5) for the image kernel (csrc/conv5_img.hip), or
    None when that kernel does not take the layer.  One image serves the three scales and every batch size; it is only
    valid for the weights as they are now (rebuilt once per pass)."""
    w = params[10]
    B = x_shape[0]
    if tuple(w.shape[1:]) != (w.shape[0], 5) or params[11].data_ptr() % 16:
        return None
    d = L.ConvDesc(B, w.shape[1], 32, w.shape[0], 5, 1, 2, 1, 1, L.PAD_ZERO, L.ACT_LRELU, P.SLOPE, L.ACT_NONE)
    if not P.conv_img_bytes(d):
        return None
    return P.conv_img_pack(d, w, backward=backward)


def pack_k5_image_pair(x_shape, params):
    """(forward, backward-data) image of the k5 layer from one pass over its weights (P.conv_img_pack2); (None, None) when the
    image kernel does not take the layer."""
    w = params[10]
    if tuple(w.shape[1:]) != (w.shape[0], 5) or params[11].data_ptr() % 16:
        return None, None
    d = L.ConvDesc(x_shape[0], w.shape[1], 32, w.shape[0], 5, 1, 2, 1, 1, L.PAD_ZERO, L.ACT_LRELU, P.SLOPE, L.ACT_NONE)
    if not P.conv_img_bytes(d):
        return None, None
    return P.conv_img_pack2(d, w)


def fork_aux(device):
    """Fork point of the aux stream, placed BEFORE the caller issues its own chain: what is put on the aux stream later
    (pack_k5_images_aside(forked=True), the G-step's real pass) then depends on nothing the caller issued in between.
    -> whether the aux stream may be used at all."""
    if not _may_fork(device):
        return False
    aux_stream(device).wait_stream(torch.cuda.current_stream(device))
    return True


def pack_k5_images_aside(x_shape, params, device, forked_at=None):
    """Both weight images of the k5 layer (forward, backward data), packed on the aux stream so that they travel beside
    the caller's chain (the generator forward): -> (fwd, bwd, event to wait for before the first use), or
    (None, None, None) when the image kernel does not take the layer / streams are serialised.  forked_at = the result
    of an earlier fork_aux(): the aux stream already branched off there and is not made to wait for the caller again."""
    if not (_may_fork(device) if forked_at is None else forked_at):
        return pack_k5_image_pair(x_shape, params) + (None,)
    main, aux = torch.cuda.current_stream(device), aux_stream(device)
    if forked_at is None:
        aux.wait_stream(main)
    with forked(aux):
        f, b = pack_k5_image_pair(x_shape, params)
        ev = torch.cuda.Event()
        ev.record(aux)
    for t in (f, b):
        if t is not None:
            t.record_stream(main)       # (allocated on aux, read on the caller's stream and its forks)
    return f, b, ev


D_HEAD_PARAM = 10      # first parameter of the discriminator's "head" (main.5 = the 1024 -> 1024 k5 conv, then judge)


_SIDE_STREAMS = {}


def aux_stream(device):
    """Side stream for work that is independent of the caller's stream (the G-step's real pass)."""
    key = (device.index, "aux")
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


def _side_streams(device, n):
    key = (device.index, n, _FORK_DEPTH)
    if key not in _SIDE_STREAMS:   # created on the first (eager) call, never during graph capture
        _SIDE_STREAMS[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
    return _SIDE_STREAMS[key]


def begin_step():
    """Forks are tracked per trainer step: whatever earlier steps (the hand-scheduled path forks too and never calls
    join_side_streams) left in the set is not this step's business."""
    _FORKED_SINCE_JOIN.clear()


def join_side_streams(device):
    """The caller's stream waits for every side stream of this device that work was issued on since begin_step().  The
    autograd engine runs a node's backward on the stream its forward ran on and syncs only what it accumulates itself: a
    Function that wrote parameter gradients straight into FlatAdam slots from a forked stream (bias gradients of convs
    issued under `forked`) is invisible to it, so the trainers join explicitly between loss.backward() and
    optimizer.step().  Under hipGraph capture only streams that are part of THIS capture are joined: recording an event
    on a stream outside the capture and waiting for it from the capturing stream is not a captured dependency at all."""
    main = torch.cuda.current_stream(device)
    capturing = torch.cuda.is_current_stream_capturing()
    for st in list(_FORKED_SINCE_JOIN):
        if st.device != device:
            continue
        _FORKED_SINCE_JOIN.discard(st)
        if st == main:
            continue
        if capturing:
            with torch.cuda.stream(st):
                if not torch.cuda.is_current_stream_capturing():
                    continue
        main.wait_stream(st)


def _on_aux(device):
    return _FORK_DEPTH == 0 and torch.cuda.current_stream(device) == aux_stream(device)


def _concurrent_scales():
    return os.environ.get("MSYNTH_STREAMS", "1") != "0"


def melgan_forward(x, params, scales=2, k5_image=None):
    """MelGanDiscriminator (reference discriminator/melgan.py:13-27): the ONE FullDiscriminator on x, pool(x), pool(pool(x)),
    layer by layer over all scales -- the weights are shared by construction, so a layer's three scales are one piece of
    work for the device (ms_conv1d_parts_*: one launch where a parts kernel takes the layer, part by part on the caller's
    stream otherwise; r01-r04 ran one pass per scale on forked streams).  -> (features[scale][6], judgements[scale], ctx);
    scales = 0: the single FullDiscriminator (discriminator/full.py:24-40)."""
    if len(params) != D_NPARAMS:
        raise RuntimeError("discriminator expects %d parameter tensors, got %d" % (D_NPARAMS, len(params)))
    L.require(x, "discriminator input")
    if x.dim() != 3 or x.shape[1] != 1:
        raise RuntimeError("discriminator input must be (B, 1, L), got %s" % (tuple(x.shape),))
    xs = [x]
    for s in range(scales):
        xs.append(P.avg_pool_fwd(xs[-1]))
    n = len(xs)
    if k5_image is None:
        k5_image = pack_k5_image(x.shape, params)
    feats, tapes = [[] for _ in range(n)], [[] for _ in range(n)]
    hs = xs
    for li, (stride, pad, groups) in enumerate(D_LAYERS):
        w, b = params[2 * li], params[2 * li + 1]
        ds = [P.conv_desc(h.shape, w.shape, stride=stride, pad=pad, groups=groups, act=L.ACT_LRELU)[0] for h in hs]
        img = k5_image if (li == 5 and k5_image is not None and P.conv_img_bytes(ds[0])) else None
        ys = P.conv1d_parts_fwd(hs, w, b, ds[0], image=img)
        for s in range(n):
            tapes[s].append((ds[s], hs[s], ys[s]))
            feats[s].append(ys[s])
        hs = ys
    w, b = params[12], params[13]
    djs = [P.conv_desc(h.shape, w.shape, pad=1)[0] for h in hs]
    js = P.conv1d_parts_fwd(hs, w, b, djs[0])
    for s in range(n):
        tapes[s].append((djs[s], hs[s], js[s]))
    return feats, js, (tapes, xs)


def melgan_backward(ctx, params, g_feats, g_judges, sink=None, need_gx=True, need_wgrad=True, cut=None,
                          k5_image_bwd=None):
    """Backward of melgan_forward, layer by layer over all scales: backward data of a layer's scales is one launch,
    and so is its weight gradient -- summed over the scales by the launch's own reduction.  The gradients may cover fewer
    batch rows than the forward pass saved (the G-step runs one pass over [fake; real] and differentiates the fake half):
    the leading rows of the saved tensors are used.  cut: called once the head's parameters (k5 layer + judge conv) are final."""
    tapes, xs = ctx
    n = len(tapes)
    sink = sink if sink is not None else GradSink(D_NPARAMS)
    k5_bwd = k5_image_bwd if k5_image_bwd is not None else pack_k5_image(xs[0].shape, params, backward=True)

    def own(s, li):
        return g_feats[s][li] if (g_feats is not None and g_feats[s] is not None and li >= 0) else None

    # the weight gradients only need the chain's gradients, never the other way round: they run on side streams beside the
    # backward-data chain (MSYNTH_DWGSTREAM=0: on the caller's stream)
    # (r05: on TWO side streams -- the head's (k5 layer, judge conv: final first, the data-parallel cut waits for them) on one,
    #  the grouped layers' and the first conv's on the other: -1 % per trainer call over ten interleaved A/B pairs, the D-step's
    #  weight-gradient chain (0.46 ms) is longer than the backward-data chain it runs beside (0.27).  MSYNTH_DWGSTREAM=1: one)
    mode = os.environ.get("MSYNTH_DWGSTREAM", "2")
    fork = _WgradFork(xs[0].device) if (need_wgrad and mode != "0") else None
    fork2 = _WgradFork(xs[0].device, 1) if (fork is not None and mode == "2" and cut is None) else None   # (single-graph step only)

    def wgrad(slot, live, xs_, gys, yas, d, w_shape):
        gw, gb, acc = sink.pair(slot)
        a, b_, c = [xs_[s] for s in live], [gys[s] for s in live], (None if yas is None else [yas[s] for s in live])

        def go():
            sink.put(slot, *P.conv1d_parts_bwd_weight(a, b_, c, d, w_shape, gw, gb, acc))
        if fork is not None:
            (fork2 if (fork2 is not None and slot < D_HEAD_PARAM) else fork).run(go, *(a + b_ + (c or [])))
        else:
            go()

    g = [None] * n
    # judge conv: g = total gradient w.r.t. feature 5 (judge path + the loss's own term on that feature)
    live = [s for s in range(n) if g_judges is not None and g_judges[s] is not None]
    if live:
        dj = tapes[live[0]][6][0]
        h5 = [tapes[s][6][1] for s in range(n)]
        if need_wgrad:
            wgrad(12, live, h5, g_judges, None, dj, params[12].shape)
        outs = P.conv1d_parts_bwd_data([g_judges[s] for s in live], None, params[12], dj, [h5[s].shape for s in live],
                                       gx_adds=[own(s, 5) for s in live])
        for s, o in zip(live, outs):
            g[s] = o
    for s in range(n):
        if g[s] is None:
            g[s] = own(s, 5)
    for li in range(5, -1, -1):
        live = [s for s in range(n) if g[s] is not None]
        prev = [own(s, li - 1) if li > 0 else None for s in range(n)]
        if live:
            d = tapes[live[0]][li][0]
            hin = [tapes[s][li][1] for s in range(n)]
            h = [tapes[s][li][2] for s in range(n)]
            if need_wgrad:
                wgrad(2 * li, live, hin, g, h, d, params[2 * li].shape)
            if li > 0 or need_gx:
                img = k5_bwd if (li == 5 and k5_bwd is not None and P.conv_img_bytes(d)) else None
                outs = P.conv1d_parts_bwd_data([g[s] for s in live], [h[s] for s in live], params[2 * li], d,
                                               [hin[s].shape for s in live], gx_adds=[prev[s] for s in live], image_bwd=img)
                for s, o in zip(live, outs):
                    g[s] = o
            else:
                for s in live:
                    g[s] = None
        for s in range(n):
            if s not in live:
                g[s] = prev[s]
        if li == 5 and cut is not None:
            if fork is not None:
                fork.join()             # the head's parameter gradients are final on the caller's stream
            cut()
    if fork is not None:
        fork.join()
    if fork2 is not None:
        fork2.join()
    gx_next = None
    for s in range(n - 1, -1, -1):
        gx = g[s]
        if gx_next is not None and need_gx:
            B = gx_next.shape[0]
            gx = P.avg_pool_bwd(gx_next, (B,) + tuple(xs[s].shape[1:]), gx_add=gx)
        gx_next = gx
    return (gx_next if need_gx else None), sink
