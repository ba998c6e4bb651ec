"""Hand-scheduled D-step / G-step of the headline model: the forward and backward schedules of
graph.py and the fused loss kernels called back to back, no autograd engine in between.

What the reference executes per trainer call (train/train.py:26-42,63-74) minus the work whose
results it discards (D-step: generator backward; G-step: real-path backward, discriminator weight
gradients).  The trainers take this path when both networks are the native MelGanGenerator /
MelGanDiscriminator, the losses are the stock hinge / feature-matching ones and both optimizers are
FlatAdam (train.py); anything else goes through the autograd Functions of functional.py, which run
the same kernels.

Every function issues on the current stream (+ the fork/join side streams of graph.py) and is
hipGraph-capturable.  `cut` (optional callback) marks the point where the "early" slice of the
stepped network's gradient bucket is final, so the data-parallel trainer can start its all-reduce
under the rest of the backward:
    D-step: after every scale's judge + k5-layer weight gradients  (parameters D_HEAD_PARAM.. = 93 %)
    G-step: before the first transposed conv / conv0 are processed  (parameters G_TAIL_PARAM.. = 47 %)
"""
import torch

from . import functional as F_
from . import graph as G

_ONE = {}


def _one(device):
    t = _ONE.get(device)
    if t is None:
        t = torch.ones((), dtype=torch.float32, device=device)
        _ONE[device] = t
    return t


def _both_buffer(samples):
    """(2 B, 1, L) buffer whose trailing half already holds the real samples (one copy instead of a concatenation of both halves)."""
    B = samples.shape[0]
    both = torch.empty((2 * B,) + tuple(samples.shape[1:]), dtype=samples.dtype, device=samples.device)
    both[B:].copy_(samples)
    return both


def _slots(params):
    """Gradient destinations = the parameters' FlatAdam bucket views (zeroed by zero_grad: accumulate)."""
    slots = [F_._bound_slot(p) for p in params]
    if any(s is None for s in slots):
        raise RuntimeError("native train step: every parameter must be bound to a FlatAdam gradient bucket "
                           "(call optim.zero_grad() before the step)")
    return G.GradSink(len(slots), slots, [True] * len(slots))


def d_step(gen_params, disc_params, samples, features, scales=2, cut=None, debug=None, loss_slot=None):
    """-> d_loss (0-d device tensor); discriminator gradients accumulate into its bucket."""
    dev = samples.device
    # hipGraphLaunch feeds the device node by node, in capture order, at ~3.4 us per node (r04, 130 + 170 nodes: 0.43 /
    # 0.59 ms of host time per launch) and the caller synchronises after every step: whatever the generator forward --
    # the head of the critical path -- does not need is issued AFTER it, on a branch forked before it
    forked_at = G.fork_aux(dev)
    # [fake; real] is ONE buffer: the generator's last conv writes its leading half, the samples are copied behind it
    both = _both_buffer(samples)
    fake, _ = G.gen_forward(features, gen_params, save=False, out=both[:samples.shape[0]])
    k5f, k5b, k5ev = G.pack_k5_images_aside(samples.shape, disc_params, dev, forked_at=forked_at)
    B = fake.shape[0]
    if k5ev is not None:
        torch.cuda.current_stream(dev).wait_event(k5ev)
    # one discriminator pass over [fake; real]: samples are independent (no batch coupling), so the
    # judgements are the same and the shared weights' gradients are summed in-kernel.  (r03 tried the real half of the
    # forward on the aux stream under the generator, both halves writing into shared full-batch buffers: 3.71 vs 3.66 ms
    # per step -- the half-batch passes cost more than the overlap returns; dropped.)
    _, judges, ctx = G.melgan_forward(both, disc_params, scales, k5_image=k5f)
    loss = F_.disc_loss_cat_fwd(judges, B)
    if loss_slot is not None:          # data parallel: the value travels with the gradient slice behind the cut
        loss_slot.copy_(loss)
    gjs = F_.disc_loss_cat_bwd(judges, B, _one(samples.device))
    if debug is not None:
        debug.update(fake=fake, judges=judges, disc_ctx=ctx, gjs=gjs)
    G.melgan_backward(ctx, disc_params, None, gjs, _slots(disc_params), need_gx=False, need_wgrad=True, cut=cut,
                      k5_image_bwd=k5b)
    return loss


def g_step(gen_params, disc_params, samples, features, scales=2, weight=10.0, cut=None, debug=None,
           loss_slot=None):
    """-> (g_loss, fake); generator gradients accumulate into its bucket."""
    dev = samples.device
    main = torch.cuda.current_stream(dev)
    # the aux branch forks here and is filled after the generator forward (capture order = launch order, see d_step)
    fork_real = G.fork_aux(dev)
    both = _both_buffer(samples)
    fake, tape = G.gen_forward(features, gen_params, save=True, out=both[:samples.shape[0]])
    # one pair of weight images for the discriminator pass(es) and the backward, packed on the aux stream beside the
    # generator forward
    k5, k5b, k5ev = G.pack_k5_images_aside(samples.shape, disc_params, dev, forked_at=fork_real)
    # ONE discriminator pass over [fake; real], as in the D-step (samples are independent): every layer runs once over both
    # halves and all scales; the backward pass differentiates the fake half (the leading rows of what was saved).  (r01-r04 ran
    # the real half on a forked stream beside the generator forward, one pass per scale.)
    if k5ev is not None:
        main.wait_event(k5ev)
    B = fake.shape[0]
    feats, judges, ctx = G.melgan_forward(both, disc_params, scales, k5_image=k5)
    f_feats = [[t[:B] for t in grp] for grp in feats]
    r_feats = [[t[B:] for t in grp] for grp in feats]
    f_judges = [j[:B] for j in judges]
    rows = slice(0, B)
    S, Lyr = len(f_feats), len(f_feats[0])
    rf = [t for grp in r_feats for t in grp]
    ff = [t for grp in f_feats for t in grp]
    # the loss is the root of this step's backward pass (upstream gradient 1): the feature-matching gradients come out of
    # the same pass over the 18 map pairs that sums the loss
    root = []
    loss, fscale = F_.gen_loss_fwd(S, Lyr, weight, rf, ff, f_judges, root_grads=root)
    if loss_slot is not None:
        loss_slot.copy_(loss)
    fused = len(root) == len(ff)
    _, g_ff, g_fj = F_.gen_loss_bwd(S, fscale, rf, ff, f_judges, _one(dev), [False] * len(rf),
                                    [not fused] * len(ff), [True] * S)
    if fused:
        g_ff = root
    g_feats = [g_ff[Lyr * s:Lyr * s + Lyr] for s in range(S)]
    gx, _ = G.melgan_backward(ctx, disc_params, g_feats, g_fj, None, need_gx=True, need_wgrad=False, k5_image_bwd=k5b)
    if debug is not None:
        debug.update(fake=fake, gen_tape=tape, disc_ctx=ctx, disc_rows=rows, r_feats=r_feats, f_feats=f_feats,
                     f_judges=f_judges, g_fake=gx)
    G.gen_backward(tape, gen_params, gx, _slots(gen_params), cut=cut)
    return loss, fake
