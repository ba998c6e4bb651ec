"""torch.autograd.Function wrappers over the C-ABI kernels.

Coarse-grained on purpose: one Function per network (generator, multi-scale discriminator) and
one per composite loss, so that autograd only stitches G -> D -> loss together and every
arithmetic pass (including gradient accumulation across the three discriminator scales and the
skip / feature-matching gradient adds) runs in the hand-written HIP kernels.  Fine-grained
Functions (conv, transposed conv, residual atom, pool, scalar losses) back the individual
nn.Modules so each of them also works on its own.
"""
import torch
from torch.autograd import Function

from . import graph as G
from . import lib as L
from . import prims as P


def _c(g):
    return None if g is None else (g if g.is_contiguous() else g.contiguous())


def _bound_slot(p):
    """p's view of its FlatAdam gradient bucket, but only while p.grad IS that view: after a plain
    `net.zero_grad()` (torch sets .grad = None) the bucket is no longer where autograd's consumers look,
    so the gradient must travel through autograd like any other (optim.FlatAdam gathers it in step())."""
    slot = getattr(p, "_ms_slot", None)
    if slot is None or p.grad is None or p.grad.data_ptr() != slot.data_ptr():
        return None
    return slot


def _sink_for(params, needs):
    """GradSink for `params`.  A parameter bound to a FlatAdam bucket (optim.py) carries
    `_ms_slot`, its view of the flat gradient bucket: the weight-grad kernels accumulate straight
    into it (torch semantics: .grad accumulates until zero_grad) and the gradient is not handed
    to autograd, so no AccumulateGrad add kernels run."""
    dests, acc, direct = [], [], []
    for p, need in zip(params, needs):
        slot = _bound_slot(p) if need else None
        dests.append(slot)
        acc.append(slot is not None)
        direct.append(slot is not None)
    return G.GradSink(len(dests), dests, acc), direct


def _grads_out(sink, direct, needs):
    return tuple(None if (d or not n) else t for t, d, n in zip(sink.t, direct, needs))


class Conv1dFn(Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, pad, dil, groups, pad_mode, act):
        d, lo = P.conv_desc(x.shape, w.shape, stride, pad, dil, groups, pad_mode, act)
        y, _ = P.conv1d_fwd(x, w, b, d, lo)
        ctx.d = d
        ctx.save_for_backward(x, w, y)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        gy = _c(gy)
        d = ctx.d
        ya = y if d.act != L.ACT_NONE else None
        gx = gw = gb = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw, gb = P.conv1d_bwd_weight(x, gy, ya, d, w.shape, want_bias=ctx.has_bias)
        if ctx.needs_input_grad[0]:
            gx = P.conv1d_bwd_data(gy, ya, w, d)
        return gx, gw, (gb if ctx.has_bias else None), None, None, None, None, None, None


class ConvTranspose1dFn(Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, pad, act):
        d, lo = P.convt_desc(x.shape, w.shape, stride, pad, act)
        y = P.convt1d_fwd(x, w, b, d, lo)
        ctx.d = d
        ctx.save_for_backward(x, w, y)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        gy = _c(gy)
        d = ctx.d
        ya = y if d.act != L.ACT_NONE else None
        gx = gw = gb = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw, gb = P.convt1d_bwd_weight(x, gy, ya, d, w.shape)
        if ctx.needs_input_grad[0]:
            gx = P.convt1d_bwd_data(gy, ya, w, d)
        return gx, gw, (gb if ctx.has_bias else None), None, None, None


class ConvTranspose2dLinesFn(Function):
    """ConvTranspose2d of the stage-1 generator on lines (util/modules.py:HipConvTranspose2d; reference
    featuregenerator/upscale.py:85-99): (B, H, Cin, W) -> (B, sH*H, Cout, 2W).  Output row sH*q + phase is a 1-D transposed
    conv (k4, s2, p1) over the channels of the input rows q + dy; the rows are stacked by ONE stream kernel for all phases
    (csrc/lines.hip), every phase is one transposed-conv launch writing its slab of a (phases, B*H, Cout, 2W) buffer, one
    stream kernel puts the slabs into image row order.  Backward: split the gradient rows by phase, the transposed conv's
    backward kernels per phase, one fold kernel sums the overlapping input rows.
    phases: ((dy, ...), (ky, ...)) per output-row phase: input-row offsets and the kernel rows they go through."""

    @staticmethod
    def forward(ctx, xl, weight, bias, phases, act):
        B, H, C, W = xl.shape
        Cout = weight.shape[1]
        nph = len(phases)
        ld = P.lines_desc(xl.shape, [list(dy) for dy, _ in phases])
        stack = P.lines_stack(xl, ld)                                              # (nph, B*H, nt*C, W)
        ws = [torch.cat([weight[:, :, k, :] for k in ky], dim=0).contiguous() for _, ky in phases]    # (nt*C, Cout, 4): weights only
        slabs = torch.empty((nph, B * H, Cout, 2 * W), dtype=torch.float32, device=xl.device)
        descs = []
        for ph in range(nph):
            d, lo = P.convt_desc(stack[ph].shape, ws[ph].shape, 2, 1, act)
            P.convt1d_fwd(stack[ph], ws[ph], bias, d, lo, out=slabs[ph])
            descs.append(d)
        if nph == 1:
            y = slabs.reshape(B, H, Cout, 2 * W)
        else:
            y = P.lines_interleave(slabs, B * H, nph, Cout * 2 * W).reshape(B, nph * H, Cout, 2 * W)
        ctx.ld, ctx.descs, ctx.phases = ld, descs, phases
        ctx.has_bias = bias is not None
        ctx.save_for_backward(stack, weight, slabs, *ws)
        return y

    @staticmethod
    def backward(ctx, gy):
        stack, weight, slabs = ctx.saved_tensors[:3]
        ws = ctx.saved_tensors[3:]
        ld, descs, phases = ctx.ld, ctx.descs, ctx.phases
        nph = len(phases)
        C = ld.C
        rows, n = ld.B * ld.H, slabs.shape[2] * slabs.shape[3]
        gy = _c(gy)
        gs = gy.reshape(slabs.shape) if nph == 1 else P.lines_interleave(gy, rows, nph, n, inverse=True).reshape(slabs.shape)
        gx = gw = gb = None
        need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        if need_w:
            gw = torch.zeros_like(weight)
        gstack = torch.empty_like(stack) if ctx.needs_input_grad[0] else None
        for ph, (_, ky) in enumerate(phases):
            d = descs[ph]
            ya = slabs[ph] if d.act != L.ACT_NONE else None
            if need_w:
                gwp, gbp = P.convt1d_bwd_weight(stack[ph], gs[ph], ya, d, ws[ph].shape)
                for j, k in enumerate(ky):                      # (weights only: kernel row k of the 2-D weight)
                    gw[:, :, k, :] += gwp[j * C:(j + 1) * C]
                gb = gbp if gb is None else gb + gbp
            if gstack is not None:
                P.convt1d_bwd_data(gs[ph], ya, ws[ph], d, out=gstack[ph])
        if gstack is not None:
            gx = P.lines_fold(gstack, ld)
        return gx, gw, (gb if ctx.has_bias else None), None, None


class ResidualAtomFn(Function):
    """x + lrelu(conv_k3(lrelu(conv_k3_dilated(x))))   (util/modules.py:384-388)"""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1, dil):
        image = None
        if x.is_cuda and G.atom_fused_ok(x.shape, w0, b0, b1, dil):      # one fused launch (csrc/atom_fused.hip)
            image = P.atom_image(w0.shape[0], x.device)
            P.atom_pack([(w0, w1, image)])
        out, rec = G.atom_forward(x, w0, b0, w1, b1, dil, save=True, image=image)
        ctx.rec = rec
        ctx.w = (w0, w1)
        ctx.dil = dil
        return out

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        sink = G.GradSink(4)
        need_w = any(ctx.needs_input_grad[1:5])
        w0, w1 = ctx.w
        image_bwd = None
        if ctx.needs_input_grad[0] and G.atom_fused_ok(g.shape, w0, w0, w0, ctx.dil) and \
                P.atom_bwd_supported(g.shape[0], g.shape[1], g.shape[2], ctx.dil):
            image_bwd = P.atom_image(w0.shape[0], g.device)
            P.atom_pack([(w0, w1, image_bwd)], backward=True)
        gx = G.atom_backward(ctx.rec, w0, w1, g, sink, 0, need_wgrad=need_w,
                             need_gx=ctx.needs_input_grad[0], image_bwd=image_bwd)
        return (gx,) + tuple(sink.t) + (None,)


class AddActFn(Function):
    """act(a + b): DilatedStack's residual layer applies the activation over the skip sum."""

    @staticmethod
    def forward(ctx, a, b, act):
        y = P.add_act(a, b, act)
        ctx.act = act
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        g = P.act_bwd(y, _c(gy), ctx.act)
        return (g if ctx.needs_input_grad[0] else None), (g if ctx.needs_input_grad[1] else None), None


class AvgPoolFn(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        return P.avg_pool_fwd(x)

    @staticmethod
    def backward(ctx, gy):
        return P.avg_pool_bwd(_c(gy), ctx.shape)


class GeneratorFn(Function):
    """Whole MelGanGenerator (generator/full.py:47-50) as one autograd node."""

    @staticmethod
    def forward(ctx, x, *params):
        save = any(ctx.needs_input_grad)
        y, tape = G.gen_forward(x, params, save)
        ctx.tape = tape
        ctx.params = params
        return y

    @staticmethod
    def backward(ctx, gy):
        if ctx.tape is None:
            return (None,) * (1 + len(ctx.params))
        needs = ctx.needs_input_grad[1:]
        sink, direct = _sink_for(ctx.params, needs)
        _, gx = G.gen_backward(ctx.tape, ctx.params, _c(gy), sink, need_gx=ctx.needs_input_grad[0])
        ctx.tape = None
        return (gx,) + _grads_out(sink, direct, needs)


class MelGanDiscFn(Function):
    """MelGanDiscriminator (discriminator/melgan.py:13-27): returns 3x6 features then 3 judgements."""

    @staticmethod
    def forward(ctx, x, scales, *params):
        feats, judges, tctx = G.melgan_forward(x, params, scales)
        ctx.tctx = tctx
        ctx.params = params
        ctx.nscale = scales + 1
        ctx.set_materialize_grads(False)
        flat = [f for group in feats for f in group] + list(judges)
        return tuple(flat)

    @staticmethod
    def backward(ctx, *gouts):
        n = ctx.nscale
        gouts = [_c(g) for g in gouts]
        g_feats = [gouts[6 * s:6 * s + 6] for s in range(n)]
        g_judges = gouts[6 * n:6 * n + n]
        need_gx = ctx.needs_input_grad[0]
        need_w = any(ctx.needs_input_grad[2:])
        needs = ctx.needs_input_grad[2:]
        sink, direct = _sink_for(ctx.params, needs)
        gx, _ = G.melgan_backward(ctx.tctx, ctx.params, g_feats, g_judges, sink, need_gx=need_gx,
                                  need_wgrad=need_w)
        ctx.tctx = None
        return (gx, None) + _grads_out(sink, direct, needs)


# ----------------------------------------------------------------------- losses

class HingeDFn(Function):
    @staticmethod
    def forward(ctx, r, f):
        ctx.save_for_backward(r, f)
        return P.hinge_d_fwd(r, f)

    @staticmethod
    def backward(ctx, g):
        r, f = ctx.saved_tensors
        gr, gf = P.hinge_d_bwd(r, f, _c(g), 1.0, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gr, gf


class NegMeanFn(Function):
    @staticmethod
    def forward(ctx, f):
        ctx.save_for_backward(f)
        return P.neg_mean_fwd(f)

    @staticmethod
    def backward(ctx, g):
        (f,) = ctx.saved_tensors
        return P.neg_mean_bwd(f, _c(g))


class L1MeanFn(Function):
    """F.l1_loss(r, f)."""

    @staticmethod
    def forward(ctx, r, f):
        ctx.save_for_backward(r, f)
        return P.l1_mean_fwd(r, f)

    @staticmethod
    def backward(ctx, g):
        r, f = ctx.saved_tensors
        g = _c(g)
        gr = P.l1_mean_bwd(f, r, g) if ctx.needs_input_grad[0] else None
        gf = P.l1_mean_bwd(r, f, g) if ctx.needs_input_grad[1] else None
        return gr, gf


class LsGFn(Function):
    @staticmethod
    def forward(ctx, j):
        ctx.save_for_backward(j)
        return P.ls_g_fwd(j)

    @staticmethod
    def backward(ctx, g):
        (j,) = ctx.saved_tensors
        return P.ls_g_bwd(j, _c(g))


class LsDFn(Function):
    @staticmethod
    def forward(ctx, r, f):
        ctx.save_for_backward(r, f)
        return P.ls_d_fwd(r, f)

    @staticmethod
    def backward(ctx, g):
        r, f = ctx.saved_tensors
        return P.ls_d_bwd(r, f, _c(g))


_COEF_CACHE = {}


def _coef(values, device):
    key = (tuple(values), device)
    t = _COEF_CACHE.get(key)
    if t is None:
        t = torch.tensor(values, dtype=torch.float32, device=device)
        _COEF_CACHE[key] = t
    return t


class MelGanDiscLossFn(Function):
    """sum_s hinge_discriminator_loss(r_s, f_s)   (loss/loss.py:21-25 with :17-18)"""

    @staticmethod
    def forward(ctx, n, *js):
        rs, fs = js[:n], js[n:]
        terms = torch.empty((n,), dtype=torch.float32, device=rs[0].device)
        for s in range(n):
            P.hinge_d_fwd(rs[s], fs[s], terms[s])
        ctx.n = n
        ctx.js = js
        return P.weighted_sum(terms, _coef([1.0] * n, terms.device))

    @staticmethod
    def backward(ctx, g):
        n, js = ctx.n, ctx.js
        g = _c(g)
        grs, gfs = [], []
        for s in range(n):
            gr, gf = P.hinge_d_bwd(js[s], js[n + s], g, 1.0, ctx.needs_input_grad[1 + s],
                                   ctx.needs_input_grad[1 + n + s])
            grs.append(gr); gfs.append(gf)
        return (None,) + tuple(grs) + tuple(gfs)


def disc_loss_cat_fwd(js, B):
    """sum_s hinge_discriminator_loss(real = j_s[B:], fake = j_s[:B]) -> 0-d device tensor."""
    n = len(js)
    if P.judge_multi_ok(js):                 # all scales in one launch
        out = torch.empty((), dtype=torch.float32, device=js[0].device)
        return P.judge_loss_multi_fwd(L.JUDGE_HINGE_D, [j[B:] for j in js], [j[:B] for j in js], out)
    terms = torch.empty((n,), dtype=torch.float32, device=js[0].device)
    for s in range(n):
        P.hinge_d_fwd(js[s][B:], js[s][:B], terms[s])
    return P.weighted_sum(terms, _coef([1.0] * n, terms.device))


def disc_loss_cat_bwd(js, B, g, need=None):
    """Gradients of the loss above w.r.t. each j_s (both halves written into one tensor)."""
    n = len(js)
    need = [True] * n if need is None else need
    if P.judge_multi_ok(js) and all(need):
        gjs = [torch.empty_like(j) for j in js]
        P.judge_loss_multi_bwd(L.JUDGE_HINGE_D, [j[B:] for j in js], [j[:B] for j in js], g, 1.0,
                               [gj[B:] for gj in gjs], [gj[:B] for gj in gjs])
        return gjs
    outs = []
    for s in range(n):
        j = js[s]
        if not need[s]:
            outs.append(None)
            continue
        gj = torch.empty_like(j)
        P.hinge_d_bwd(j[B:], j[:B], g, 1.0, gr=gj[B:], gf=gj[:B])
        outs.append(gj)
    return outs


class MelGanDiscLossCatFn(Function):
    """The same loss on judgements of ONE discriminator pass over [fake; real] (batch 2B): fake =
    j[:B], real = j[B:].  The gradient is written into the two halves of one tensor -- slicing the
    judgements with autograd instead costs a zero fill and a copy per slice in backward."""

    @staticmethod
    def forward(ctx, n, B, *js):
        ctx.cfg = (n, B)
        ctx.js = js
        return disc_loss_cat_fwd(js, B)

    @staticmethod
    def backward(ctx, g):
        n, B = ctx.cfg
        need = [ctx.needs_input_grad[2 + s] for s in range(n)]
        return (None, None) + tuple(disc_loss_cat_bwd(ctx.js, B, _c(g), need))


def gen_loss_fwd(S, Lyr, weight, rf, ff, fj, root_grads=None):
    """sum_s mean(-fj_s) + weight * sum_{s,l} (1/S)(1/Lyr) l1(rf, ff) -> (0-d device tensor, fscale).

    root_grads: a list; when given and the loss is the ROOT of the backward pass (upstream gradient 1, the hand-scheduled
    step), the feature-matching gradients w.r.t. ff are produced in the same pass over the maps and appended to it
    (gen_loss_bwd is then called with need_f all False)."""
    nf = S * Lyr
    dev = fj[0].device
    fscale = float(weight) * (1.0 / S) * (1.0 / Lyr)
    if nf <= L.L1_MULTI_MAX:     # all feature-matching terms in one launch pair
        if P.judge_multi_ok(fj):  # ... and the S adversarial terms in one launch
            terms = torch.empty((2,), dtype=torch.float32, device=dev)
            P.judge_loss_multi_fwd(L.JUDGE_NEG_MEAN, None, list(fj), terms[0:])
            g_ff = P.l1_mean_multi_fwd_bwd(rf, ff, [1.0] * nf, terms[1:], fscale) if root_grads is not None else None
            if g_ff is not None:
                root_grads.extend(g_ff)
            else:
                P.l1_mean_multi_fwd(rf, ff, [1.0] * nf, terms[1:])
            return P.weighted_sum(terms, _coef([1.0, fscale], dev)), fscale
        terms = torch.empty((S + 1,), dtype=torch.float32, device=dev)
        for s in range(S):
            P.neg_mean_fwd(fj[s], terms[s])
        P.l1_mean_multi_fwd(rf, ff, [1.0] * nf, terms[S:])
        return P.weighted_sum(terms, _coef([1.0] * S + [fscale], dev)), fscale
    terms = torch.empty((S + nf,), dtype=torch.float32, device=dev)
    for s in range(S):
        P.neg_mean_fwd(fj[s], terms[s])
    for i in range(nf):
        P.l1_mean_fwd(rf[i], ff[i], terms[S + i])
    return P.weighted_sum(terms, _coef([1.0] * S + [fscale] * nf, dev)), fscale


def gen_loss_bwd(S, fscale, rf, ff, fj, g, need_r, need_f, need_j):
    """-> (grads w.r.t. rf, ff, fj); entries are None where need_* is False."""
    nf = len(rf)
    if nf <= L.L1_MULTI_MAX:
        g_rf = P.l1_mean_multi_bwd(ff, rf, [1.0] * nf, g, fscale, list(need_r)) if any(need_r) else [None] * nf
        g_ff = P.l1_mean_multi_bwd(rf, ff, [1.0] * nf, g, fscale, list(need_f)) if any(need_f) else [None] * nf
    else:
        g_rf = [P.l1_mean_bwd(ff[i], rf[i], g, fscale) if need_r[i] else None for i in range(nf)]
        g_ff = [P.l1_mean_bwd(rf[i], ff[i], g, fscale) if need_f[i] else None for i in range(nf)]
    if P.judge_multi_ok(fj) and all(need_j):
        g_fj = [torch.empty_like(t) for t in fj]
        P.judge_loss_multi_bwd(L.JUDGE_NEG_MEAN, None, list(fj), g, 1.0, None, g_fj)
    else:
        g_fj = [P.neg_mean_bwd(fj[s], g) if need_j[s] else None for s in range(S)]
    return g_rf, g_ff, g_fj


class MelGanGenLossFn(Function):
    """sum_s mean(-f_j) + weight * sum_{s,l} (1/S)(1/L) l1(r_f, f_f)   (loss/loss.py:28-78)

    inputs: S, Lyr, weight, then S*Lyr real features, S*Lyr fake features, S fake judgements."""

    @staticmethod
    def forward(ctx, S, Lyr, weight, *ts):
        nf = S * Lyr
        rf, ff, fj = ts[:nf], ts[nf:2 * nf], ts[2 * nf:2 * nf + S]
        loss, fscale = gen_loss_fwd(S, Lyr, weight, rf, ff, fj)
        ctx.cfg = (S, nf, fscale)
        ctx.ts = ts
        return loss

    @staticmethod
    def backward(ctx, g):
        S, nf, fscale = ctx.cfg
        ts = ctx.ts
        rf, ff, fj = ts[:nf], ts[nf:2 * nf], ts[2 * nf:2 * nf + S]
        need = ctx.needs_input_grad[3:]
        g_rf, g_ff, g_fj = gen_loss_bwd(S, fscale, rf, ff, fj, _c(g), need[:nf], need[nf:2 * nf],
                                        [need[2 * nf + s] for s in range(S)])
        return (None, None, None) + tuple(g_rf) + tuple(g_ff) + tuple(g_fj)


# --------------------------------------------------------------- RealMelGan building blocks
# (reference featuresynth/experiment/realmelgan.py: weight-normed convs, pre-activation
#  ResnetBlock, reflection padding, AvgPool1d(4, 2, 1, count_include_pad=False))

class WeightNormFn(Function):
    """w = g * v / ||v||  (torch.nn.utils.weight_norm with dim=0, realmelgan.py:24-29)."""

    @staticmethod
    def forward(ctx, v, g):
        ctx.save_for_backward(v, g)
        return P.weight_norm_fwd(v, g)

    @staticmethod
    def backward(ctx, gw):
        v, g = ctx.saved_tensors
        gv, gg = P.weight_norm_bwd(v, g, _c(gw))
        return gv, gg


class WeightNormMultiFn(Function):
    """(v0, g0, v1, g1, ...) -> (w0, w1, ...): all weight-normed layers of a network at once."""

    @staticmethod
    def forward(ctx, *vg):
        vs, gs = list(vg[0::2]), list(vg[1::2])
        ctx.save_for_backward(*vg)
        # parameters bound to a FlatAdam bucket: gradients accumulate straight into their slots
        # (see _sink_for) instead of going through one AccumulateGrad add kernel per tensor
        slots = [_bound_slot(t) for t in vg]
        ctx.slots = slots if all(s is not None for s in slots) else None
        return tuple(P.weight_norm_multi_fwd(vs, gs))

    @staticmethod
    def backward(ctx, *gws):
        vg = ctx.saved_tensors
        vs, gs = list(vg[0::2]), list(vg[1::2])
        live = [i for i, gw in enumerate(gws) if gw is not None]
        out = [None] * len(vg)
        if not live:
            return tuple(out)
        direct = ctx.slots is not None and all(ctx.needs_input_grad)
        into = ([ctx.slots[2 * i] for i in live], [ctx.slots[2 * i + 1] for i in live]) if direct else None
        gvs, ggs = P.weight_norm_multi_bwd([vs[i] for i in live], [gs[i] for i in live],
                                           [_c(gws[i]) for i in live], into=into)
        if not direct:
            for k, i in enumerate(live):
                out[2 * i], out[2 * i + 1] = gvs[k], ggs[k]
        return tuple(out)


class Conv1dExFn(Function):
    """conv1d with an optional activation IN FRONT (in_act, applied on load), reflection padding,
    a fused activation behind and a fused residual add:  y = residual + act(conv(in_act(x)))."""

    @staticmethod
    def forward(ctx, x, w, b, residual, stride, pad, dil, groups, pad_mode, act, in_act):
        d, lo = P.conv_desc(x.shape, w.shape, stride, pad, dil, groups, pad_mode, act, in_act)
        y, y_act = P.conv1d_fwd(x, w, b, d, lo, residual=residual,
                                want_y_act=(residual is not None and act != L.ACT_NONE))
        ctx.d = d
        ctx.in_act = in_act
        ctx.has = (b is not None, residual is not None)
        ctx.save_for_backward(x, w, y_act)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y_act = ctx.saved_tensors
        gy = _c(gy)
        d = ctx.d
        ya = y_act if d.act != L.ACT_NONE else None
        gx = gw = gb = None
        has_b, has_r = ctx.has
        if ctx.needs_input_grad[1] or (has_b and ctx.needs_input_grad[2]):
            gw, gb = P.conv1d_bwd_weight(x, gy, ya, d, w.shape, want_bias=has_b)
        if ctx.needs_input_grad[0]:
            gx = P.conv1d_bwd_data(gy, ya, w, d)            # d loss / d in_act(x)
            if ctx.in_act != L.ACT_NONE:
                gx = P.act_bwd(x, gx, ctx.in_act)            # through the activation in front
        gr = gy if (has_r and ctx.needs_input_grad[3]) else None
        return gx, gw, (gb if has_b else None), gr, None, None, None, None, None, None, None


class ConvTranspose1dExFn(Function):
    """y = act(conv_transpose1d(in_act(x)))."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, act, in_act):
        d, lo = P.convt_desc(x.shape, w.shape, stride, pad, act, in_act)
        y = P.convt1d_fwd(x, w, b, d, lo)
        ctx.d = d
        ctx.in_act = in_act
        ctx.has_bias = b is not None
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        gy = _c(gy)
        d = ctx.d
        ya = y if d.act != L.ACT_NONE else None
        gx = gw = gb = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw, gb = P.convt1d_bwd_weight(x, gy, ya, d, w.shape)
        if ctx.needs_input_grad[0]:
            gx = P.convt1d_bwd_data(gy, ya, w, d)
            if ctx.in_act != L.ACT_NONE:
                gx = P.act_bwd(x, gx, ctx.in_act)
        return gx, gw, (gb if ctx.has_bias else None), None, None, None, None


class AvgPoolKFn(Function):
    """F.avg_pool1d(x, k)."""

    @staticmethod
    def forward(ctx, x, k):
        ctx.cfg = (tuple(x.shape), k)
        return P.avg_poolk_fwd(x, k)

    @staticmethod
    def backward(ctx, gy):
        shape, k = ctx.cfg
        return P.avg_poolk_bwd(_c(gy), shape, k), None


class AvgPool421Fn(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        return P.avg_pool421_fwd(x)

    @staticmethod
    def backward(ctx, gy):
        return P.avg_pool421_bwd(_c(gy), ctx.shape)
