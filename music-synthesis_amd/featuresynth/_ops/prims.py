"""Tensor-level wrappers of the C-ABI kernels: shape checks, output allocation, launch on the
current stream.  No autograd here (see functional.py)."""
import os

import numpy as np
import ctypes

import torch

from . import lib as L
from .. import _workload as W


_WHICH = {"fwd": 0, "bwd_data": 1, "bwd_weight": 2}


def _ccost(d, which, **kw):
    return lambda: dict(W.conv_cost(d.B, d.Cin, d.Lin, d.Cout, d.K, d.stride, d.pad, d.dil, d.groups,
                                    which, **kw),
                        geom=(d.B, d.Cin, d.Lin, d.Cout, d.K, d.stride, d.dil, d.groups),
                        kernel=L.load().ms_conv1d_kernel_name(d, _WHICH[which]).decode())


def _tcost(d, which, abi_which, **kw):
    m = W.convt_as_conv(d.B, d.Cin, d.Lin, d.Cout, d.K, d.stride, d.pad)
    return lambda: dict(W.conv_cost(which=which, **m, **kw),
                        geom=(d.B, d.Cin, d.Lin, d.Cout, d.K, d.stride, 1, 1),
                        kernel=L.load().ms_convt1d_kernel_name(d, abi_which).decode())


def _scost(n, reads, writes, flops_per=1):
    return lambda: {"flops": flops_per * n, "bytes": 4 * n * (reads + writes)}

SLOPE = 0.2


def conv_desc(x_shape, w_shape, stride=1, pad=0, dil=1, groups=1, pad_mode=L.PAD_ZERO,
              act=L.ACT_NONE, in_act=L.ACT_NONE):
    B, Cin, Lin = x_shape
    Cout, Cg, K = w_shape
    if Cg * groups != Cin:
        raise RuntimeError("conv1d: weight %s does not match input channels %d / groups %d" %
                           (tuple(w_shape), Cin, groups))
    d = L.ConvDesc(B, Cin, Lin, Cout, K, stride, pad, dil, groups, pad_mode, act, SLOPE, in_act)
    lout = L.load().ms_conv1d_out_len(d)
    if lout <= 0:
        raise RuntimeError("conv1d: invalid geometry x=%s w=%s stride=%d pad=%d dil=%d" %
                           (tuple(x_shape), tuple(w_shape), stride, pad, dil))
    return d, lout


def _out(out, shape, device, what):
    """Caller-provided output (e.g. one half of a batch-concatenated buffer) or a fresh tensor."""
    if out is None:
        return torch.empty(shape, dtype=torch.float32, device=device)
    L.require(out, what)
    if tuple(out.shape) != tuple(shape):
        raise RuntimeError("%s: output buffer %s != %s" % (what, tuple(out.shape), tuple(shape)))
    return out


def conv1d_fwd(x, w, b, d, lout, residual=None, want_y_act=False, out=None):
    L.require(x, "conv1d input"); L.require(w, "conv1d weight")
    if b is not None:
        L.require(b, "conv1d bias")
    y = _out(out, (d.B, d.Cout, lout), x.device, "conv1d output")
    y_act = None
    if residual is not None:
        L.require(residual, "conv1d residual")
        if residual.shape != y.shape:
            raise RuntimeError("conv1d: residual shape %s != output shape %s" %
                               (tuple(residual.shape), tuple(y.shape)))
        if want_y_act:
            y_act = torch.empty_like(y)
    lib = L.load()
    nws = lib.ms_conv1d_workspace_bytes(d, 0)
    ws = L.workspace(nws, x.device)
    L.call("ms_conv1d_fwd", _ccost(d, "fwd", extra_reads=int(residual is not None),
                                   extra_writes=int(y_act is not None)),
           d, x.data_ptr(), w.data_ptr(), L.ptr(b), L.ptr(residual), y.data_ptr(), L.ptr(y_act),
           L.ptr(ws), nws, L.stream())
    return y, (y_act if y_act is not None else y)


def conv1d_bwd_data(gy, y_act, w, d, gx_add=None):
    L.require(gy, "conv1d grad_output")
    gx = torch.empty((d.B, d.Cin, d.Lin), dtype=torch.float32, device=gy.device)
    if gx_add is not None:
        L.require(gx_add, "conv1d gx_add")
        assert gx_add.shape == gx.shape
    lib = L.load()
    nws = lib.ms_conv1d_workspace_bytes(d, 1)
    ws = L.workspace(nws, gy.device)
    L.call("ms_conv1d_bwd_data", _ccost(d, "bwd_data", act_read=y_act is not None,
                                        extra_reads=int(gx_add is not None)),
           d, gy.data_ptr(), L.ptr(y_act), w.data_ptr(), L.ptr(gx_add), gx.data_ptr(), L.ptr(ws), nws,
           L.stream())
    return gx


def conv1d_bwd_weight(x, gy, y_act, d, w_shape, gw=None, gb=None, accumulate=False, want_bias=True):
    L.require(gy, "conv1d grad_output")
    if gw is None:
        gw = torch.empty(tuple(w_shape), dtype=torch.float32, device=gy.device)
        accumulate = False
    if gb is None and want_bias:
        gb = (torch.zeros if accumulate else torch.empty)((d.Cout,), dtype=torch.float32, device=gy.device)
    lib = L.load()
    nws = lib.ms_conv1d_workspace_bytes(d, 2)
    ws = L.workspace(nws, gy.device)
    L.call("ms_conv1d_bwd_weight", _ccost(d, "bwd_weight", act_read=y_act is not None),
           d, x.data_ptr(), gy.data_ptr(), L.ptr(y_act), gw.data_ptr(), L.ptr(gb),
           1.0 if accumulate else 0.0, L.ptr(ws), nws, L.stream())
    return gw, gb


def conv1d_bwd_weight_multi(jobs):
    """jobs: list of (x, gy, y_act, desc, w_shape, gw, gb, accumulate) -- as many conv1d_bwd_weight calls,
    issued through ONE C-ABI entry (one launch pair when the geometries agree: the six k3 convs of a
    ResidualStack).  Returns [(gw, gb)] in job order."""
    out = []
    lib = L.load()
    for lo in range(0, len(jobs), L.WGRAD_MULTI_MAX):
        chunk = jobs[lo:lo + L.WGRAD_MULTI_MAX]
        if len(chunk) == 1:
            x, gy, ya, d, ws_, gw, gb, acc = chunk[0][:8]
            if is_signs(ya):
                raise RuntimeError("conv1d_bwd_weight_multi: sign words only travel through the batched launch")
            out.append(conv1d_bwd_weight(x, gy, ya, d, ws_, gw, gb, acc))
            continue
        md = L.WgradMultiDesc()
        md.count = len(chunk)
        costs = []
        keep = []
        for k, job in enumerate(chunk):
            (x, gy, ya, d, w_shape, gw, gb, acc), bounds = job[:8], job[8:]
            L.require(gy, "conv1d grad_output"); L.require(x, "conv1d input")
            if len(bounds) == 2 and bounds[0] is not None and bounds[1] is not None:    # bounds of |x| and |gy| (fused atoms)
                md.xmax[k], md.gmax[k] = bounds[0].data_ptr(), bounds[1].data_ptr()
                keep += list(bounds)
            if gw is None:
                gw = torch.empty(tuple(w_shape), dtype=torch.float32, device=gy.device)
                acc = False
            if gb is None:
                gb = (torch.zeros if acc else torch.empty)((d.Cout,), dtype=torch.float32, device=gy.device)
            md.conv[k] = d
            md.x[k], md.gy[k] = x.data_ptr(), gy.data_ptr()
            if is_signs(ya):
                md.y_signs[k] = ya.data_ptr()
            else:
                md.y_act[k] = L.ptr(ya)
            md.gw[k], md.gb[k], md.beta[k] = gw.data_ptr(), gb.data_ptr(), 1.0 if acc else 0.0
            costs.append(_ccost(d, "bwd_weight", act_read=ya is not None))
            out.append((gw, gb))
        nws = lib.ms_conv1d_bwd_weight_multi_workspace_bytes(md)
        ws = L.workspace(nws, chunk[0][1].device)

        def cost(costs=costs):
            cs = [c() for c in costs]
            tot = dict(cs[0])
            tot["flops"] = sum(c["flops"] for c in cs)
            tot["bytes"] = sum(c["bytes"] for c in cs)
            tot["batched"] = len(cs)
            return tot
        L.call("ms_conv1d_bwd_weight_multi", cost, md, L.ptr(ws), nws, L.stream())
    return out


# ---- one layer, shared weights, several inputs (the discriminator's scales): ms_conv1d_parts_*
def _part_desc(d, B, Lin):
    return L.ConvDesc(B, d.Cin, Lin, d.Cout, d.K, d.stride, d.pad, d.dil, d.groups, d.pad_mode, d.act, d.slope, d.in_act)


def _parts_cost(d, shapes, which, name, **kw):
    """shapes: [(B, Lin)] of the parts."""
    def cost():
        cs = [W.conv_cost(B, d.Cin, Lin, d.Cout, d.K, d.stride, d.pad, d.dil, d.groups, which, **kw) for B, Lin in shapes]
        nw = 4 * (d.Cout * (d.Cin // d.groups) * d.K + d.Cout)
        return {"flops": sum(c["flops"] for c in cs), "bytes": sum(c["bytes"] for c in cs) - (len(cs) - 1) * nw,   # weights once
                "geom": ("parts", tuple(shapes), d.Cin, d.Cout, d.K, d.stride, d.dil, d.groups),
                "kernel": L.load().ms_conv1d_kernel_name(_part_desc(d, *shapes[0]), _WHICH[which]).decode(), "parts": len(cs)}
    return cost


def conv1d_parts_fwd(xs, w, b, d, image=None):
    """The layer d (its B / Lin are ignored) on every tensor of xs with the same weights -> [y_i].  One launch where a
    parts kernel takes the geometry (grouped k41 layers; the k5 layer on its weight image), else one call per part."""
    if not 1 <= len(xs) <= L.CONV_PARTS_MAX:
        raise RuntimeError("conv1d_parts: 1 .. %d parts" % L.CONV_PARTS_MAX)
    L.require(w, "conv1d weight")
    if b is not None:
        L.require(b, "conv1d bias")
    parts = L.ConvParts()
    parts.count = len(xs)
    ys, shapes = [], []
    for i, x in enumerate(xs):
        L.require(x, "conv1d input")
        if x.dim() != 3 or x.shape[1] != d.Cin:
            raise RuntimeError("conv1d_parts: input %d is %s, expected (B, %d, L)" % (i, tuple(x.shape), d.Cin))
        B, Lin = int(x.shape[0]), int(x.shape[2])
        lo = L.load().ms_conv1d_out_len(_part_desc(d, B, Lin))
        if lo <= 0:
            raise RuntimeError("conv1d_parts: invalid geometry for part %d: %s" % (i, tuple(x.shape)))
        y = torch.empty((B, d.Cout, lo), dtype=torch.float32, device=x.device)
        parts.B[i], parts.Lin[i], parts.x[i], parts.y[i] = B, Lin, x.data_ptr(), y.data_ptr()
        ys.append(y); shapes.append((B, Lin))
    lib = L.load()
    nws = lib.ms_conv1d_parts_workspace_bytes(d, parts, 0, 1 if image is not None else 0)
    ws = L.workspace(nws, xs[0].device)
    L.call("ms_conv1d_parts_fwd", _parts_cost(d, shapes, "fwd", "fwd"), d, parts, w.data_ptr(), L.ptr(b), L.ptr(image),
           L.ptr(ws), nws, L.stream())
    return ys


def conv1d_parts_bwd_data(gys, y_acts, w, d, in_shapes, gx_adds=None, image_bwd=None):
    """Input gradients of the layer for every part: gys[i] (B_i, Cout, Lout_i) -> (B_i, Cin, in_shapes[i][-1]).  y_acts[i]: the
    saved outputs (may hold MORE batch rows than gys[i]: the leading ones are used).  gx_adds[i]: optional addends."""
    n = len(gys)
    parts = L.ConvParts()
    parts.count = n
    gxs, shapes = [], []
    for i in range(n):
        gy = L.require(gys[i], "conv1d grad_output")
        B, Lin = int(gy.shape[0]), int(in_shapes[i][-1])
        ya = y_acts[i] if (y_acts is not None and d.act != L.ACT_NONE) else None
        if ya is not None:
            L.require(ya, "conv1d saved output")
            if ya.shape[0] < B or tuple(ya.shape[1:]) != tuple(gy.shape[1:]):
                raise RuntimeError("conv1d_parts: saved output %s does not cover gradient %s" % (tuple(ya.shape), tuple(gy.shape)))
        ga = gx_adds[i] if gx_adds is not None else None
        gx = torch.empty((B, d.Cin, Lin), dtype=torch.float32, device=gy.device)
        if ga is not None:
            L.require(ga, "conv1d gx_add")
            if ga.shape != gx.shape:
                raise RuntimeError("conv1d_parts: gx_add %s != input gradient %s" % (tuple(ga.shape), tuple(gx.shape)))
        parts.B[i], parts.Lin[i] = B, Lin
        parts.gy[i], parts.y_act[i], parts.gx_add[i], parts.gx[i] = gy.data_ptr(), L.ptr(ya), L.ptr(ga), gx.data_ptr()
        gxs.append(gx); shapes.append((B, Lin))
    lib = L.load()
    nws = lib.ms_conv1d_parts_workspace_bytes(d, parts, 1, 1 if image_bwd is not None else 0)
    ws = L.workspace(nws, gys[0].device)
    L.call("ms_conv1d_parts_bwd_data", _parts_cost(d, shapes, "bwd_data", "bwd_data", act_read=d.act != L.ACT_NONE,
                                                   extra_reads=int(gx_adds is not None and any(g is not None for g in gx_adds))),
           d, parts, w.data_ptr(), L.ptr(image_bwd), L.ptr(ws), nws, L.stream())
    return gxs


def conv1d_parts_bwd_weight(xs, gys, y_acts, d, w_shape, gw=None, gb=None, accumulate=False):
    """gw (+)= sum over the parts of the layer's weight gradient; xs[i] / y_acts[i] may hold more batch rows than gys[i]."""
    n = len(gys)
    dev_ = gys[0].device
    if gw is None:
        gw = torch.empty(tuple(w_shape), dtype=torch.float32, device=dev_)
        accumulate = False
    if gb is None:
        gb = (torch.zeros if accumulate else torch.empty)((d.Cout,), dtype=torch.float32, device=dev_)
    parts = L.ConvParts()
    parts.count = n
    shapes = []
    for i in range(n):
        gy, x = L.require(gys[i], "conv1d grad_output"), L.require(xs[i], "conv1d input")
        B, Lin = int(gy.shape[0]), int(x.shape[2])
        if x.shape[0] < B or x.shape[1] != d.Cin:
            raise RuntimeError("conv1d_parts: input %s does not cover gradient %s" % (tuple(x.shape), tuple(gy.shape)))
        ya = y_acts[i] if (y_acts is not None and d.act != L.ACT_NONE) else None
        if ya is not None and (ya.shape[0] < B or tuple(ya.shape[1:]) != tuple(gy.shape[1:])):
            raise RuntimeError("conv1d_parts: saved output %s does not cover gradient %s" % (tuple(ya.shape), tuple(gy.shape)))
        parts.B[i], parts.Lin[i] = B, Lin
        parts.x[i], parts.gy[i], parts.y_act[i] = x.data_ptr(), gy.data_ptr(), L.ptr(ya)
        shapes.append((B, Lin))
    lib = L.load()
    nws = lib.ms_conv1d_parts_workspace_bytes(d, parts, 2, 0)
    ws = L.workspace(nws, dev_)
    L.call("ms_conv1d_parts_bwd_weight", _parts_cost(d, shapes, "bwd_weight", "bwd_weight", act_read=d.act != L.ACT_NONE),
           d, parts, gw.data_ptr(), gb.data_ptr(), 1.0 if accumulate else 0.0, L.ptr(ws), nws, L.stream())
    return gw, gb


# ---- dense k5 conv on short rows with pre-split weight images (csrc/conv5_img.hip)
def conv_img_bytes(d):
    """Bytes of the weight image the image kernel wants for this conv geometry; 0 = not taken."""
    return int(L.load().ms_conv1d_img_bytes(d))


def conv_img_pack(d, w, backward=False):
    """-> image tensor (uint8) of w for the forward (backward-data) pass of every conv with d's channels / taps."""
    L.require(w, "conv1d weight")
    img = torch.empty(conv_img_bytes(d), dtype=torch.uint8, device=w.device)
    L.call("ms_conv1d_img_pack", _scost(w.numel(), 1, 1.5), d, w.data_ptr(), 1 if backward else 0, img.data_ptr(), L.stream())
    return img


def conv_img_pack2(d, w):
    """-> (forward image, backward-data image) of w: one pass over the weights for their common scale, then the two packs."""
    L.require(w, "conv1d weight")
    img = torch.empty(conv_img_bytes(d), dtype=torch.uint8, device=w.device)
    imgb = torch.empty(conv_img_bytes(d), dtype=torch.uint8, device=w.device)
    L.call("ms_conv1d_img_pack2", _scost(w.numel(), 1, 2.5), d, w.data_ptr(), img.data_ptr(), imgb.data_ptr(), L.stream())
    return img, imgb


def conv1d_img_fwd(x, image, b, d, lout, out=None):
    L.require(x, "conv1d input")
    y = _out(out, (d.B, d.Cout, lout), x.device, "conv1d output")
    nws = L.load().ms_conv1d_img_workspace_bytes(d, 0)
    ws = L.workspace(nws, x.device)

    def cost():
        return dict(W.conv_cost(d.B, d.Cin, d.Lin, d.Cout, d.K, d.stride, d.pad, d.dil, d.groups, "fwd"),
                    geom=(d.B, d.Cin, d.Lin, d.Cout, d.K, d.stride, d.dil, d.groups))
    L.call("ms_conv1d_img_fwd", cost, d, x.data_ptr(), image.data_ptr(), L.ptr(b), y.data_ptr(), L.ptr(ws), nws, L.stream())
    return y


def conv1d_img_bwd_data(gy, y_act, image_bwd, d, gx_add=None):
    L.require(gy, "conv1d grad_output")
    gx = torch.empty((d.B, d.Cin, d.Lin), dtype=torch.float32, device=gy.device)
    nws = L.load().ms_conv1d_img_workspace_bytes(d, 1)
    ws = L.workspace(nws, gy.device)

    def cost():
        return dict(W.conv_cost(d.B, d.Cin, d.Lin, d.Cout, d.K, d.stride, d.pad, d.dil, d.groups, "bwd_data",
                                act_read=y_act is not None, extra_reads=int(gx_add is not None)),
                    geom=(d.B, d.Cin, d.Lin, d.Cout, d.K, d.stride, d.dil, d.groups))
    L.call("ms_conv1d_img_bwd_data", cost, d, gy.data_ptr(), L.ptr(y_act), image_bwd.data_ptr(), L.ptr(gx_add), gx.data_ptr(),
           L.ptr(ws), nws, L.stream())
    return gx


# ---- fused ResidualAtom forward (csrc/atom_fused.hip)
def atom_supported(B, C, Lg, dil):
    return bool(L.load().ms_residual_atom_supported(L.AtomDesc(B, C, Lg, dil, SLOPE)))


def atom_image(C, device):
    """Caller-owned buffer for the pre-split weight image of one atom (both convs)."""
    return torch.empty(int(L.load().ms_residual_atom_image_bytes(C)), dtype=torch.uint8, device=device)


def atom_bwd_supported(B, C, Lg, dil):
    return bool(L.load().ms_residual_atom_bwd_supported(L.AtomDesc(B, C, Lg, dil, SLOPE)))


def atom_pack(jobs, backward=False):
    """jobs: list of (w0, w1, image): splits the fp32 weights of every atom into its image, one launch per 16.
    backward: the images of the backward-data kernel (rows = input channels, taps flipped)."""
    for lo in range(0, len(jobs), L.ATOM_PACK_MAX):
        chunk = jobs[lo:lo + L.ATOM_PACK_MAX]
        d = L.AtomPackDesc()
        d.count = len(chunk)
        n = 0
        for k, (w0, w1, img) in enumerate(chunk):
            d.backward[k] = 1 if backward else 0
            L.require(w0, "atom weight"); L.require(w1, "atom weight")
            C = w0.shape[0]
            if tuple(w0.shape) != (C, C, 3) or tuple(w1.shape) != (C, C, 3):
                raise RuntimeError("residual atom: weights must be (C, C, 3), got %s / %s" % (tuple(w0.shape), tuple(w1.shape)))
            d.C[k], d.w0[k], d.w1[k], d.image[k] = C, w0.data_ptr(), w1.data_ptr(), img.data_ptr()
            n += 2 * w0.numel()
        L.call("ms_residual_atom_pack_multi", _scost(n, 1, 1.58), d, L.stream())


def is_signs(t):
    """Sign words of an activation (csrc/atom_fused.hip, MASK: int16, (B, C / 32, 2, L), bit 15 - r of word (b, blk, h, l)
    = activation[b, 32 blk + (r & 3) + 8 (r >> 2) + 4 h, l] > 0) in place of the fp32 tensor."""
    return t is not None and t.dtype == torch.int16


def stack_signs_ok(x, dils):
    """A training-mode stack of atoms may save SIGN WORDS of u (and of t, beside t itself) instead of the fp32 u: forward,
    backward data and the batched weight gradients all take them at this geometry (MSYNTH_ATOM_SIGNS=0 / MSYNTH_WMULTI=0:
    never)."""
    if os.environ.get("MSYNTH_WMULTI", "1") != "1" or not (1 <= len(dils) <= L.STACK_MAX) or x.dim() != 3:
        return False
    return bool(L.load().ms_residual_stack_signs_supported(ctypes.byref(_stack_desc(x, dils))))


class AtomAux:
    """What a training-mode fused atom forward leaves for the backward pass BESIDE the activations t and u: the sign words of t
    (when u is saved as sign words too) and the launch's operand bounds ([0] of x, [1] of t) for the batched weight gradients.
    Carried in the tape record, never as attributes of tensors (a view / contiguous() of t would silently drop those)."""
    __slots__ = ("t_signs", "amax")

    def __init__(self, t_signs=None, amax=None):
        self.t_signs, self.amax = t_signs, amax


def atom_fwd(x, image, b0, b1, dil, save, signs=False):
    """-> (y, t, u, aux): y = x + lrelu(conv1(lrelu(conv_d(x) + b0)) + b1); t, u (the activations the backward pass needs)
    and aux (AtomAux) only when save, else None.  signs (with save): u is returned as its sign words and aux.t_signs holds
    t's -- what the backward pass reads instead of the two fp32 tensors wherever only the LeakyReLU derivative is needed."""
    L.require(x, "residual atom input"); L.require(b0, "bias"); L.require(b1, "bias")
    B, C, Lg = x.shape
    y = torch.empty_like(x)
    t = torch.empty_like(x) if save else None
    d = L.AtomDesc(B, C, Lg, dil, SLOPE)
    if save and signs:
        su = torch.empty((B, C // 32, 2, Lg), dtype=torch.int16, device=x.device)
        st = torch.empty((B, C // 32, 2, Lg), dtype=torch.int16, device=x.device)
        amax = _amax_buffer(x.device)

        def cost_s():
            c0 = W.conv_cost(B, C, Lg, C, 3, 1, dil, dil, 1, "fwd")
            # moves x (read), y and t (written) and two sign-word tensors of 1 / 32 of an fp32 tensor each
            return {"flops": 2 * c0["flops"], "bytes": int(4 * x.numel() * (3 + 1 / 16)) + 4 * 2 * (3 * C * C + C),
                    "geom": (B, C, Lg, C, 3, 1, dil, 1)}
        L.call("ms_residual_atom_fwd_signs", cost_s, d, x.data_ptr(), image.data_ptr(), b0.data_ptr(), b1.data_ptr(),
               y.data_ptr(), t.data_ptr(), st.data_ptr(), su.data_ptr(), L.ptr(amax), L.stream())
        return y, t, su, AtomAux(st, amax)
    u = torch.empty_like(x) if save else None
    # training: the launch's per-workgroup operand maxima ([0] of x, [1] of t) go to the weight-gradient kernel, which takes its
    # block scales from them (no extra pass over the tensors)
    amax = _amax_buffer(x.device) if save else None

    def cost():
        c0 = W.conv_cost(B, C, Lg, C, 3, 1, dil, dil, 1, "fwd")
        return {"flops": 2 * c0["flops"], "bytes": 4 * x.numel() * (2 + 2 * int(save)) + 4 * 2 * (3 * C * C + C),
                "geom": (B, C, Lg, C, 3, 1, dil, 1)}
    L.call("ms_residual_atom_fwd", cost, d, x.data_ptr(), image.data_ptr(), b0.data_ptr(), b1.data_ptr(),
           y.data_ptr(), L.ptr(t), L.ptr(u), L.ptr(amax), L.stream())
    return y, t, u, (AtomAux(None, amax) if save else None)


def _stack_desc(x, dils):
    B, C, Lg = x.shape
    d = L.StackDesc()
    d.B, d.C, d.L, d.count, d.slope = B, C, Lg, len(dils), SLOPE
    for i, dl in enumerate(dils):
        d.dil[i] = dl
    return d


def stack_supported(x, dils):
    """The whole ResidualStack (len(dils) atoms) in one launch, inference only (csrc/stack_fused.hip)."""
    return 1 <= len(dils) <= L.STACK_MAX and x.dim() == 3 and bool(L.load().ms_residual_stack_supported(ctypes.byref(_stack_desc(x, dils))))


def stack_fwd(x, images, b0s, b1s, dils):
    """-> y = atom[n-1](... atom[0](x)), nothing saved.  images / b0s / b1s: per atom (forward images of atom_pack)."""
    L.require(x, "residual stack input")
    for b in list(b0s) + list(b1s):
        L.require(b, "bias")
    B, C, Lg = x.shape
    n = len(dils)
    y = torch.empty_like(x)
    d = _stack_desc(x, dils)
    arr = ctypes.c_void_p * n
    im = arr(*[t.data_ptr() for t in images])
    pb0 = arr(*[t.data_ptr() for t in b0s])
    pb1 = arr(*[t.data_ptr() for t in b1s])

    def cost():
        fl = sum(2 * W.conv_cost(B, C, Lg, C, 3, 1, dl, dl, 1, "fwd")["flops"] for dl in dils)
        return {"flops": fl, "bytes": 4 * x.numel() * 2 + n * 4 * 2 * (3 * C * C + C), "geom": (B, C, Lg, C, 3, 1, tuple(dils), 1)}
    L.call("ms_residual_stack_fwd", cost, d, x.data_ptr(), ctypes.cast(im, ctypes.c_void_p), ctypes.cast(pb0, ctypes.c_void_p),
           ctypes.cast(pb1, ctypes.c_void_p), y.data_ptr(), L.stream())
    return y


def _amax_buffer(device):
    if not L.load().ms_residual_atom_publishes_amax():
        return None
    return torch.empty((2, L.ATOM_AMAX_N), dtype=torch.float32, device=device)


def atom_bwd_data(g, u, t, image_bwd, dil, t_signs=None):
    """-> (gt, gx, amax): gt = conv1^T(g * lrelu'(u)) (raw), gx = g + conv_d^T(gt * lrelu'(t)) -- the atom's backward data, one
    launch; amax = the launch's operand maxima ([0] of g, [1] of gt lrelu'(t)) for the weight gradients, or None.
    u as sign words (atom_fwd(..., signs=True)) needs t_signs, the forward's AtomAux.t_signs."""
    sg = is_signs(u)
    for a, nm in ((g, "grad_output"), (t, "t")) + (() if sg else ((u, "y_act"),)):
        L.require(a, "residual atom " + nm)
    B, C, Lg = g.shape
    gt, gx = torch.empty_like(g), torch.empty_like(g)
    d = L.AtomDesc(B, C, Lg, dil, SLOPE)
    amax = _amax_buffer(g.device)
    if sg:
        if not is_signs(t_signs):
            raise RuntimeError("residual atom backward: u is sign words, t's sign words (AtomAux.t_signs) are required")

        def cost_s():
            c0 = W.conv_cost(B, C, Lg, C, 3, 1, dil, dil, 1, "bwd_data")
            # moves g (read), gt and gx (written) and the sign words of u and t (1 / 32 of an fp32 tensor each)
            return {"flops": 2 * c0["flops"], "bytes": int(4 * g.numel() * (3 + 1 / 16)) + 4 * 2 * 3 * C * C,
                    "geom": (B, C, Lg, C, 3, 1, dil, 1)}
        L.call("ms_residual_atom_bwd_data_signs", cost_s, d, g.data_ptr(), u.data_ptr(), t_signs.data_ptr(), image_bwd.data_ptr(),
               gt.data_ptr(), gx.data_ptr(), L.ptr(amax), L.stream())
        return gt, gx, amax

    def cost():
        c0 = W.conv_cost(B, C, Lg, C, 3, 1, dil, dil, 1, "bwd_data")
        return {"flops": 2 * c0["flops"], "bytes": 4 * g.numel() * 5 + 4 * 2 * 3 * C * C, "geom": (B, C, Lg, C, 3, 1, dil, 1)}
    L.call("ms_residual_atom_bwd_data", cost, d, g.data_ptr(), u.data_ptr(), t.data_ptr(), image_bwd.data_ptr(),
           gt.data_ptr(), gx.data_ptr(), L.ptr(amax), L.stream())
    return gt, gx, amax


def convt_desc(x_shape, w_shape, stride, pad, act=L.ACT_NONE, in_act=L.ACT_NONE):
    B, Cin, Lin = x_shape
    Cin2, Cout, K = w_shape
    if Cin2 != Cin:
        raise RuntimeError("conv_transpose1d: weight %s does not match input channels %d" %
                           (tuple(w_shape), Cin))
    d = L.ConvTDesc(B, Cin, Lin, Cout, K, stride, pad, act, SLOPE, in_act)
    lout = L.load().ms_convt1d_out_len(d)
    if lout <= 0:
        raise RuntimeError("conv_transpose1d: invalid geometry")
    return d, lout


def convt1d_fwd(x, w, b, d, lout, out=None, img=None):
    """img: the layer's pre-packed forward weight image (convt_img_pack), when the caller packed it ahead."""
    L.require(x, "conv_transpose1d input"); L.require(w, "conv_transpose1d weight")
    if convt_img_bytes(d):
        return convt1d_img_fwd(x, w, b, d, lout, out=out, img=img)
    y = _out(out, (d.B, d.Cout, lout), x.device, "conv_transpose1d output")
    lib = L.load()
    nws = lib.ms_convt1d_workspace_bytes(d, 0)
    ws = L.workspace(nws, x.device)
    L.call("ms_convt1d_fwd", _tcost(d, "bwd_data", 0), d, x.data_ptr(), w.data_ptr(), L.ptr(b),
           y.data_ptr(), L.ptr(ws), nws, L.stream())
    return y


def convt_img_bytes(d):
    """Bytes of the weight image the transposed-conv image kernel wants (csrc/convt_img.hip); 0 = geometry not taken."""
    return int(L.load().ms_convt1d_img_bytes(d))


def convt_img_pack(d, w):
    """-> forward weight image of a transposed conv the image kernels take (csrc/convt_img.hip, convt_fwd_short.hip)."""
    L.require(w, "conv_transpose1d weight")
    img = torch.empty(convt_img_bytes(d), dtype=torch.uint8, device=w.device)
    L.call("ms_convt1d_img_pack", _scost(w.numel(), 1, 0.75), d, w.data_ptr(), img.data_ptr(), L.stream())
    return img


def convt1d_img_fwd(x, w, b, d, lout, out=None, img=None):
    """ConvTranspose1d forward on a pre-split weight image (packed here unless the caller did: one small launch)."""
    L.require(x, "conv_transpose1d input")
    if img is None:
        img = convt_img_pack(d, w)
    y = _out(out, (d.B, d.Cout, lout), x.device, "conv_transpose1d output")
    nws = L.load().ms_convt1d_img_workspace_bytes(d)
    ws = L.workspace(nws, x.device)
    L.call("ms_convt1d_img_fwd", _tcost(d, "bwd_data", 0), d, x.data_ptr(), img.data_ptr(), L.ptr(b), y.data_ptr(), L.ptr(ws), nws,
           L.stream())
    return y


def convt_bwd_img_bytes(d):
    """Bytes of the weight image the transposed-conv backward-data image kernel wants (csrc/convt_bwd_img.hip); 0 = not taken."""
    return int(L.load().ms_convt1d_bwd_img_bytes(d))


def convt_bwd_img_pack(d, w):
    """-> backward-data weight image of a transposed conv (csrc/convt_bwd_img.hip)."""
    L.require(w, "conv_transpose1d weight")
    img = torch.empty(convt_bwd_img_bytes(d), dtype=torch.uint8, device=w.device)
    L.call("ms_convt1d_bwd_img_pack", _scost(w.numel(), 1, 1.5, 0), d, w.data_ptr(), img.data_ptr(), L.stream())
    return img


def convt1d_bwd_data(gy, y_act, w, d, out=None, img=None):
    """img: the layer's pre-packed backward-data image (convt_bwd_img_pack), when the caller packed it ahead."""
    L.require(gy, "conv_transpose1d grad_output")
    gx = _out(out, (d.B, d.Cin, d.Lin), gy.device, "conv_transpose1d grad_input")
    lib = L.load()
    nimg = convt_bwd_img_bytes(d)
    if nimg:
        if img is None:
            img = convt_bwd_img_pack(d, w)
        nws = lib.ms_convt1d_bwd_img_workspace_bytes(d)
        ws = L.workspace(nws, gy.device)
        L.call("ms_convt1d_bwd_img_data", _tcost(d, "fwd", 1, extra_reads=int(y_act is not None)),
               d, gy.data_ptr(), L.ptr(y_act), img.data_ptr(), gx.data_ptr(), L.ptr(ws), nws, L.stream())
        return gx
    nws = lib.ms_convt1d_workspace_bytes(d, 1)
    ws = L.workspace(nws, gy.device)
    L.call("ms_convt1d_bwd_data", _tcost(d, "fwd", 1, extra_reads=int(y_act is not None)),
           d, gy.data_ptr(), L.ptr(y_act), w.data_ptr(), gx.data_ptr(), L.ptr(ws), nws, L.stream())
    return gx


def convt1d_bwd_weight(x, gy, y_act, d, w_shape, gw=None, gb=None, accumulate=False):
    L.require(gy, "conv_transpose1d grad_output")
    if gw is None:
        gw = torch.empty(tuple(w_shape), dtype=torch.float32, device=gy.device)
        accumulate = False
    if gb is None:
        gb = (torch.zeros if accumulate else torch.empty)((d.Cout,), dtype=torch.float32, device=gy.device)
    lib = L.load()
    nws = lib.ms_convt1d_workspace_bytes(d, 2)
    ws = L.workspace(nws, gy.device)
    L.call("ms_convt1d_bwd_weight", _tcost(d, "bwd_weight", 2, act_read=y_act is not None),
           d, x.data_ptr(), gy.data_ptr(), L.ptr(y_act), gw.data_ptr(), gb.data_ptr(),
           1.0 if accumulate else 0.0, L.ptr(ws), nws, L.stream())
    return gw, gb


def pool_out_len(lin):
    return (lin + 4 - 4) // 2 + 1


def avg_pool_fwd(x, out=None):
    L.require(x, "avg_pool1d input")
    B, C, Lin = x.shape
    y = _out(out, (B, C, pool_out_len(Lin)), x.device, "avg_pool1d output")
    L.call("ms_avg_pool1d_4_2_2_fwd", _scost(x.numel(), 1, 0.5), x.data_ptr(), y.data_ptr(), B * C, Lin,
           L.stream())
    return y


def avg_pool_bwd(gy, x_shape, gx_add=None):
    L.require(gy, "avg_pool1d grad_output")
    B, C, Lin = x_shape
    gx = torch.empty((B, C, Lin), dtype=torch.float32, device=gy.device)
    L.call("ms_avg_pool1d_4_2_2_bwd", _scost(gx.numel(), 0.5 + int(gx_add is not None), 1),
           gy.data_ptr(), L.ptr(gx_add), gx.data_ptr(), B * C, Lin, L.stream())
    return gx


def pool421_out_len(lin):
    return (lin + 2 - 4) // 2 + 1


def avg_pool421_fwd(x):
    """nn.AvgPool1d(4, stride=2, padding=1, count_include_pad=False)."""
    L.require(x, "avg_pool1d input")
    B, C, Lin = x.shape
    y = torch.empty((B, C, pool421_out_len(Lin)), dtype=torch.float32, device=x.device)
    L.call("ms_avg_pool1d_4_2_1_fwd", _scost(x.numel(), 1, 0.5), x.data_ptr(), y.data_ptr(), B * C, Lin,
           L.stream())
    return y


def avg_pool421_bwd(gy, x_shape, gx_add=None):
    L.require(gy, "avg_pool1d grad_output")
    B, C, Lin = x_shape
    gx = torch.empty((B, C, Lin), dtype=torch.float32, device=gy.device)
    L.call("ms_avg_pool1d_4_2_1_bwd", _scost(gx.numel(), 0.5 + int(gx_add is not None), 1),
           gy.data_ptr(), L.ptr(gx_add), gx.data_ptr(), B * C, Lin, L.stream())
    return gx


def avg_poolk_fwd(x, k):
    """F.avg_pool1d(x, k)."""
    L.require(x, "avg_pool1d input")
    B, C, Lin = x.shape
    if k < 1 or Lin < k:
        raise RuntimeError("avg_pool1d: window %d does not fit %d samples" % (k, Lin))
    y = torch.empty((B, C, Lin // k), dtype=torch.float32, device=x.device)
    L.call("ms_avg_pool1d_k_fwd", _scost(x.numel(), 1, 1.0 / k), x.data_ptr(), y.data_ptr(), B * C, Lin, k, L.stream())
    return y


def avg_poolk_bwd(gy, x_shape, k):
    L.require(gy, "avg_pool1d grad_output")
    B, C, Lin = x_shape
    gx = torch.empty((B, C, Lin), dtype=torch.float32, device=gy.device)
    L.call("ms_avg_pool1d_k_bwd", _scost(gx.numel(), 1.0 / k, 1), gy.data_ptr(), gx.data_ptr(), B * C, Lin, k, L.stream())
    return gx


def weight_norm_fwd(v, g):
    """w = g * v / ||v|| over all dims but 0 (torch.nn.utils.weight_norm, dim=0)."""
    L.require(v, "weight_v"); L.require(g, "weight_g")
    rows, cols = v.shape[0], v.numel() // v.shape[0]
    w = torch.empty_like(v)
    L.call("ms_weight_norm_fwd", _scost(v.numel(), 2, 1, 3), v.data_ptr(), g.data_ptr(), w.data_ptr(), rows,
           cols, L.stream())
    return w


def weight_norm_bwd(v, g, gw, gv=None, gg=None):
    L.require(gw, "weight grad")
    acc = gv is not None
    if gv is None:
        gv = torch.empty_like(v)
        gg = torch.empty_like(g)
    rows, cols = v.shape[0], v.numel() // v.shape[0]
    L.call("ms_weight_norm_bwd", _scost(v.numel(), 3, 1, 6), v.data_ptr(), g.data_ptr(), gw.data_ptr(),
           gv.data_ptr(), gg.data_ptr(), rows, cols, 1.0 if acc else 0.0, L.stream())
    return gv, gg


def _wn_chunks(n):
    return [(i, min(i + L.WN_MULTI_MAX, n)) for i in range(0, n, L.WN_MULTI_MAX)]


def weight_norm_multi_fwd(vs, gs):
    """Effective weights of every weight-normed layer of a network in one launch per 64 tensors."""
    ws = [torch.empty_like(v) for v in vs]
    for lo, hi in _wn_chunks(len(vs)):
        d = L.WnMultiDesc()
        d.count = hi - lo
        n = 0
        for i in range(lo, hi):
            v, g = vs[i], gs[i]
            L.require(v, "weight_v"); L.require(g, "weight_g")
            k = i - lo
            d.v[k], d.g[k], d.out[k] = v.data_ptr(), g.data_ptr(), ws[i].data_ptr()
            d.rows[k], d.cols[k] = v.shape[0], v.numel() // v.shape[0]
            n += v.numel()
        L.call("ms_weight_norm_multi_fwd", _scost(n, 2, 1, 3), d, L.stream())
    return ws


def weight_norm_multi_bwd(vs, gs, gws, into=None):
    """`into` = (gv tensors, gg tensors) to ACCUMULATE into (flat-bucket slots); else fresh outputs."""
    if into is not None:
        gvs, ggs = into
        for t in list(gvs) + list(ggs):
            L.require(t, "gradient slot")
    else:
        gvs = [torch.empty_like(v) for v in vs]
        ggs = [torch.empty_like(g) for g in gs]
    for lo, hi in _wn_chunks(len(vs)):
        d = L.WnMultiDesc()
        d.count = hi - lo
        n = 0
        for i in range(lo, hi):
            L.require(gws[i], "weight grad")
            k = i - lo
            d.v[k], d.g[k], d.out[k] = vs[i].data_ptr(), gs[i].data_ptr(), gws[i].data_ptr()
            d.gv[k], d.gg[k] = gvs[i].data_ptr(), ggs[i].data_ptr()
            d.rows[k], d.cols[k] = vs[i].shape[0], vs[i].numel() // vs[i].shape[0]
            n += vs[i].numel()
        L.call("ms_weight_norm_multi_bwd", _scost(n, 3, 1, 6), d, 1.0 if into is not None else 0.0, L.stream())
    return gvs, ggs


def act_bwd(y_act, gy, act):
    """gy * act'(.) evaluated from y_act (for LeakyReLU: the sign of y_act)."""
    L.require(y_act, "activation"); L.require(gy, "grad")
    out = torch.empty_like(gy)
    L.call("ms_act_bwd", _scost(gy.numel(), 2, 1), y_act.data_ptr(), gy.data_ptr(), out.data_ptr(),
           gy.numel(), act, SLOPE, L.stream())
    return out


def lines_desc(x_shape, phases):
    """x_shape (B, H, C, W); phases: [[dy, ...], ...] -- the input-row offsets of every output-row phase (csrc/lines.hip)."""
    B, H, C, W = x_shape
    d = L.LinesDesc()
    d.B, d.H, d.C, d.W = B, H, C, W
    d.phases, d.taps = len(phases), len(phases[0])
    if d.phases > L.LINES_MAX_PHASES or d.taps > L.LINES_MAX_TAPS or any(len(t) != d.taps for t in phases):
        raise RuntimeError("lines: %d phases x %s taps not supported" % (len(phases), [len(t) for t in phases]))
    for ph, taps in enumerate(phases):
        for j, dy in enumerate(taps):
            d.dy[ph * L.LINES_MAX_TAPS + j] = dy
    return d


def lines_stack(x, d):
    """(B, H, C, W) lines -> (phases, B*H, taps*C, W): the channel-stacked input rows of every output-row phase, one pass."""
    L.require(x, "lines_stack input")
    out = torch.empty((d.phases, d.B * d.H, d.taps * d.C, d.W), dtype=torch.float32, device=x.device)
    L.call("ms_lines_stack", _scost(x.numel(), 1, d.phases * d.taps, 0), d, x.data_ptr(), out.data_ptr(), L.stream())
    return out


def lines_fold(gstack, d):
    """Transpose of lines_stack: (phases, B*H, taps*C, W) -> (B, H, C, W), overlapping rows summed."""
    L.require(gstack, "lines_fold input")
    gx = torch.empty((d.B, d.H, d.C, d.W), dtype=torch.float32, device=gstack.device)
    L.call("ms_lines_fold", _scost(gx.numel(), d.phases * d.taps, 1, d.phases * d.taps), d, gstack.data_ptr(), gx.data_ptr(),
           L.stream())
    return gx


def lines_interleave(src, rows, phases, n, inverse=False):
    """(phases, rows, n) -> (rows, phases, n), or back with inverse (flat buffers; returns a flat tensor)."""
    L.require(src, "lines_interleave input")
    if src.numel() != rows * phases * n:
        raise RuntimeError("lines_interleave: %d elements, expected %d x %d x %d" % (src.numel(), phases, rows, n))
    dst = torch.empty(src.numel(), dtype=torch.float32, device=src.device)
    L.call("ms_lines_interleave", _scost(src.numel(), 1, 1, 0), src.data_ptr(), dst.data_ptr(), rows, phases, n,
           1 if inverse else 0, L.stream())
    return dst


def add(a, b):
    L.require(a, "add lhs"); L.require(b, "add rhs")
    out = torch.empty_like(a)
    L.call("ms_add", _scost(a.numel(), 2, 1), a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), L.stream())
    return out


def add_act(a, b, act):
    """act(a + b)."""
    L.require(a, "add_act lhs"); L.require(b, "add_act rhs")
    if a.shape != b.shape:
        raise RuntimeError("add_act: shape mismatch %s vs %s" % (tuple(a.shape), tuple(b.shape)))
    out = torch.empty_like(a)
    L.call("ms_add_act", _scost(a.numel(), 2, 1, 2), a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), act,
           SLOPE, L.stream())
    return out


def add_(a, b):
    """a += b (in place)."""
    L.require(a, "add_ lhs"); L.require(b, "add_ rhs")
    if a.shape != b.shape:
        raise RuntimeError("add_: shape mismatch %s vs %s" % (tuple(a.shape), tuple(b.shape)))
    L.call("ms_add", _scost(a.numel(), 2, 1), a.data_ptr(), b.data_ptr(), a.data_ptr(), a.numel(), L.stream())
    return a


def _reduce(fn_name, a, b, out=None):
    lib = L.load()
    n = a.numel()
    if out is None:
        out = torch.empty((), dtype=torch.float32, device=a.device)
    nws = lib.ms_reduce_workspace_bytes(n)
    ws = L.workspace(nws, a.device)
    if b is None:
        L.call(fn_name, _scost(n, 1, 0, 2), a.data_ptr(), n, out.data_ptr(), L.ptr(ws), nws, L.stream())
    else:
        L.call(fn_name, _scost(n, 2, 0, 3), a.data_ptr(), b.data_ptr(), n, out.data_ptr(), L.ptr(ws), nws,
               L.stream())
    return out


def hinge_d_fwd(r, f, out=None):
    return _reduce("ms_hinge_d_fwd", L.require(r, "real judgement"), L.require(f, "fake judgement"), out)


def neg_mean_fwd(f, out=None):
    return _reduce("ms_neg_mean_fwd", L.require(f, "judgement"), None, out)


def l1_mean_fwd(r, f, out=None):
    return _reduce("ms_l1_mean_fwd", L.require(r, "real feature"), L.require(f, "fake feature"), out)


def ls_g_fwd(j, out=None):
    return _reduce("ms_ls_g_fwd", L.require(j, "judgement"), None, out)


def ls_d_fwd(r, f, out=None):
    return _reduce("ms_ls_d_fwd", L.require(r, "real judgement"), L.require(f, "fake judgement"), out)


def hinge_d_bwd(r, f, gout, scale=1.0, want_r=True, want_f=True, gr=None, gf=None):
    """gr / gf may be preallocated (e.g. the two halves of one [fake; real] gradient tensor)."""
    if gr is None:
        gr = torch.empty_like(r) if want_r else None
    if gf is None:
        gf = torch.empty_like(f) if want_f else None
    L.call("ms_hinge_d_bwd", _scost(r.numel(), 2, 2), r.data_ptr(), f.data_ptr(), r.numel(),
           gout.data_ptr(), scale, L.ptr(gr), L.ptr(gf), L.stream())
    return gr, gf


def neg_mean_bwd(f, gout, scale=1.0):
    gf = torch.empty_like(f)
    L.call("ms_neg_mean_bwd", _scost(f.numel(), 0, 1), f.numel(), gout.data_ptr(), scale, gf.data_ptr(),
           L.stream())
    return gf


def l1_mean_bwd(r, f, gout, scale=1.0, gf=None):
    acc = gf is not None
    if gf is None:
        gf = torch.empty_like(f)
    L.call("ms_l1_mean_bwd", _scost(f.numel(), 2 + int(acc), 1, 2), r.data_ptr(), f.data_ptr(), f.numel(),
           gout.data_ptr(), scale, gf.data_ptr(), 1 if acc else 0, L.stream())
    return gf


def _l1_multi_desc(rs, fs, ws, gfs=None):
    if len(rs) > L.L1_MULTI_MAX or len(rs) != len(fs) or len(rs) != len(ws):
        raise RuntimeError("l1_mean_multi: at most %d tensor pairs" % L.L1_MULTI_MAX)
    d = L.L1MultiDesc()
    d.count = len(rs)
    for i, (r, f) in enumerate(zip(rs, fs)):
        L.require(r, "real feature"); L.require(f, "fake feature")
        if r.shape != f.shape:
            raise RuntimeError("l1_mean_multi: shape mismatch %s vs %s" % (tuple(r.shape), tuple(f.shape)))
        d.r[i], d.f[i], d.n[i], d.w[i] = r.data_ptr(), f.data_ptr(), r.numel(), float(ws[i])
        d.gf[i] = gfs[i].data_ptr() if (gfs is not None and gfs[i] is not None) else None
    return d


def l1_mean_multi_fwd(rs, fs, ws, out):
    """out[0] = sum_i ws[i] * mean(|fs[i] - rs[i]|) in one launch pair (the 18 feature-matching terms)."""
    d = _l1_multi_desc(rs, fs, ws)
    lib = L.load()
    nws = lib.ms_l1_mean_multi_workspace_bytes(d)
    wsb = L.workspace(nws, out.device)
    n = sum(int(r.numel()) for r in rs)
    L.call("ms_l1_mean_multi_fwd", _scost(n, 2, 0, 2), d, out.data_ptr(), L.ptr(wsb), nws, L.stream())
    return out


def l1_mean_multi_bwd(rs, fs, ws, gout, scale, need):
    """Gradients w.r.t. fs[i] for the i with need[i] (others None), one launch."""
    gfs = [torch.empty_like(f) if nd else None for f, nd in zip(fs, need)]
    d = _l1_multi_desc(rs, fs, ws, gfs)
    n = sum(int(f.numel()) for f, nd in zip(fs, need) if nd)
    L.call("ms_l1_mean_multi_bwd", _scost(n, 2, 1, 2), d, gout.data_ptr(), float(scale), L.stream())
    return gfs


def l1_mean_multi_fwd_bwd(rs, fs, ws, out, gconst):
    """out[0] = sum_i ws[i] * mean(|fs[i] - rs[i]|) AND the gradients w.r.t. every fs[i] for an upstream gradient known on
    the host (gconst), in one pass over the maps; -> list of gradients, or None when a map is not 16-byte shaped (the
    caller then uses the two separate calls)."""
    gfs = [torch.empty_like(f) for f in fs]
    d = _l1_multi_desc(rs, fs, ws, gfs)
    lib = L.load()
    nws = lib.ms_l1_mean_multi_fwd_bwd_workspace_bytes(d)
    if nws == 0:
        return None
    wsb = L.workspace(nws, out.device)
    n = sum(int(r.numel()) for r in rs)
    L.call("ms_l1_mean_multi_fwd_bwd", _scost(n, 2, 1, 2), d, out.data_ptr(), float(gconst), L.ptr(wsb), nws, L.stream())
    return gfs


def ls_g_bwd(j, gout, scale=1.0):
    gj = torch.empty_like(j)
    L.call("ms_ls_g_bwd", _scost(j.numel(), 1, 1), j.data_ptr(), j.numel(), gout.data_ptr(), scale,
           gj.data_ptr(), L.stream())
    return gj


def ls_d_bwd(r, f, gout, scale=1.0):
    gr, gf = torch.empty_like(r), torch.empty_like(f)
    L.call("ms_ls_d_bwd", _scost(r.numel(), 2, 2), r.data_ptr(), f.data_ptr(), r.numel(),
           gout.data_ptr(), scale, gr.data_ptr(), gf.data_ptr(), L.stream())
    return gr, gf


def judge_multi_ok(ts):
    return 0 < len(ts) <= L.JUDGE_MULTI_MAX and all(t is not None and t.numel() <= L.JUDGE_MULTI_NMAX for t in ts)


def _judge_desc(kind, rs, fs, grs=None, gfs=None):
    d = L.JudgeMultiDesc()
    d.count, d.kind = len(fs), kind
    for i, f in enumerate(fs):
        L.require(f, "judgement")
        d.f[i], d.n[i] = f.data_ptr(), f.numel()
        if rs is not None:
            L.require(rs[i], "real judgement")
            if rs[i].numel() != f.numel():
                raise RuntimeError("judge loss: real / fake judgement sizes differ")
            d.r[i] = rs[i].data_ptr()
        d.gr[i] = grs[i].data_ptr() if (grs is not None and grs[i] is not None) else None
        d.gf[i] = gfs[i].data_ptr() if (gfs is not None and gfs[i] is not None) else None
    return d


def judge_loss_multi_fwd(kind, rs, fs, out):
    """out[0] = sum over the scales of the hinge-D / negative-mean term, one launch."""
    d = _judge_desc(kind, rs, fs)
    n = sum(int(f.numel()) for f in fs)
    L.call("ms_judge_loss_multi_fwd", _scost(n, 2 if rs is not None else 1, 0, 3), d, out.data_ptr(), L.stream())
    return out


def judge_loss_multi_bwd(kind, rs, fs, gout, scale, grs, gfs):
    d = _judge_desc(kind, rs, fs, grs, gfs)
    n = sum(int(f.numel()) for f in fs)
    L.call("ms_judge_loss_multi_bwd", _scost(n, 2 if rs is not None else 0, 2), d, gout.data_ptr(), float(scale),
           L.stream())


def weighted_sum(terms, coef):
    """terms, coef: 1-D fp32 device tensors of equal length -> 0-d tensor sum(coef*terms)."""
    out = torch.empty((), dtype=torch.float32, device=terms.device)
    L.call("ms_weighted_sum", None, terms.data_ptr(), coef.data_ptr(), terms.numel(), out.data_ptr(),
           L.stream())
    return out


def adam_step(p, g, m, v, step, lr, b1, b2, eps, grad_scale=1.0):
    for t, nm in ((p, "param"), (g, "grad"), (m, "exp_avg"), (v, "exp_avg_sq")):
        L.require(t, "adam " + nm)
    if step.dtype != torch.int32 or not step.is_cuda:
        raise RuntimeError("adam: step must be an int32 device tensor")
    L.call("ms_adam_step", _scost(p.numel(), 4, 3, 12), p.data_ptr(), g.data_ptr(), m.data_ptr(),
           v.data_ptr(), p.numel(), lr, b1, b2, eps, grad_scale, step.data_ptr(), L.stream())


def audio2mel(audio, window, basis, n_fft, hop):
    L.require(audio, "audio"); L.require(window, "window"); L.require(basis, "mel_basis")
    B, N = audio.shape
    n_mel = basis.shape[0]
    lib = L.load()
    frames = lib.ms_audio2mel_frames(N, n_fft, hop)
    if frames <= 0:
        raise RuntimeError("Audio2Mel: %d samples are too few for one %d-sample frame" % (N, n_fft))
    out = torch.empty((B, n_mel, frames), dtype=torch.float32, device=audio.device)
    L.call("ms_audio2mel_fwd", None, audio.data_ptr(), B, N, window.data_ptr(), n_fft, hop,
           basis.data_ptr(), n_mel, out.data_ptr(), L.stream())
    return out


def resample_sinc(x, ratio, interp_win, interp_delta, num_table):
    """x (rows, n_in) -> (rows, ceil(n_in * ratio)): band-limited sinc interpolation with the given half window."""
    L.require(x, "audio"); L.require(interp_win, "interp_win"); L.require(interp_delta, "interp_delta")
    rows, n_in = x.shape
    n_out = int(np.ceil(n_in * ratio))
    y = torch.empty((rows, n_out), dtype=torch.float32, device=x.device)
    L.call("ms_resample_sinc_fwd", None, x.data_ptr(), rows, n_in, y.data_ptr(), n_out, float(ratio),
           interp_win.data_ptr(), interp_delta.data_ptr(), interp_win.numel(), int(num_table), L.stream())
    return y


def peak_normalize_(x, scale):
    """rows of x scaled in place to max|row| = scale."""
    L.require(x, "audio")
    rows, n = x.shape
    ws = torch.empty((rows,), dtype=torch.float32, device=x.device)
    L.call("ms_peak_normalize", None, x.data_ptr(), rows, n, float(scale), ws.data_ptr(), L.stream())
    return x
