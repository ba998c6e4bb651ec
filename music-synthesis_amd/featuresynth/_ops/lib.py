"""ctypes binding of libmsynth_hip.so (the C ABI declared in include/msynth.h).

There is no CPU fallback: if the shared library is missing, or an op is handed a
tensor that is not a contiguous fp32 HIP-device tensor, a RuntimeError is raised.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MSYNTH_LIB: another build of the same library -- kernel A/B experiments, tools/scratch; never set in normal operation)
LIB_PATH = os.environ.get("MSYNTH_LIB") or os.path.join(os.path.dirname(_HERE), "_lib", "libmsynth_hip.so")

ACT_NONE, ACT_LRELU, ACT_TANH = 0, 1, 2
PAD_ZERO, PAD_REFLECT = 0, 1

_c_int = ctypes.c_int32
_c_i64 = ctypes.c_int64
_c_f = ctypes.c_float
_vp = ctypes.c_void_p
_sz = ctypes.c_size_t


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, _c_int) for n in ("B", "Cin", "Lin", "Cout", "K", "stride", "pad", "dil",
                                      "groups", "pad_mode", "act")] + [("slope", _c_f), ("in_act", _c_int)]


class ConvTDesc(ctypes.Structure):
    _fields_ = [(n, _c_int) for n in ("B", "Cin", "Lin", "Cout", "K", "stride", "pad", "act")] + \
               [("slope", _c_f), ("in_act", _c_int)]


L1_MULTI_MAX = 24


class L1MultiDesc(ctypes.Structure):            # ms_l1_multi_desc
    _fields_ = [("count", _c_int), ("reserved", _c_int),
                ("r", _vp * L1_MULTI_MAX), ("f", _vp * L1_MULTI_MAX), ("gf", _vp * L1_MULTI_MAX),
                ("n", ctypes.c_int64 * L1_MULTI_MAX), ("w", _c_f * L1_MULTI_MAX)]


JUDGE_MULTI_MAX = 8
JUDGE_MULTI_NMAX = 1 << 20
JUDGE_HINGE_D, JUDGE_NEG_MEAN = 0, 1


class JudgeMultiDesc(ctypes.Structure):         # ms_judge_multi_desc
    _fields_ = [("count", _c_int), ("kind", _c_int),
                ("r", _vp * JUDGE_MULTI_MAX), ("f", _vp * JUDGE_MULTI_MAX),
                ("gr", _vp * JUDGE_MULTI_MAX), ("gf", _vp * JUDGE_MULTI_MAX),
                ("n", ctypes.c_int64 * JUDGE_MULTI_MAX)]


CONV_PARTS_MAX = 3


class ConvParts(ctypes.Structure):              # ms_conv1d_parts
    _fields_ = [("count", _c_int), ("reserved", _c_int), ("B", _c_int * CONV_PARTS_MAX), ("Lin", _c_int * CONV_PARTS_MAX),
                ("x", _vp * CONV_PARTS_MAX), ("y", _vp * CONV_PARTS_MAX), ("gy", _vp * CONV_PARTS_MAX),
                ("y_act", _vp * CONV_PARTS_MAX), ("gx_add", _vp * CONV_PARTS_MAX), ("gx", _vp * CONV_PARTS_MAX)]


WGRAD_MULTI_MAX = 8


class WgradMultiDesc(ctypes.Structure):         # ms_wgrad_multi_desc
    _fields_ = [("count", _c_int), ("reserved", _c_int),
                ("conv", ConvDesc * WGRAD_MULTI_MAX),
                ("x", _vp * WGRAD_MULTI_MAX), ("gy", _vp * WGRAD_MULTI_MAX), ("y_act", _vp * WGRAD_MULTI_MAX),
                ("gw", _vp * WGRAD_MULTI_MAX), ("gb", _vp * WGRAD_MULTI_MAX),
                ("beta", _c_f * WGRAD_MULTI_MAX),
                ("xmax", _vp * WGRAD_MULTI_MAX), ("gmax", _vp * WGRAD_MULTI_MAX), ("y_signs", _vp * WGRAD_MULTI_MAX)]


ATOM_AMAX_N = 1024


ATOM_PACK_MAX = 16


class AtomDesc(ctypes.Structure):               # ms_atom_desc
    _fields_ = [("B", _c_int), ("C", _c_int), ("L", _c_int), ("dil", _c_int), ("slope", _c_f)]


STACK_MAX = 3


class StackDesc(ctypes.Structure):              # ms_stack_desc
    _fields_ = [("B", _c_int), ("C", _c_int), ("L", _c_int), ("count", _c_int), ("dil", _c_int * STACK_MAX),
                ("slope", _c_f)]


class AtomPackDesc(ctypes.Structure):           # ms_atom_pack_desc
    _fields_ = [("count", _c_int), ("reserved", _c_int), ("C", _c_int * ATOM_PACK_MAX),
                ("w0", _vp * ATOM_PACK_MAX), ("w1", _vp * ATOM_PACK_MAX), ("image", _vp * ATOM_PACK_MAX),
                ("backward", _c_int * ATOM_PACK_MAX)]


LINES_MAX_PHASES, LINES_MAX_TAPS = 2, 3


class LinesDesc(ctypes.Structure):              # ms_lines_desc
    _fields_ = [("B", _c_int), ("H", _c_int), ("C", _c_int), ("W", _c_int), ("phases", _c_int), ("taps", _c_int),
                ("dy", _c_int * (LINES_MAX_PHASES * LINES_MAX_TAPS))]


WN_MULTI_MAX = 64


class WnMultiDesc(ctypes.Structure):            # ms_wn_multi_desc
    _fields_ = [("count", _c_int), ("reserved", _c_int),
                ("v", _vp * WN_MULTI_MAX), ("g", _vp * WN_MULTI_MAX), ("out", _vp * WN_MULTI_MAX),
                ("gv", _vp * WN_MULTI_MAX), ("gg", _vp * WN_MULTI_MAX),
                ("rows", _c_int * WN_MULTI_MAX), ("cols", _c_int * WN_MULTI_MAX)]


PROFILE_NAME_MAX = 160


class ProfileRecord(ctypes.Structure):          # ms_profile_record
    _fields_ = [("kernels", _c_int), ("products", _c_int), ("device_us", ctypes.c_double),
                ("kernel", ctypes.c_char * PROFILE_NAME_MAX)]


# name -> (restype, argtypes); every symbol include/msynth.h declares
SIGNATURES = {
    "ms_version": (_c_int, []),
    "ms_status_string": (ctypes.c_char_p, [_c_int]),
    "ms_conv1d_out_len": (_c_int, [ctypes.POINTER(ConvDesc)]),
    "ms_conv1d_fwd": (_c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ms_conv1d_bwd_data": (_c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ms_conv1d_bwd_weight": (_c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _c_f, _vp, _sz, _vp]),
    "ms_conv1d_parts_launches": (_c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(ConvParts), _c_int, _c_int]),
    "ms_conv1d_parts_workspace_bytes": (_sz, [ctypes.POINTER(ConvDesc), ctypes.POINTER(ConvParts), _c_int, _c_int]),
    "ms_conv1d_parts_fwd": (_c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(ConvParts), _vp, _vp, _vp, _vp, _sz, _vp]),
    "ms_conv1d_parts_bwd_data": (_c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(ConvParts), _vp, _vp, _vp, _sz, _vp]),
    "ms_conv1d_parts_bwd_weight": (_c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(ConvParts), _vp, _vp, _c_f, _vp, _sz, _vp]),
    "ms_conv1d_bwd_weight_multi_workspace_bytes": (_sz, [ctypes.POINTER(WgradMultiDesc)]),
    "ms_conv1d_bwd_weight_multi": (_c_int, [ctypes.POINTER(WgradMultiDesc), _vp, _sz, _vp]),
    "ms_residual_atom_image_bytes": (_sz, [_c_int]),
    "ms_residual_atom_supported": (_c_int, [ctypes.POINTER(AtomDesc)]),
    "ms_residual_atom_pack_multi": (_c_int, [ctypes.POINTER(AtomPackDesc), _vp]),
    "ms_residual_atom_publishes_amax": (_c_int, []),
    "ms_residual_atom_fwd": (_c_int, [ctypes.POINTER(AtomDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ms_residual_stack_supported": (_c_int, [ctypes.POINTER(StackDesc)]),
    "ms_residual_stack_signs_supported": (_c_int, [ctypes.POINTER(StackDesc)]),
    "ms_residual_atom_sign_words": (_sz, [ctypes.POINTER(AtomDesc)]),
    "ms_residual_atom_fwd_signs": (_c_int, [ctypes.POINTER(AtomDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ms_residual_atom_bwd_data_signs": (_c_int, [ctypes.POINTER(AtomDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ms_residual_stack_fwd": (_c_int, [ctypes.POINTER(StackDesc), _vp, _vp, _vp, _vp, _vp, _vp]),
    "ms_residual_atom_bwd_supported": (_c_int, [ctypes.POINTER(AtomDesc)]),
    "ms_residual_atom_bwd_data": (_c_int, [ctypes.POINTER(AtomDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ms_conv1d_img_bytes": (_sz, [ctypes.POINTER(ConvDesc)]),
    "ms_conv1d_img_workspace_bytes": (_sz, [ctypes.POINTER(ConvDesc), _c_int]),
    "ms_conv1d_img_pack": (_c_int, [ctypes.POINTER(ConvDesc), _vp, _c_int, _vp, _vp]),
    "ms_conv1d_img_pack2": (_c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp]),
    "ms_conv1d_img_fwd": (_c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ms_conv1d_img_bwd_data": (_c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ms_conv1d_workspace_bytes": (_sz, [ctypes.POINTER(ConvDesc), _c_int]),
    "ms_conv1d_kernel_name": (ctypes.c_char_p, [ctypes.POINTER(ConvDesc), _c_int]),
    "ms_convt1d_kernel_name": (ctypes.c_char_p, [ctypes.POINTER(ConvTDesc), _c_int]),
    "ms_debug_install_crash_handler": (_c_int, []),
    "ms_profile_kernels": (None, [_c_int]),
    "ms_profile_take": (_c_int, [ctypes.POINTER(ProfileRecord)]),
    "ms_convt1d_out_len": (_c_int, [ctypes.POINTER(ConvTDesc)]),
    "ms_convt1d_fwd": (_c_int, [ctypes.POINTER(ConvTDesc), _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ms_convt1d_bwd_data": (_c_int, [ctypes.POINTER(ConvTDesc), _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ms_convt1d_bwd_weight": (_c_int, [ctypes.POINTER(ConvTDesc), _vp, _vp, _vp, _vp, _vp, _c_f, _vp, _sz, _vp]),
    "ms_convt1d_workspace_bytes": (_sz, [ctypes.POINTER(ConvTDesc), _c_int]),
    "ms_convt1d_img_bytes": (_sz, [ctypes.POINTER(ConvTDesc)]),
    "ms_convt1d_img_pack": (_c_int, [ctypes.POINTER(ConvTDesc), _vp, _vp, _vp]),
    "ms_convt1d_img_workspace_bytes": (_sz, [ctypes.POINTER(ConvTDesc)]),
    "ms_convt1d_img_fwd": (_c_int, [ctypes.POINTER(ConvTDesc), _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ms_convt1d_bwd_img_bytes": (_sz, [ctypes.POINTER(ConvTDesc)]),
    "ms_convt1d_bwd_img_workspace_bytes": (_sz, [ctypes.POINTER(ConvTDesc)]),
    "ms_convt1d_bwd_img_pack": (_c_int, [ctypes.POINTER(ConvTDesc), _vp, _vp, _vp]),
    "ms_convt1d_bwd_img_data": (_c_int, [ctypes.POINTER(ConvTDesc), _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ms_avg_pool1d_4_2_2_fwd": (_c_int, [_vp, _vp, _c_i64, _c_int, _vp]),
    "ms_avg_pool1d_4_2_2_bwd": (_c_int, [_vp, _vp, _vp, _c_i64, _c_int, _vp]),
    "ms_avg_pool1d_4_2_1_fwd": (_c_int, [_vp, _vp, _c_i64, _c_int, _vp]),
    "ms_avg_pool1d_4_2_1_bwd": (_c_int, [_vp, _vp, _vp, _c_i64, _c_int, _vp]),
    "ms_avg_pool1d_k_fwd": (_c_int, [_vp, _vp, _c_i64, _c_int, _c_int, _vp]),
    "ms_avg_pool1d_k_bwd": (_c_int, [_vp, _vp, _c_i64, _c_int, _c_int, _vp]),
    "ms_weight_norm_fwd": (_c_int, [_vp, _vp, _vp, _c_int, _c_int, _vp]),
    "ms_weight_norm_bwd": (_c_int, [_vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_f, _vp]),
    "ms_judge_loss_multi_fwd": (_c_int, [ctypes.POINTER(JudgeMultiDesc), _vp, _vp]),
    "ms_judge_loss_multi_bwd": (_c_int, [ctypes.POINTER(JudgeMultiDesc), _vp, _c_f, _vp]),
    "ms_weight_norm_multi_fwd": (_c_int, [ctypes.POINTER(WnMultiDesc), _vp]),
    "ms_weight_norm_multi_bwd": (_c_int, [ctypes.POINTER(WnMultiDesc), _c_f, _vp]),
    "ms_lines_stack": (_c_int, [ctypes.POINTER(LinesDesc), _vp, _vp, _vp]),
    "ms_lines_fold": (_c_int, [ctypes.POINTER(LinesDesc), _vp, _vp, _vp]),
    "ms_lines_interleave": (_c_int, [_vp, _vp, _c_i64, _c_int, _c_i64, _c_int, _vp]),
    "ms_act_bwd": (_c_int, [_vp, _vp, _vp, _c_i64, _c_int, _c_f, _vp]),
    "ms_add": (_c_int, [_vp, _vp, _vp, _c_i64, _vp]),
    "ms_add_act": (_c_int, [_vp, _vp, _vp, _c_i64, _c_int, _c_f, _vp]),
    "ms_reduce_workspace_bytes": (_sz, [_c_i64]),
    "ms_hinge_d_fwd": (_c_int, [_vp, _vp, _c_i64, _vp, _vp, _sz, _vp]),
    "ms_hinge_d_bwd": (_c_int, [_vp, _vp, _c_i64, _vp, _c_f, _vp, _vp, _vp]),
    "ms_neg_mean_fwd": (_c_int, [_vp, _c_i64, _vp, _vp, _sz, _vp]),
    "ms_neg_mean_bwd": (_c_int, [_c_i64, _vp, _c_f, _vp, _vp]),
    "ms_l1_mean_fwd": (_c_int, [_vp, _vp, _c_i64, _vp, _vp, _sz, _vp]),
    "ms_l1_mean_bwd": (_c_int, [_vp, _vp, _c_i64, _vp, _c_f, _vp, _c_int, _vp]),
    "ms_l1_mean_multi_workspace_bytes": (_sz, [ctypes.POINTER(L1MultiDesc)]),
    "ms_l1_mean_multi_fwd": (_c_int, [ctypes.POINTER(L1MultiDesc), _vp, _vp, _sz, _vp]),
    "ms_l1_mean_multi_bwd": (_c_int, [ctypes.POINTER(L1MultiDesc), _vp, _c_f, _vp]),
    "ms_l1_mean_multi_fwd_bwd_workspace_bytes": (_sz, [ctypes.POINTER(L1MultiDesc)]),
    "ms_l1_mean_multi_fwd_bwd": (_c_int, [ctypes.POINTER(L1MultiDesc), _vp, _c_f, _vp, _sz, _vp]),
    "ms_ls_g_fwd": (_c_int, [_vp, _c_i64, _vp, _vp, _sz, _vp]),
    "ms_ls_g_bwd": (_c_int, [_vp, _c_i64, _vp, _c_f, _vp, _vp]),
    "ms_ls_d_fwd": (_c_int, [_vp, _vp, _c_i64, _vp, _vp, _sz, _vp]),
    "ms_ls_d_bwd": (_c_int, [_vp, _vp, _c_i64, _vp, _c_f, _vp, _vp, _vp]),
    "ms_weighted_sum": (_c_int, [_vp, _vp, _c_int, _vp, _vp]),
    "ms_adam_step": (_c_int, [_vp, _vp, _vp, _vp, _c_i64, _c_f, _c_f, _c_f, _c_f, _c_f, _vp, _vp]),
    "ms_audio2mel_frames": (_c_int, [_c_int, _c_int, _c_int]),
    "ms_audio2mel_fwd": (_c_int, [_vp, _c_int, _c_int, _vp, _c_int, _c_int, _vp, _c_int, _vp, _vp]),
    "ms_resample_sinc_fwd": (_c_int, [_vp, _c_int, _c_int, _vp, _c_int, ctypes.c_double, _vp, _vp, _c_int, _c_int, _vp]),
    "ms_peak_normalize": (_c_int, [_vp, _c_int, _c_int, _c_f, _vp, _vp]),
    "ms_comm_unique_id": (_c_int, [_vp]),
    "ms_comm_init": (_c_int, [_vp, _c_int, _c_int, ctypes.POINTER(_vp)]),
    "ms_comm_world": (_c_int, [_vp]),
    "ms_comm_rank": (_c_int, [_vp]),
    "ms_allreduce_f32": (_c_int, [_vp, _vp, _c_i64, _vp]),
    "ms_comm_destroy": (_c_int, [_vp]),
    "ms_comm_last_error": (ctypes.c_char_p, []),
}
COMM_ID_BYTES = 128

_LIB = None


def load():
    """Loads libmsynth_hip.so (after torch, so both share torch's HIP runtime)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "featuresynth (MI355X build): %s is missing; build it with "
                "`make -C music-synthesis_amd/csrc` or __graft_entry__.build(). There is no CPU "
                "or PyTorch fallback for the hot path." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


def check(rc, what):
    if rc != 0:
        msg = load().ms_status_string(rc)
        raise RuntimeError("%s failed: %s (ms_status %d)" % (what, msg.decode() if msg else "?", rc))


def ptr(t):
    return None if t is None else t.data_ptr()


def require(t, name):
    """The hot path takes contiguous fp32 tensors resident on a HIP device, nothing else."""
    if not isinstance(t, torch.Tensor):
        raise RuntimeError("%s: expected a torch.Tensor, got %r" % (name, type(t)))
    if not t.is_cuda:
        raise RuntimeError(
            "%s: featuresynth (MI355X build) only runs on a HIP device tensor; got device %s. "
            "There is no CPU fallback." % (name, t.device))
    if t.dtype != torch.float32:
        raise RuntimeError("%s: expected float32, got %s" % (name, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("%s: expected a contiguous tensor" % name)
    return t


def stream():
    return torch.cuda.current_stream().cuda_stream


def workspace(nbytes, device):
    """Scratch for one call, from torch's caching allocator (stream-ordered, and when a hipGraph
    is being captured it comes from that graph's private pool, so replays stay valid)."""
    if nbytes <= 0:
        return None
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


# ---- optional per-launch timing (bench.py's roofline leg; off in normal operation)
PROFILE = None      # list of (symbol, cost dict, start event, end event) while enabled


def profile_begin():
    """Per-launch timing on: every C-ABI call is recorded with (a) the device time of the kernels it launched, taken from the
    dispatches' own begin / end timestamps (ms_profile_kernels: what rocprofv3 reports), and (b) a stream event pair around
    the call (reads 0.2-2.5 us more per kernel: launch latency).  The calling thread's launches are serialised while on."""
    global PROFILE
    PROFILE = []
    load().ms_profile_kernels(1)


def profile_end(calibrate=False):
    """-> list of (symbol, cost, milliseconds); synchronises the device.  The milliseconds are the device time of the
    call's kernels (cost["event_ms"] keeps the stream-event reading, cost["kernels"] their number).

    calibrate=True also returns the reading of an EMPTY stream event pair (the records are stream commands themselves)."""
    global PROFILE
    rec, PROFILE = PROFILE, None
    load().ms_profile_kernels(0)
    empty = []
    if calibrate:
        for _ in range(64):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); e1.record()
            empty.append((e0, e1))
    torch.cuda.synchronize()
    out = []
    for name, cost, e0, e1, dev_us, nk in rec:
        ev_ms = e0.elapsed_time(e1)
        cost["event_ms"], cost["kernels"] = ev_ms, nk
        out.append((name, cost, dev_us * 1e-3 if nk else ev_ms))
    if not calibrate:
        return out
    gaps = sorted(e0.elapsed_time(e1) for e0, e1 in empty)
    return out, gaps[len(gaps) // 2]


def call(name, cost_fn, *args):
    """Launches C-ABI entry `name` on the current stream; raises on a non-zero status."""
    fn = getattr(load(), name)
    if PROFILE is None:
        check(fn(*args), name)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lib = load()
    rec = ProfileRecord()
    lib.ms_profile_take(ctypes.byref(rec))          # (drops what an unrecorded call may have left)
    e0.record()
    rc = fn(*args)
    e1.record()
    check(rc, name)
    cost = cost_fn() if cost_fn else {}
    check(lib.ms_profile_take(ctypes.byref(rec)), "ms_profile_take")
    noted = rec.kernel.decode()
    if noted:                   # the instantiation the launcher really dispatched (split-K aware) and its arithmetic
        cost["kernel"] = noted
    cost["products"] = int(rec.products)
    PROFILE.append((name, cost, e0, e1, float(rec.device_us), int(rec.kernels)))
