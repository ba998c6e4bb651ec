"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" on CPU for tests).  The reference is single-process (SURVEY.md 2, rows
22-23), so this layer has no reference counterpart: batch elements are independent and every
loss is a mean over batch x time, hence averaging the per-rank gradients of equal shards equals
the global-batch gradient up to fp32 summation order.

The exchange is a summing all-reduce of the stepped network's flat gradient bucket per trainer
call (D grads after a D-step, 22.6 MB; G grads after a G-step, 18.1 MB), cut in two slices at a
parameter boundary so that the slice whose gradients are complete first travels under the rest of
the backward pass (train.py); the 1/world scale is folded into the fused Adam kernel
(FlatAdam.grad_scale).

Two transports for the bucket, same collective:
  * "torch" (default): torch.distributed's process group (ProcessGroupNCCL = RCCL);
  * "abi" (MSYNTH_COMM=abi): the C ABI's own RCCL communicator (ms_comm_init / ms_allreduce_f32,
    include/msynth.h) on a dedicated HIP stream; bootstrap = broadcast of the 128-byte unique id
    through the torch.distributed group.
"""
import ctypes
import os

import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def init_from_env(backend=None, force=False):
    """Initialises the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun).
    force=True also builds a one-rank group (exercises the RCCL path on a single GPU)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if (ws <= 1 and not force) or (dist.is_available() and dist.is_initialized()):
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend, device_id=torch.device(
            "cuda", int(os.environ.get("LOCAL_RANK", "0"))))
    else:
        dist.init_process_group(backend)


class _Done:
    def wait(self):
        return True


class _StreamWork:
    """An all-reduce enqueued on a side HIP stream: wait() makes the CURRENT stream wait for it
    (no host block), like ProcessGroupNCCL's Work.wait()."""

    def __init__(self, stream):
        self.event = torch.cuda.Event()
        self.event.record(stream)

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)
        return True


# ---- C-ABI communicator (MSYNTH_COMM=abi)
_ABI = {"comm": None, "stream": None}


def _abi_wanted():
    return os.environ.get("MSYNTH_COMM", "torch") == "abi"


def abi_comm():
    """The process's ms_comm_t (created on first use, collectively over all ranks)."""
    if _ABI["comm"] is None:
        from ._ops import lib as L
        lib = L.load()
        # the C ABI binds the RCCL instance already in the process: make sure torch's copy is
        rccl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(rccl):
            ctypes.CDLL(rccl, mode=ctypes.RTLD_GLOBAL)
        ident = torch.zeros(L.COMM_ID_BYTES, dtype=torch.uint8)
        if rank() == 0:
            buf = (ctypes.c_char * L.COMM_ID_BYTES)()
            L.check(lib.ms_comm_unique_id(buf), "ms_comm_unique_id (%s)" % lib.ms_comm_last_error().decode())
            ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        if is_distributed():
            dev = ident.cuda() if dist.get_backend() == "nccl" else ident
            dist.broadcast(dev, 0)
            ident = dev.cpu()
        raw = (ctypes.c_char * L.COMM_ID_BYTES).from_buffer_copy(bytes(ident.numpy().tobytes()))
        out = ctypes.c_void_p()
        L.check(lib.ms_comm_init(raw, world_size(), rank(), ctypes.byref(out)),
                "ms_comm_init (%s)" % lib.ms_comm_last_error().decode())
        _ABI["comm"] = out
        _ABI["stream"] = torch.cuda.Stream()
    return _ABI["comm"]


def abi_comm_destroy():
    if _ABI["comm"] is not None:
        from ._ops import lib as L
        torch.cuda.synchronize()
        L.load().ms_comm_destroy(_ABI["comm"])
        _ABI["comm"] = None


def _abi_allreduce(flat):
    from ._ops import lib as L
    comm = abi_comm()
    side = _ABI["stream"]
    side.wait_stream(torch.cuda.current_stream())
    L.require(flat, "all-reduce bucket")
    lib = L.load()
    rc = lib.ms_allreduce_f32(comm, flat.data_ptr(), flat.numel(), side.cuda_stream)
    if rc != 0:
        raise RuntimeError("ms_allreduce_f32 failed: %s" % lib.ms_comm_last_error().decode())
    flat.record_stream(side)
    return _StreamWork(side)


def allreduce_sum_async(flat, force=False):
    """Starts the in-place summing all-reduce of a flat fp32 tensor behind the work already queued
    on the current stream and returns a handle; handle.wait() orders the current stream behind it.
    Work queued on the current stream AFTER this call overlaps the collective."""
    if not (is_distributed() or force):
        return _Done()
    if _abi_wanted() and flat.is_cuda:
        return _abi_allreduce(flat)
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)


def allreduce_sum_(flat):
    """In-place summing all-reduce of one flat fp32 bucket (no-op for a single process)."""
    allreduce_sum_async(flat).wait()
    return flat


def broadcast_(flat, src=0):
    if is_distributed():
        dist.broadcast(flat, src)
    return flat


def allreduce_mean_scalar(t):
    """Mean over ranks of a 0-d tensor (loss reporting)."""
    if is_distributed():
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t = t / dist.get_world_size()
    return t


def barrier():
    if is_distributed():
        dist.barrier()
