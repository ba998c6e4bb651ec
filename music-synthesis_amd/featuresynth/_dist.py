"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" on CPU for tests).  The reference is single-process (SURVEY.md 2, rows
22-23), so this layer has no reference counterpart: batch elements are independent and every
loss is a mean over batch x time, hence averaging the per-rank gradients of equal shards equals
the global-batch gradient up to fp32 summation order.

The exchange is one summing all-reduce of the stepped network's flat gradient bucket per trainer
call (D grads after a D-step, 22.6 MB; G grads after a G-step, 18.1 MB); the 1/world scale is
folded into the fused Adam kernel (FlatAdam.grad_scale).
"""
import os

import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def init_from_env(backend=None):
    """Initialises the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws <= 1 or (dist.is_available() and dist.is_initialized()):
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend, device_id=torch.device(
            "cuda", int(os.environ.get("LOCAL_RANK", "0"))))
    else:
        dist.init_process_group(backend)


def allreduce_sum_(flat):
    """In-place summing all-reduce of one flat fp32 bucket (no-op for a single process)."""
    if is_distributed():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
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
