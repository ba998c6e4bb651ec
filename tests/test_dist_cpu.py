"""Data-parallel path on CPU: two gloo ranks.  Checks the flat-bucket all-reduce plumbing and the
property the DP design rests on: averaging the shard gradients of equal shards
reproduces the global-batch gradient, so the Adam update is the same."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "music-synthesis_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(2)
    from featuresynth import _dist
    from featuresynth._synthetic import synthetic_features, synthetic_samples, synthetic_state_dict
    from oracle import oracle as O
    from oracle import torch_graph as TG
    _dist.init_from_env("gloo")
    assert _dist.world_size() == world and _dist.rank() == rank and _dist.is_distributed()

    # 1. flat-bucket plumbing
    flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
    _dist.allreduce_sum_(flat)
    assert torch.equal(flat, torch.arange(10, dtype=torch.float32) * 3)
    b = torch.full((4,), float(rank))
    _dist.broadcast_(b, 0)
    assert float(b.sum()) == 0.0
    m = _dist.allreduce_mean_scalar(torch.tensor(float(rank)))
    assert abs(float(m) - 0.5) < 1e-7

    # 2. shard-mean == global-batch gradient for the D-step (tiny window to stay fast)
    B, T = 2, 4
    gsd = synthetic_state_dict(O.generator_param_shapes(80), seed=7, bias_scale=0.02)
    dsd = synthetic_state_dict(O.discriminator_param_shapes(), seed=8, bias_scale=0.02)
    samples = [synthetic_samples(B, T * 256, rank=r) for r in range(world)]
    feats = [synthetic_features(B, 80, T, rank=r) for r in range(world)]

    def d_grads(s, f):
        gp, dp = TG.to_params(gsd, False), TG.to_params(dsd, True)
        with torch.no_grad():
            fake = TG.generator(gp, torch.from_numpy(f))
        _, fj = TG.discriminator(dp, fake)
        _, rj = TG.discriminator(dp, torch.from_numpy(s))
        TG.disc_loss(rj, fj).backward()
        return torch.cat([dp[k].grad.reshape(-1) for k in dp])

    local = d_grads(samples[rank], feats[rank])
    _dist.allreduce_sum_(local)
    local /= world
    if rank == 0:
        glob = d_grads(np.concatenate(samples), np.concatenate(feats))
        err = float((local - glob).norm() / glob.norm())
        np.save(os.path.join(out_dir, "err.npy"), np.array([err]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    err = float(np.load(str(tmp_path / "err.npy"))[0])
    assert err < 2e-4, err      # fp32 summation order only


def test_single_process_helpers_are_noops():
    from featuresynth import _dist
    assert _dist.world_size() == 1 and _dist.rank() == 0 and not _dist.is_distributed()
    t = torch.ones(3)
    assert _dist.allreduce_sum_(t) is t and torch.equal(t, torch.ones(3))
