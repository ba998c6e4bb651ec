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

    # 1b. the trainer's exchange: the bucket travels as two slices cut at a parameter boundary, the
    #     early (tail) slice first and asynchronously, then the late (head) slice; same result as one
    #     all-reduce of the whole bucket, and views of the bucket are reduced in place
    bucket = (torch.arange(64, dtype=torch.float32) + 1) * (rank + 1)
    whole = bucket.clone()
    _dist.allreduce_sum_(whole)
    w_early = _dist.allreduce_sum_async(bucket[40:])
    bucket[:40] += 0.0                                   # (work issued while the early slice travels)
    w_late = _dist.allreduce_sum_async(bucket[:40])
    w_early.wait(); w_late.wait()
    assert torch.equal(bucket, whole)

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


def test_graphed_step_eager_cut_sequence():
    """_GraphedStep (train/train.py) on an eager call: the body's cut() points run the `between` actions
    (the data-parallel all-reduces) in order, exactly once each, at the point of the cut."""
    from featuresynth.train.train import _GraphedStep
    log = []

    def body(s, f, cut):
        log.append("fwd+head")
        cut()
        log.append("tail")
        cut()
        log.append("adam")
        return {"loss": 1.0}

    step = _GraphedStep(body, [lambda: log.append("allreduce early"), lambda: log.append("allreduce late + wait")])
    x = torch.zeros(2, 1, 8)
    assert step(x, x) == {"loss": 1.0}
    assert log == ["fwd+head", "allreduce early", "tail", "allreduce late + wait", "adam"]
    bad = _GraphedStep(lambda s, f, cut: {"loss": 0.0}, [lambda: None])
    with pytest.raises(AssertionError):
        bad(x, x)


def test_bucket_split_points():
    """The cut points of the two buckets (graph.D_HEAD_PARAM / G_TAIL_PARAM) against the state_dict
    order of the reference's modules: D head = main.5 (k5 conv) + judge = 93 % of the bytes; G tail =
    main.1 (first conv) + main.3 (first transposed conv)."""
    import featuresynth as fs
    from featuresynth._ops import graph as G
    d = [k for k, _ in fs.MelGanDiscriminator().named_parameters()]
    g = [k for k, _ in fs.MelGanGenerator(32, 80).named_parameters()]
    assert d[G.D_HEAD_PARAM] == "disc.main.5.weight" and d[G.D_HEAD_PARAM:] == [
        "disc.main.5.weight", "disc.main.5.bias", "disc.judge.weight", "disc.judge.bias"]
    assert g[:G.G_TAIL_PARAM] == ["main.1.weight", "main.1.bias", "main.3.weight", "main.3.bias"]
    nd = [p.numel() for p in fs.MelGanDiscriminator().parameters()]
    assert sum(nd[G.D_HEAD_PARAM:]) / sum(nd) > 0.92


def test_single_process_helpers_are_noops():
    from featuresynth import _dist
    assert _dist.world_size() == 1 and _dist.rank() == 0 and not _dist.is_distributed()
    t = torch.ones(3)
    assert _dist.allreduce_sum_(t) is t and torch.equal(t, torch.ones(3))


def test_host_copies_only_in_the_final_segment():
    """Constraint recorded in DESIGN_HISTORY.md section 6 ("graph faults"): the device-to-host result copies of a
    multi-segment train step belong to its LAST segment; _to_host refuses anything else (on an eager call and
    under capture alike -- the guard sits in front of any device work)."""
    from featuresynth.train.train import _GraphedStep, _TrainerBase
    tr = _TrainerBase.__new__(_TrainerBase)

    def bad_body(s, f, cut):
        tr._to_host({})          # in front of a cut that is still to come
        cut()
        return {}

    tr._runner = _GraphedStep(bad_body, [lambda: None])
    x = torch.zeros(2, 1, 8)
    with pytest.raises(RuntimeError, match="last graph segment"):
        tr._runner(x, x)
    assert tr._runner.segment is None

    def good_body(s, f, cut):
        cut()
        return tr._to_host({})

    tr._runner = _GraphedStep(good_body, [lambda: None])
    assert tr._runner(x, x) == {}


def test_split_point_needs_module_parameter_order():
    """the sliced exchange's cut offset is only valid when the optimizer's bucket lists the module's
    parameters in module order; any other optimizer gets a whole-bucket all-reduce (offset 0)."""
    import featuresynth as fs
    from featuresynth.train.train import _TrainerBase
    d = fs.MelGanDiscriminator()
    ps = list(d.parameters())

    class Opt:
        def __init__(self, params):
            self.param_groups = [{"params": params}]
    assert _TrainerBase._bucket_in_module_order(Opt(ps), d)
    assert not _TrainerBase._bucket_in_module_order(Opt(ps[::-1]), d)
    assert not _TrainerBase._bucket_in_module_order(Opt(ps[:-1]), d)
