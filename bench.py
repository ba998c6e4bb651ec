#!/usr/bin/env python3
"""Headline benchmark: stage-2 GAN train-step audio samples/sec (22.05 kHz, 8192-sample window).

    python bench.py --gpus 1 --steps 20 --warmup 6
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one trainer call on one batch, exactly like the reference's training_loop
(train/train.py:80-92): calls alternate D, G, D, G ... (experiment/experiment.py:141-144).
Workload = BASELINE.json configs[2]/[3]: B = 32 per GPU (weak scaling), 80-mel x 32-frame
features -> 8192-sample windows, synthetic inputs already resident in HBM, seed-7 N(0, 0.02)
weights identical on every rank, FlatAdam(1e-4, (0.5, 0.9)).  The timed region includes the
loss .item() syncs and the G-step's D2H copy of `fake` (they are part of the reference's trainer
contract, train.py:39-42,74).  value = world * B * 8192 * K / max-over-ranks(wall time).

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      dominant kernel (by device time in an instrumented eager D+G pass): algorithmic
                FLOPs per launch / mean DEVICE duration of the launch (the dispatches' own begin / end
                timestamps through hipExtLaunchKernelGGL events: what rocprofv3 reports; the stream-event
                reading around the call is kept beside it), against the ceiling of the pipe the kernel
                runs on.  The launcher of every dense kernel records, with the kernel's name, how many
                16-bit matrix-pipe products it spends per fp32 multiply (ms_profile_record.products):
                6 -> 2500 / 6 = 416.7 TFLOP/s (exact bf16 x 3 split), 3 -> 2500 / 3 = 833.3 (block-scaled
                fp16 x 2 split), 0 -> 157.3 (fp32-input MFMA / vector FMA); streams: 8 TB/s
  step_roofline fractions of the measured step: `frac` vs sum max(bytes/8 TB/s, flops/157.3 TFLOP/s) over
                the layer spec (SURVEY 8(d)'s contract); `frac_as_run` vs the same sum over the launches
                that really ran, each priced on the pipe its launcher recorded; three memory-bound
                fractions (layer-granular bytes, the fused byte model of the program as scheduled, and
                SURVEY 8(d)'s 135 MB per element per call)
  value_exact   the same bench in a child process with the operand-scheme switches at NP=3: every fp32
                multiply as six exact bf16 partial products (no 22-bit operand anywhere)
  cpu_baseline  the torch-functional CPU restatement of the reference graph (oracle/torch_graph.py)
                timed on this host's cores on a bounded sample (N=1 runs only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "music-synthesis_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK = 8.0e12        # B/s   (MI355X_MICROARCH.md: HBM3E peak)
PMC_FILE = "r05_pmc_traffic.json"      # profiles/: TCC traffic of every kernel (tools/pmc_traffic.sh), stamped with code_version()
F32_PEAK = 157.3e12      # FLOP/s (fp32 vector == fp32-input MFMA peak)
WINDOW = 8192


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def code_version():
    """Hash of the kernel sources and the scheduling layer: stamps profiles/*.json so that a PMC traffic file
    is only quoted for the code it was measured on (the GPU box has no .git)."""
    import glob
    import hashlib
    h = hashlib.sha1()
    pkg = os.path.join(ROOT, "music-synthesis_amd")
    files = sorted(glob.glob(os.path.join(pkg, "csrc", "*.hip")) + glob.glob(os.path.join(pkg, "csrc", "*.h")) +
                   glob.glob(os.path.join(pkg, "featuresynth", "_ops", "*.py")) +
                   [os.path.join(pkg, "featuresynth", "train", "train.py"), os.path.join(ROOT, "include", "msynth.h")])
    for f in files:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:12]


BF16_PEAK = 2500.0e12    # FLOP/s dense bf16 / fp16 MFMA (MI355X_MICROARCH.md)

DTYPE = ("f32 storage and accumulate; multiplies on the 16-bit matrix pipe with split fp32 operands: block-scaled fp16 x 2 "
         "(22 significand bits, 3 products per multiply) in the atoms, the k5 layer, the grouped convs, the stride-8 transposed "
         "forward and the atom weight gradients; exact bf16 x 3 (6 products) or fp32-input MFMA elsewhere")
DTYPE_EXACT = "f32 storage and accumulate; every multiply fp32-exact (bf16 x 3 split, 6 products, or fp32-input MFMA / vector FMA)"


# switches that put every 22-bit (fp16 x 2) kernel back on fp32-exact arithmetic (DESIGN_HISTORY.md "Tuning switches")
EXACT_ENV = {"MSYNTH_ATOM_NP": "3", "MSYNTH_C5_NP": "3", "MSYNTH_W5_NP": "3", "MSYNTH_WROWS3_NP": "3", "MSYNTH_GCONV3": "0", "MSYNTH_CONVTIMG": "0"}


def pipe_peak(products):
    """FLOP/s ceiling of a kernel by the arithmetic its launcher recorded (ms_profile_record.products)."""
    return BF16_PEAK / products if products else F32_PEAK


PIPE_TEXT = {6: "bf16 MFMA, exact 3-way operand split, 6 products per fp32 multiply, fp32 accumulate: peak = 2500 / 6",
             3: "fp16 MFMA, block-scaled 2-way operand split (22 significand bits), 3 products per fp32 multiply, "
                "fp32 accumulate: peak = 2500 / 3",
             0: "fp32-input MFMA / vector FMA: peak = 157.3"}


def host_cpu_share(cap=16):
    """Threads this process may really use: affinity mask, cgroup CPU quota, and the GPU box's
    per-GPU share (16) -- os.cpu_count() reports the whole host and oversubscribes."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def launch_workers(args):
    """`python bench.py --gpus N` outside torchrun: this process never touches the GPU (no torch import);
    it starts N fresh worker processes, one per GPU, through torch.distributed.run and relays rank 0's
    JSON line (the workers inherit stdout) and their exit status."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("[bench] launching %d workers: %s" % (args.gpus, " ".join(cmd)))
    return subprocess.call(cmd, env=env)


def child_bench(args, env_extra, what):
    """The same bench (timed region only: no roofline / CPU / config-2 legs) in a fresh child process with extra environment
    switches -- the library reads its operand-scheme switches once per process.  The parent is idle meanwhile.
    -> the child's JSON line as a dict, or {"error": ...}."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.update(env_extra)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--batch", str(args.batch), "--mels", str(args.mels), "--prime", str(args.prime), "--model", args.model,
           "--no-cpu-baseline", "--no-roofline", "--no-gforward", "--no-exact", "--no-dp-overhead"]
    log("[bench] child leg %s: %s" % (what, " ".join("%s=%s" % kv for kv in sorted(env_extra.items()))))
    try:
        out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    except subprocess.TimeoutExpired:
        return {"error": "child timed out"}
    for line in reversed(out.stdout.splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return {"error": "child exited %d: %s" % (out.returncode, out.stderr[-400:])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--mels", type=int, default=80)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--model", default="melgan", choices=["melgan", "realmelgan", "twostage"],
                    help="melgan = the north-star variant (headline); realmelgan = the weight-normed "
                         "variant of experiment/realmelgan.py (SURVEY.md 8(f) row 1, 128 mels); twostage = "
                         "BASELINE config 5: every step is one stage-1 trainer call (2-D conv mel GAN, "
                         "featureexperiment.py) plus one stage-2 trainer call (the headline vocoder, 128 mels)")
    ap.add_argument("--prime", type=int, default=60,
                    help="untimed calls in front of the warm-up steps, on top of the four that load code objects and capture the "
                         "graphs (clock ramp); the profiling scripts pass 0 to keep their traces at 15 D+G pairs")
    ap.add_argument("--no-exact", action="store_true",
                    help="skip the value_exact leg (the same bench in a child process on fp32-exact arithmetic)")
    ap.add_argument("--no-dp-overhead", action="store_true",
                    help="skip the dp_overhead leg (the N = 1 step under the data-parallel control flow, child process)")
    ap.add_argument("--no-gforward", action="store_true",
                    help="skip the BASELINE config-2 leg (generator forward, B=1): keeps its B=1 dispatches out "
                         "of a rocprofv3 trace of the train step")
    args = ap.parse_args()

    # decided before torch is imported: a multi-GPU run not started by torchrun launches its own workers
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_workers(args))

    import numpy as np
    import torch
    import featuresynth as fs
    from featuresynth import _dist
    from featuresynth import _workload as W
    from featuresynth import loss as LS
    from featuresynth._ops import lib as L
    from featuresynth._synthetic import (module_param_shapes, synthetic_features,
                                         synthetic_samples, synthetic_state_dict)
    from featuresynth.train import DiscriminatorTrainer, GeneratorTrainer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("--gpus %d != WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the hot path)")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # MSYNTH_BENCH_SHARE_GPU=1: rehearsal of the N > 1 control flow on a one-GPU box (every rank on
    # cuda:0, gloo instead of RCCL); the line it prints is marked and is not a measurement
    share = os.environ.get("MSYNTH_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # MSYNTH_DP_FORCE=1 (the dp_overhead leg): the data-parallel control flow over a one-rank RCCL communicator
    _dist.init_from_env("gloo" if share else "nccl", force=os.environ.get("MSYNTH_DP_FORCE") == "1")
    rank = _dist.rank()
    L.load()
    if world > 1:
        log("[bench] rank %d/%d on %s: process group backend %s (%s), gradient transport %s" % (
            rank, _dist.world_size(), torch.cuda.get_device_name(device), torch.distributed.get_backend(),
            "RCCL over xGMI" if not share else "rehearsal", os.environ.get("MSYNTH_COMM", "torch")))

    B, T = args.batch, WINDOW // 256
    gen_loss = LS.mel_gan_gen_loss
    if args.model == "realmelgan":
        from featuresynth.experiment import realmelgan as R
        args.mels = 128
        g, d = R.Generator(args.mels, 32, 3), R.Discriminator(3, 16, 4, 4)
        gen_loss = R.mel_gan_gen_loss
        args.no_roofline = args.no_cpu_baseline = True     # priced for the headline model only
    else:
        if args.model == "twostage":
            args.mels = 128                     # the stage-1 generator emits 128-bin spectrograms
            args.no_roofline = args.no_cpu_baseline = True     # priced for the headline model only
        g = fs.MelGanGenerator(T, args.mels)
        d = fs.MelGanDiscriminator()
    gsd = synthetic_state_dict(module_param_shapes(g), seed=7)
    dsd = synthetic_state_dict(module_param_shapes(d), seed=7)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in gsd.items()})
    d.load_state_dict({k: torch.from_numpy(v) for k, v in dsd.items()})
    g.to(device); d.to(device)
    g_optim = fs.FlatAdam(g.parameters(), lr=1e-4, betas=(0.5, 0.9))
    d_optim = fs.FlatAdam(d.parameters(), lr=1e-4, betas=(0.5, 0.9))
    gt = GeneratorTrainer(g, g_optim, d, d_optim, gen_loss)
    dt = DiscriminatorTrainer(g, g_optim, d, d_optim, LS.mel_gan_disc_loss)

    nbatches = 4   # distinct pre-staged batches, cycled
    batches = [(torch.from_numpy(synthetic_samples(B, WINDOW, rank=rank * 16 + i)).to(device),
                torch.from_numpy(synthetic_features(B, args.mels, T, rank=rank * 16 + i)).to(device))
               for i in range(nbatches)]

    stage1 = None
    if args.model == "twostage":
        # The reference trains stage 1 on its own against a frozen vocoder (featureexperiment.py:94-100,280-284)
        # and has no joint step: here a step = the stage-1 trainer call of this phase (D or G) on a batch of
        # real spectrograms + fresh noise, followed by the stage-2 trainer call of the same phase.
        from featuresynth.experiment import TwoDimGeneratorFeatureExperiment
        torch.manual_seed(7)
        stage1 = TwoDimGeneratorFeatureExperiment(vocoder_network=g).to(device)
        s1_batches = []
        for i in range(nbatches):
            rng = np.random.default_rng(300 + rank * 16 + i)
            s1_batches.append((torch.from_numpy((rng.standard_normal((B, 128, 512)) * 0.5).astype(np.float32)).to(device),
                               torch.from_numpy(rng.standard_normal((B, 128, 1)).astype(np.float32)).to(device)))

    def call(i):
        s, f = batches[i % nbatches]
        if stage1 is not None:
            spec, noise = s1_batches[i % nbatches]
            r1 = (stage1.d_trainer if i % 2 == 0 else stage1.g_trainer).train(spec, noise)
            r2 = dt.train(s, f) if i % 2 == 0 else gt.train(s, f)
            r2.update({"s1_" + k: v for k, v in r1.items() if k != "fake"})
            return r2
        return dt.train(s, f) if i % 2 == 0 else gt.train(s, f)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    last = {}
    # priming (untimed, before the W warm-up steps): the first call of each trainer runs eagerly (loads code
    # objects, sizes the flat buckets) and the second captures its hipGraph; a small --warmup must not push
    # those one-off costs into the timed region
    # ... and the clocks: a 20-step timed region is 70 ms, shorter than the GPU's ramp under a fresh load (measured: 3.49 ms
    # per step over 20 steps, 3.45 over 600 on the same box) -- 60 more untimed calls (~0.2 s) before the W warm-up steps
    for i in range(4 + args.prime):
        last.update(call(i))
    for i in range(args.warmup):
        last.update(call(i))
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        last.update(call(i))
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    value = world * B * WINDOW * args.steps / elapsed
    graphs = bool(dt._runner and dt._runner.graphs and gt._runner and gt._runner.graphs)
    log("[bench] rank %d: %d steps in %.3f s -> %.4g samples/s (hipGraph replay: %s); d_loss %.5f g_loss %.5f"
        % (rank, args.steps, elapsed, value, graphs, last.get("d_loss", float("nan")),
           last.get("g_loss", float("nan"))))

    result = {
        "metric": "GAN train-step audio samples/sec (22.05 kHz, 8192-sample window)",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "prime": 4 + args.prime, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPE_EXACT if all(os.environ.get(k) == v for k, v in EXACT_ENV.items()) else DTYPE,
        "data": "synthetic",
        "config": {"workload": ("stage-2 GAN train step: alternating D/G trainer calls, MelGAN "
                                "generator + 3-scale discriminator + feature-matching loss "
                                "(BASELINE.json configs[2]%s)" % ("" if world == 1 else "/[3]"))
                               if args.model == "melgan" else
                               ("two-stage (BASELINE.json configs[4]): per step one stage-1 trainer call (2-D conv mel "
                                "GAN, 128 x 512 spectrograms, least-squares losses) + one stage-2 trainer call (the "
                                "headline vocoder at 128 mels); samples/s counts the stage-2 windows"
                                if args.model == "twostage" else
                                "weight-normed MelGAN of experiment/realmelgan.py (SURVEY.md 8(f) row 1), same step"),
                   "per_gpu_batch": B, "global_batch": world * B, "window": WINDOW,
                   "mels": args.mels, "optimizer": "FlatAdam(1e-4,(0.5,0.9))",
                   "parallelism": "dp%d" % world, "hipgraph": graphs,
                   "graph_segments": [len(e[0]) for e in dt._runner.graphs.values()] if graphs else [],
                   "gradient_exchange": None if world == 1 else
                   "%s all-reduce of the stepped net's flat bucket in two slices (early slice overlapped with "
                   "the rest of the backward), transport=%s" % ("gloo" if share else "RCCL",
                                                                os.environ.get("MSYNTH_COMM", "torch")),
                   "switches": {k: v for k, v in sorted(os.environ.items()) if k.startswith("MSYNTH_")}},
    }

    if share:
        result["rehearsal"] = "ranks share cuda:0 over gloo: control-flow check only, not a measurement"
    if rank == 0 and not args.no_roofline:
        # instrumented eager pass: every kernel the library launches carries its dispatch's begin / end timestamps
        # (ms_profile_kernels), plus a stream event pair around every C-ABI call for comparison
        s, f = batches[0]
        os.environ["MSYNTH_STREAMS"] = "0"      # serialise the discriminator scales: clean per-kernel times
        L.profile_begin()
        out_d = dt._fwd_bwd(s, f); d_optim.step()
        mark = len(L.PROFILE)
        out_g = gt._fwd_bwd(s, f); g_optim.step()
        rec, ev_ms = L.profile_end(calibrate=True)
        os.environ.pop("MSYNTH_STREAMS", None)
        del out_d, out_g
        # Device durations.  (r02 took stream-event readings minus the reading of an EMPTY event pair, 4.6-4.9 us, and
        # over-corrected by 8 %; r03 first took the raw event readings, which sit 0.2-2.5 us above the traced durations
        # depending on the box; now the durations ARE the dispatch timestamps, the event readings are only reported.)
        log("[bench] empty event pair: %.2f us (reported only)" % (ev_ms * 1e3))
        agg = {}
        for name, cost, ms in rec:
            k = cost.get("kernel") or name
            a = agg.setdefault(k, {"ms": 0.0, "n": 0, "flops": 0.0, "bytes": 0.0, "event_ms": 0.0, "products": 0, "pipe_s": 0.0})
            a["ms"] += ms; a["n"] += 1
            a["products"] = max(a["products"], cost.get("products", 0))
            a["pipe_s"] += cost.get("flops", 0) / pipe_peak(cost.get("products", 0))     # matrix / vector time at the pipe's ceiling
            a["event_ms"] += cost.get("event_ms", ms)
            a["flops"] += cost.get("flops", 0); a["bytes"] += cost.get("bytes", 0)
        tot_ms = sum(a["ms"] for a in agg.values())
        top = sorted(agg.items(), key=lambda kv: -kv[1]["ms"])
        log("[bench] instrumented eager D+G pass: %.2f ms of kernels in %d launches" % (tot_ms, len(rec)))
        for k, a in top[:40]:
            log("    %-46s x%-4d %8.3f ms  %7.2f TFLOP/s  %7.1f GB/s" % (
                k, a["n"], a["ms"], a["flops"] / a["ms"] / 1e9 if a["ms"] else 0,
                a["bytes"] / a["ms"] / 1e6 if a["ms"] else 0))
        by_geom = {}
        for name, cost, ms in rec:
            kk = (cost.get("kernel") or name, name, cost.get("geom"))
            g_ = by_geom.setdefault(kk, [0.0, 0, 0.0])
            g_[0] += ms; g_[1] += 1; g_[2] += cost.get("flops", 0)
        log("[bench] per-layer (kernel, C-ABI entry, (B,Cin,Lin,Cout,K,stride,dil,groups)):")
        for kk, g_ in sorted(by_geom.items(), key=lambda kv: -kv[1][0])[:200]:
            log("    %-40s %-22s %-44s x%-3d %7.3f ms %7.2f TF/s" % (
                kk[0], kk[1], str(kk[2]), g_[1], g_[0], g_[2] / g_[0] / 1e9 if g_[0] else 0))
        # dominant kernel = the kernel TEMPLATE with the largest device time (its instantiations differ
        # only in tile shape / taps / fused epilogue; rocprofv3 lists them separately)
        fam = {}
        for k_, a_ in agg.items():
            f_ = fam.setdefault(k_.split("<")[0], {"ms": 0.0, "n": 0, "flops": 0.0, "bytes": 0.0, "event_ms": 0.0, "pipe_s": 0.0,
                                                   "inst": {}, "products": set()})
            for key in ("ms", "n", "flops", "bytes", "event_ms", "pipe_s"):
                f_[key] += a_[key]
            f_["products"].add(a_["products"])
            f_["inst"][k_] = {"launches": a_["n"], "avg_launch_us": 1e3 * a_["ms"] / a_["n"], "products": a_["products"],
                              "tflops": a_["flops"] / a_["ms"] / 1e9 if a_["ms"] else 0.0}
        k, a = sorted(fam.items(), key=lambda kv: -kv[1]["ms"])[0]
        avg_s = a["ms"] / a["n"] / 1e3
        fl, by = a["flops"] / a["n"], a["bytes"] / a["n"]
        # ceiling of the pipe the kernel runs on, from what its launcher recorded (ms_profile_record.products): fp32 FLOPs executed
        # on the 16-bit matrix pipe with split operands can go no faster than its peak / (partial products per fp32 multiply).
        # A template whose instantiations differ in arithmetic is priced launch by launch (pipe_s = sum flops_i / peak_i).
        products = max(a["products"])
        pipe_pk = a["flops"] / a["pipe_s"] if a["pipe_s"] else F32_PEAK          # flop-weighted ceiling over the launches
        compute_bound = a["pipe_s"] >= a["bytes"] / HBM_PEAK
        if compute_bound:
            roof = {"bound": "mfma", "achieved": fl / avg_s / 1e12, "peak": pipe_pk / 1e12, "unit": "TFLOP/s"}
        else:
            roof = {"bound": "hbm", "achieved": by / avg_s / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s"}
        roof["frac"] = roof["achieved"] / roof["peak"]
        # both legs, whichever one binds
        roof["tflops"] = fl / avg_s / 1e12
        roof["gbps"] = by / avg_s / 1e9
        roof["frac_of_hbm"] = roof["gbps"] / (HBM_PEAK / 1e9)
        roof["products_per_multiply"] = sorted(a["products"])
        roof["pipe"] = PIPE_TEXT[products]
        roof["pipe_peak"] = pipe_pk / 1e12
        roof["frac_of_pipe"] = roof["tflops"] / roof["pipe_peak"]
        roof["fp32_peak"] = F32_PEAK / 1e12                      # SURVEY 8(d)'s fp32 roofline, for continuity
        roof["frac_of_fp32_peak"] = roof["tflops"] / roof["fp32_peak"]
        roof["traffic"] = None
        # HBM bytes per launch of that kernel from the TCC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
        # in separate passes, gfx950 x2 fetch correction: tools/pmc_traffic.sh), quoted only when the file was
        # collected on THIS code (code_version stamp)
        ver = code_version()
        roof["code_version"] = ver
        pmc_file = os.path.join(ROOT, "profiles", PMC_FILE)
        if os.path.exists(pmc_file):
            with open(pmc_file) as fh:
                pmc = json.load(fh)
            if pmc.get("code_version") == ver:
                tb, nd = 0.0, 0
                for name, rec_ in pmc.get("kernels", {}).items():
                    if ("::" + k + "<") in name or ("::" + k + "(") in name or name.startswith(k):
                        tb += rec_["hbm_bytes_per_launch"] * rec_["dispatches"]
                        nd += rec_["dispatches"]
                if nd:
                    roof["traffic"] = tb / nd
                    roof["traffic_source"] = "profiles/%s (%d dispatches, code %s)" % (PMC_FILE, nd, ver)
            else:
                roof["traffic_source"] = "profiles/%s is for code %s, benched code is %s: not quoted" % (
                    PMC_FILE, pmc.get("code_version"), ver)
        roof.update({"kernel": k, "launches_per_DG_pair": a["n"], "avg_launch_us": avg_s * 1e6,
                     "timing": "device begin / end timestamps of every kernel (hipExtLaunchKernelGGL start / stop events -- the "
                               "durations rocprofv3 reports) in an eager, stream-serialised D+G pass; a C-ABI call that launches "
                               "several kernels (split-K finish, pack) is charged their sum",
                     "avg_launch_us_stream_events": 1e3 * a["event_ms"] / a["n"],
                     "event_pair_overhead_us": ev_ms * 1e3,
                     "algorithmic_flops_per_launch": fl, "algorithmic_bytes_per_launch": by,
                     "share_of_kernel_time": a["ms"] / tot_ms,
                     "instantiations": dict(sorted(a["inst"].items(), key=lambda kv: -kv[1]["launches"] * kv[1]["avg_launch_us"])),
                     "families": {kf: {"ms_per_DG_pair": round(v["ms"], 4), "launches": v["n"], "products": sorted(v["products"]),
                                       "tflops": round(v["flops"] / v["ms"] / 1e9, 2) if v["ms"] else 0.0,
                                       "gbps": round(v["bytes"] / v["ms"] / 1e6, 1) if v["ms"] else 0.0}
                                  for kf, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])[:12]}})
        result["roofline"] = roof
        launches = W.d_step_launches(B, args.mels, T) + W.g_step_launches(B, args.mels, T)
        fused = W.fused_d_step_launches(B, args.mels, T) + W.fused_g_step_launches(B, args.mels, T)
        ideal = W.roofline_seconds(launches, HBM_PEAK, F32_PEAK)
        ideal_hbm = W.roofline_seconds(launches, HBM_PEAK, float("inf"))
        ideal_hbm_fused = W.roofline_seconds(fused, HBM_PEAK, float("inf"))
        survey_bytes = W.SURVEY_MB_PER_ELEMENT_PER_CALL * 1e6 * 2 * B
        # the launches that really ran, each on the pipe its launcher recorded
        as_run = sum(max(c.get("bytes", 0) / HBM_PEAK, c.get("flops", 0) / pipe_peak(c.get("products", 0))) for _, c, _ in rec)
        as_run_3 = sum(max(c.get("bytes", 0) / HBM_PEAK, c.get("flops", 0) / (BF16_PEAK / 3 if c.get("products", 0) else F32_PEAK))
                       for _, c, _ in rec)
        run_bytes = sum(c.get("bytes", 0) for _, c, _ in rec)
        run_flops = sum(c.get("flops", 0) for _, c, _ in rec)
        by_products = {}
        for _, c, ms in rec:
            bp = by_products.setdefault(str(c.get("products", 0)), {"launches": 0, "gflop": 0.0, "kernel_ms": 0.0})
            bp["launches"] += 1; bp["gflop"] += c.get("flops", 0) / 1e9; bp["kernel_ms"] += ms
        pair_s = 2 * elapsed / args.steps
        result["step_roofline"] = {
            "ideal_ms_per_DG_pair": ideal * 1e3, "measured_ms_per_DG_pair": pair_s * 1e3,
            "frac": ideal / pair_s,
            "frac_definition": "sum over the layer spec's launches of max(bytes / 8 TB/s, flops / 157.3 TFLOP/s) / measured (SURVEY 8(d); "
                               "above 0.5 it says the work left the fp32 pipe, not that a roof is near)",
            "ideal_ms_as_run": as_run * 1e3, "frac_as_run": as_run / pair_s,
            "frac_as_run_definition": "sum over the launches of the instrumented pair of max(bytes / 8 TB/s, flops / pipe peak), the "
                                      "pipe peak being what the launcher recorded for that kernel: 2500 / 3 (fp16 x 2), 2500 / 6 "
                                      "(bf16 x 3) or 157.3 TFLOP/s (fp32 MFMA / vector); bytes = what that launch has to move",
            "ideal_ms_all_split_kernels_at_3_products": as_run_3 * 1e3, "frac_all_split_kernels_at_3_products": as_run_3 / pair_s,
            "by_products_per_multiply": by_products,
            "ideal_ms_memory_bound": ideal_hbm * 1e3, "frac_memory_bound": ideal_hbm / pair_s,
            "frac_memory_bound_definition": "sum of the LAYER-GRANULAR bytes / 8 TB/s / measured (every conv a launch that reads its "
                                            "input and writes its output: north_star's memory-bound roofline on SURVEY 8(d)'s byte model)",
            "ideal_ms_memory_bound_fused": ideal_hbm_fused * 1e3, "frac_memory_bound_fused": ideal_hbm_fused / pair_s,
            "frac_memory_bound_fused_definition": "the same on the byte model of the program as scheduled (_workload.fused_*: one launch "
                                                  "per atom pass / inference stack, sign words, one [fake; real] discriminator pass)",
            "ideal_ms_memory_bound_survey": survey_bytes / HBM_PEAK * 1e3, "frac_memory_bound_survey": survey_bytes / HBM_PEAK / pair_s,
            "frac_memory_bound_survey_definition": "SURVEY 8(d)'s 135 MB per element per call x 2 B / 8 TB/s / measured",
            "algorithmic_gflop_per_DG_pair": W.totals(launches)["flops"] / 1e9,
            "algorithmic_mb_per_DG_pair": W.totals(launches)["bytes"] / 1e6,
            "fused_mb_per_DG_pair": W.totals(fused)["bytes"] / 1e6,
            "as_run_mb_per_DG_pair": run_bytes / 1e6, "as_run_gflop_per_DG_pair": run_flops / 1e9,
            "launches_per_DG_pair": len(rec), "kernels_per_DG_pair": sum(c.get("kernels", 0) for _, c, _ in rec),
            "kernel_ms_per_DG_pair_eager": tot_ms}
        # whole-pair HBM traffic from the TCC counters over the bytes the launches have to move (same code_version rule as
        # roofline.traffic)
        sr = result["step_roofline"]
        sr["traffic_ratio"] = None
        if os.path.exists(pmc_file) and pmc.get("code_version") == ver and pmc.get("dg_pairs"):
            tot_b = sum(r_["hbm_bytes_per_launch"] * r_["dispatches"] for r_ in pmc.get("kernels", {}).values())
            sr["pmc_hbm_mb_per_DG_pair"] = tot_b / pmc["dg_pairs"] / 1e6
            sr["traffic_ratio"] = sr["pmc_hbm_mb_per_DG_pair"] / sr["as_run_mb_per_DG_pair"]
            sr["traffic_ratio_vs_layer_granular"] = sr["pmc_hbm_mb_per_DG_pair"] / sr["algorithmic_mb_per_DG_pair"]
            sr["traffic_source"] = "profiles/%s (%d D+G pairs, eager, code %s)" % (PMC_FILE, pmc["dg_pairs"], ver)

    if rank == 0 and args.model == "twostage":
        # step roofline of the two-stage pair (no per-kernel leg): layer specs of both stages, _workload.py
        launches = (W.stage1_d_step_launches(B) + W.stage1_g_step_launches(B)
                    + W.d_step_launches(B, args.mels, T) + W.g_step_launches(B, args.mels, T))
        s1 = W.stage1_d_step_launches(B) + W.stage1_g_step_launches(B)
        pair_s = 2 * elapsed / args.steps
        ideal = W.roofline_seconds(launches, HBM_PEAK, F32_PEAK)
        ideal_pipe = W.roofline_seconds(launches, HBM_PEAK, BF16_PEAK / 6)
        ideal_hbm = W.roofline_seconds(launches, HBM_PEAK, float("inf"))
        result["step_roofline"] = {
            "ideal_ms_per_DG_pair": ideal * 1e3, "measured_ms_per_DG_pair": pair_s * 1e3, "frac": ideal / pair_s,
            "frac_definition": "sum over the launches of BOTH stages' D+G calls of max(bytes / 8 TB/s, flops / 157.3 TFLOP/s) "
                               "/ measured (featuresynth/_workload.py: stage1_*_launches + d/g_step_launches at 128 mels)",
            "ideal_ms_split_bf16_pipe": ideal_pipe * 1e3, "frac_split_bf16_pipe": ideal_pipe / pair_s,
            "ideal_ms_memory_bound": ideal_hbm * 1e3, "frac_memory_bound": ideal_hbm / pair_s,
            "stage1_ideal_ms_per_DG_pair": W.roofline_seconds(s1, HBM_PEAK, F32_PEAK) * 1e3,
            "stage1_algorithmic_gflop_per_DG_pair": W.totals(s1)["flops"] / 1e9,
            "stage1_algorithmic_mb_per_DG_pair": W.totals(s1)["bytes"] / 1e6,
            "algorithmic_gflop_per_DG_pair": W.totals(launches)["flops"] / 1e9,
            "algorithmic_mb_per_DG_pair": W.totals(launches)["bytes"] / 1e6}

    if rank == 0 and args.model == "realmelgan":
        # step roofline of the weight-normed variant (no per-kernel leg): layer spec in featuresynth/_workload.py
        launches = W.real_d_step_launches(B, args.mels, T) + W.real_g_step_launches(B, args.mels, T)
        pair_s = 2 * elapsed / args.steps
        ideal = W.roofline_seconds(launches, HBM_PEAK, F32_PEAK)
        ideal_pipe = W.roofline_seconds(launches, HBM_PEAK, BF16_PEAK / 6)
        ideal_hbm = W.roofline_seconds(launches, HBM_PEAK, float("inf"))
        result["step_roofline"] = {
            "ideal_ms_per_DG_pair": ideal * 1e3, "measured_ms_per_DG_pair": pair_s * 1e3, "frac": ideal / pair_s,
            "frac_definition": "sum over the launches of the D+G calls of max(bytes / 8 TB/s, flops / 157.3 TFLOP/s) / measured "
                               "(featuresynth/_workload.py: real_d_step_launches + real_g_step_launches; weight normalisation not priced)",
            "ideal_ms_split_bf16_pipe": ideal_pipe * 1e3, "frac_split_bf16_pipe": ideal_pipe / pair_s,
            "ideal_ms_memory_bound": ideal_hbm * 1e3, "frac_memory_bound": ideal_hbm / pair_s,
            "algorithmic_gflop_per_DG_pair": W.totals(launches)["flops"] / 1e9,
            "algorithmic_mb_per_DG_pair": W.totals(launches)["bytes"] / 1e6}

    if rank == 0 and not args.no_roofline and not args.no_gforward:
        # BASELINE config 2: generator forward only, B=1, 80-bin mel x 32 frames -> 8192 samples
        feat1 = torch.from_numpy(np.random.default_rng(1).standard_normal((1, args.mels, T)).astype(np.float32)).to(device)
        with torch.no_grad():
            for _ in range(5):
                g(feat1)
            torch.cuda.synchronize()
            gg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gg):
                y1 = g(feat1)
            gg.replay(); torch.cuda.synchronize()
            n_it = 200
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n_it):
                gg.replay()
            e1.record(); torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / n_it
        g1 = W.generator_launches(1, args.mels, T, "fwd")
        ideal1 = W.roofline_seconds(g1, HBM_PEAK, F32_PEAK)
        result["generator_forward_b1"] = {"latency_us": us, "samples_per_s": WINDOW / (us * 1e-6),
                                          "ideal_us": ideal1 * 1e6 if ideal1 else None,
                                          "roofline_frac": ideal1 / (us * 1e-6) if ideal1 else None,
                                          "note": "BASELINE config 2 (hipGraph replay of the 30-layer forward); the "
                                                  "ideal is sum over layers of max(bytes / 8 TB/s, flops / 157.3 TFLOP/s): "
                                                  "at B=1 the chain is launch- and latency-bound, not roofline-bound"}
        log("[bench] generator forward B=1: %.1f us -> %.4g samples/s" % (us, WINDOW / (us * 1e-6)))
        del gg, y1

    if rank == 0 and world == 1 and args.model == "melgan" and not args.no_exact:
        # the same timed region with every multiply fp32-exact: what the 22-bit operand scheme buys, stated beside the headline
        r = child_bench(args, EXACT_ENV, "value_exact")
        if "value" in r:
            result["value_exact"] = r["value"]
            result["exact"] = {"value": r["value"], "ms_per_step": r["ms_per_step"], "dtype": r["dtype"], "switches": EXACT_ENV,
                               "headline_over_exact": value / r["value"]}
        else:
            result["value_exact"] = None
            result["exact"] = r

    if rank == 0 and world == 1 and args.model == "melgan" and not args.no_dp_overhead:
        # Multi-GPU readiness measurable on one GPU: the N = 1 step under the data-parallel control flow (three graph
        # segments per call, the stepped net's gradient bucket all-reduced in two slices over a ONE-rank RCCL communicator,
        # both transports) against the plain single-graph step; all three as child processes in the same order on this box.
        plain = child_bench(args, {}, "dp_overhead/plain")
        legs = {}
        for comm in ("torch", "abi"):
            legs[comm] = child_bench(args, {"MSYNTH_DP_FORCE": "1", "MSYNTH_COMM": comm}, "dp_overhead/" + comm)
        dp = {"plain_ms_per_step": plain.get("ms_per_step"), "error": plain.get("error")}
        for comm, r in legs.items():
            dp[comm] = ({"ms_per_step": r["ms_per_step"], "graph_segments": r["config"]["graph_segments"],
                         "overhead_ms_per_step": r["ms_per_step"] - plain["ms_per_step"],
                         "overhead_frac": r["ms_per_step"] / plain["ms_per_step"] - 1.0}
                        if "ms_per_step" in r and "ms_per_step" in plain else r)
        if "ms_per_step" in plain and all("ms_per_step" in r for r in legs.values()):
            best = min(r["ms_per_step"] for r in legs.values())
            # weak scaling efficiency at N = step(1) / step(N); step(N) = dp step + un-hidden all-reduce time
            dp["unhidden_allreduce_budget_ms_per_step_for_90pct"] = plain["ms_per_step"] / 0.9 - best
            dp["note"] = ("one rank: the collectives move no data, so this is the cost of the control flow alone (two more graph "
                          "launches per call, event waits, the RCCL launches); at 8 GPUs the D-step exchanges 22.6 MB and the "
                          "G-step 18.1 MB per call, the early slice under the rest of the backward pass")
        result["dp_overhead"] = dp

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import torch_graph as TG
        cores = host_cpu_share()
        torch.set_num_threads(cores)
        log("[bench] cpu baseline on %d threads (os.cpu_count()=%s) ..." % (cores, os.cpu_count()))
        tr = TG.Trainer(gsd, dsd)
        s_cpu = torch.from_numpy(synthetic_samples(B, WINDOW, rank=0))
        f_cpu = torch.from_numpy(synthetic_features(B, args.mels, T, rank=0))
        c0 = time.perf_counter()
        tr.d_step(s_cpu, f_cpu)                      # warm-up (allocator, MKLDNN primitives)
        log("[bench]   warm-up D-step %.1f s" % (time.perf_counter() - c0))
        npairs = 4                                   # bounded sample: about 10 s of CPU work
        c0 = time.perf_counter()
        for _ in range(npairs):
            tr.d_step(s_cpu, f_cpu)
            tr.g_step(s_cpu, f_cpu)
        cpu_s = time.perf_counter() - c0
        result["cpu_baseline"] = {
            "value": 2 * npairs * B * WINDOW / cpu_s, "unit": "samples/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": "%d alternating D/G trainer calls at B=%d (after 1 warm-up D-step) of the torch-functional "
                      "CPU restatement of the reference graph (oracle/torch_graph.py), %.1f s" % (2 * npairs, B, cpu_s)}
        log("[bench] cpu baseline: %.4g samples/s on %d threads" % (result["cpu_baseline"]["value"],
                                                                    torch.get_num_threads()))

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
