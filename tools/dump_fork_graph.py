"""Topology of the captured RealMelGan train-step hipGraphs, discriminators serial vs on forked streams
(MSYNTH_REAL_FORK): every captured graph is written as DOT (MSYNTH_GRAPH_DOT_DIR -> hipGraphDebugDotPrint through
torch.cuda.CUDAGraph.debug_dump) and summarised: node kinds, roots / sinks, edges, widest antichain estimate.

    python3 tools/dump_fork_graph.py <outdir>          (one process per fork setting, started here)
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(fork, out):
    os.environ["MSYNTH_REAL_FORK"] = fork
    os.environ["MSYNTH_GRAPH_DOT_DIR"] = out
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import featuresynth as fs
    from featuresynth import loss as LS
    from featuresynth._synthetic import synthetic_features, synthetic_samples
    from featuresynth.experiment import realmelgan as R
    from featuresynth.train import DiscriminatorTrainer, GeneratorTrainer
    import test_gpu_realmelgan as T
    g, d, _, _ = T._nets()
    go = fs.FlatAdam(g.parameters(), lr=0.0, betas=(0.5, 0.9)); do = fs.FlatAdam(d.parameters(), lr=0.0, betas=(0.5, 0.9))
    dt = DiscriminatorTrainer(g, go, d, do, LS.mel_gan_disc_loss); gt = GeneratorTrainer(g, go, d, do, R.mel_gan_gen_loss)
    s = torch.from_numpy(synthetic_samples(2, 1024, rank=1)).cuda(); f = torch.from_numpy(synthetic_features(2, 128, 4, rank=1)).cuda()
    for i in range(6):
        r = dt.train(s, f) if i % 2 == 0 else gt.train(s, f)
        torch.cuda.synchronize()
        print("fork=%s call %d %s" % (fork, i, {k: v for k, v in r.items() if k != "fake"}), flush=True)
    print("fork=%s graph status: D %s / G %s" % (fork, dt.graph_status(), gt.graph_status()), flush=True)


def summarise(path):
    txt = open(path).read()
    nodes = {}
    for m in re.finditer(r'^\s*"?([\w.]+)"?\s*\[(.*?)\];', txt, re.M | re.S):
        name, attrs = m.group(1), m.group(2)
        if name in ("graph", "node", "edge"):
            continue
        lab = re.search(r'label\s*=\s*"(.*?)"', attrs, re.S)
        nodes[name] = lab.group(1) if lab else ""
    edges = [(a, b) for a, b in re.findall(r'"?([\w.]+)"?\s*->\s*"?([\w.]+)"?', txt)]
    indeg, outdeg = collections.Counter(), collections.Counter()
    for a, b in edges:
        outdeg[a] += 1; indeg[b] += 1
        nodes.setdefault(a, ""); nodes.setdefault(b, "")
    kinds = collections.Counter()
    for n, lab in nodes.items():
        k = "kernel"
        low = lab.lower()
        for key in ("memcpy", "memset", "event_record", "eventrecord", "event_wait", "eventwait", "empty", "host", "child"):
            if key in low:
                k = key
                break
        kinds[k] += 1
    roots = [n for n in nodes if indeg[n] == 0]
    sinks = [n for n in nodes if outdeg[n] == 0]
    fan_in = sorted(((indeg[n], n) for n in nodes if indeg[n] > 1), reverse=True)[:5]
    fan_out = sorted(((outdeg[n], n) for n in nodes if outdeg[n] > 1), reverse=True)[:5]
    dup = len(edges) - len(set(edges))
    return ("%s: %d nodes %s, %d edges (%d duplicate), %d roots, %d sinks, nodes with fan-in > 1: %d (max %s), fan-out > 1: %d (max %s)"
            % (os.path.basename(path), len(nodes), dict(kinds), len(edges), dup, len(roots), len(sinks),
               sum(1 for n in nodes if indeg[n] > 1), fan_in[:1], sum(1 for n in nodes if outdeg[n] > 1), fan_out[:1]),
            [nodes[r][:60] for r in roots[:6]], [nodes[r][:60] for r in sinks[:6]])


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--worker":
        worker(sys.argv[2], sys.argv[3])
        sys.exit(0)
    out = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/forkdot")
    for fork in ("0", "1"):
        d = os.path.join(out, "fork" + fork)
        os.makedirs(d, exist_ok=True)
        rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--worker", fork, d])
        print("fork=%s worker rc=%d" % (fork, rc), flush=True)
        for fn in sorted(os.listdir(d)):
            if fn.endswith(".dot"):
                line, roots, sinks = summarise(os.path.join(d, fn))
                print("  " + line)
                print("     roots:", roots)
                print("     sinks:", sinks)
