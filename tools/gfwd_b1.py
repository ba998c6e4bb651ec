"""BASELINE config 2 (generator forward, B=1): per-launch listing (eager, HIP events) and hipGraph-replay latency."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import numpy as np, torch
import featuresynth as fs
from featuresynth._ops import lib as L
from featuresynth._synthetic import module_param_shapes, synthetic_state_dict
g = fs.MelGanGenerator(32, 80)
g.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(module_param_shapes(g), seed=7).items()})
g.cuda()
x = torch.from_numpy(np.random.default_rng(1).standard_normal((1, 80, 32)).astype(np.float32)).cuda()
with torch.no_grad():
    for _ in range(3): g(x)
    if "--list" in sys.argv:
        L.profile_begin(); g(x); rec, ev = L.profile_end(calibrate=True)
        tot = 0.0
        for name, cost, ms in rec:
            tot += ms - ev
            print("%-22s %-52s %-40s %7.1f us" % (name, cost.get("kernel", ""), str(cost.get("geom")), (ms - ev) * 1e3))
        print("launches %d, sum %.1f us" % (len(rec), tot * 1e3))
    torch.cuda.synchronize()
    gg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gg):
        y = g(x)
    gg.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): gg.replay()
    e1.record(); torch.cuda.synchronize()
    print("env", {k: v for k, v in os.environ.items() if k.startswith("MSYNTH")}, "graph replay: %.1f us per forward" % (1e3 * e0.elapsed_time(e1) / 200))
