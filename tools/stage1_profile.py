"""Per-launch listing of one stage-1 D-step + G-step (eager, HIP events on the C-ABI launches) plus wall time incl.
the torch data-movement kernels in between."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import numpy as np, torch
from featuresynth._ops import lib as L
from featuresynth.experiment import TwoDimGeneratorFeatureExperiment
os.environ["MSYNTH_GRAPH"] = os.environ.get("MSYNTH_GRAPH", "0")
torch.manual_seed(0)
dev = torch.device("cuda", 0)
exp = TwoDimGeneratorFeatureExperiment().to(dev)
B = 32
rng = np.random.default_rng(0)
spec = torch.from_numpy((rng.standard_normal((B, 128, 512)) * 0.5).astype(np.float32)).to(dev)
noise = torch.from_numpy(rng.standard_normal((B, 128, 1)).astype(np.float32)).to(dev)
for _ in range(2):
    exp.d_trainer.train(spec, noise); exp.g_trainer.train(spec, noise)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    exp.d_trainer.train(spec, noise); exp.g_trainer.train(spec, noise)
torch.cuda.synchronize()
print("eager wall per D+G pair: %.2f ms" % ((time.perf_counter() - t0) / 3 * 1e3))
L.profile_begin()
exp.d_trainer.train(spec, noise); exp.g_trainer.train(spec, noise)
rec, ev = L.profile_end(calibrate=True)
agg = {}
for name, cost, ms in rec:
    k = (cost.get("kernel") or name, str(cost.get("geom")))
    a = agg.setdefault(k, [0.0, 0, 0.0]); a[0] += ms - ev; a[1] += 1; a[2] += cost.get("flops", 0)
tot = sum(a[0] for a in agg.values())
print("C-ABI kernels: %.2f ms in %d launches" % (tot, len(rec)))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][0])[:28]:
    print("  %-52s %-44s x%-3d %7.3f ms %6.1f TF/s" % (k[0][:52], k[1], a[1], a[0], a[2] / a[0] / 1e9 if a[0] else 0))
