"""Inter-step bubble of the synchronous trainer API: events around each graphed step vs the host clock."""
import os, runpy, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
from featuresynth.train import train as T
orig = T._GraphedStep.__call__
REC = []
def call(self, s, f):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record(); out = orig(self, s, f); e1.record(); t1 = time.perf_counter()
    REC.append((t0, t1, e0, e1))
    return out
T._GraphedStep.__call__ = call
sys.argv = [os.path.join(ROOT, "bench.py"), "--steps", "200", "--warmup", "20", "--no-cpu-baseline", "--no-roofline", "--no-gforward"]
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
except SystemExit:
    pass
torch.cuda.synchronize()
R = REC[-200:]
for par, name in ((0, "first kind"), (1, "second kind")):
    span = [R[i][2].elapsed_time(R[i][3]) * 1e3 for i in range(par, len(R) - 1, 2)]
    wall = [(R[i + 1][0] - R[i][0]) * 1e6 for i in range(par, len(R) - 1, 2)]
    host = [(R[i][1] - R[i][0]) * 1e6 for i in range(par, len(R) - 1, 2)]
    gap = [(R[i + 1][0] - R[i][1]) * 1e6 for i in range(par, len(R) - 1, 2)]
    print("%s: device span %.0f us, wall to next call %.0f us (bubble %.0f us), host in launch %.0f us, host between return of launch and next call %.0f us" % (
        name, statistics.median(span), statistics.median(wall), statistics.median(wall) - statistics.median(span), statistics.median(host), statistics.median(gap)), file=sys.stderr)
