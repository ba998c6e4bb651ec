"""Host cost of hipGraphLaunch for the two step graphs: wraps CUDAGraph.replay with a host clock and runs bench.py."""
import os, runpy, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
_orig = torch.cuda.CUDAGraph.replay
T = []
def timed(self):
    t0 = time.perf_counter(); _orig(self); T.append(time.perf_counter() - t0)
torch.cuda.CUDAGraph.replay = timed
sys.argv = [os.path.join(ROOT, "bench.py"), "--steps", "200", "--warmup", "20", "--no-cpu-baseline", "--no-roofline", "--no-gforward"]
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
except SystemExit:
    pass
import statistics
d = T[-200::2]; g = T[-199::2]
print("host replay() time: even calls median %.1f us, odd calls median %.1f us (n=%d) env %s" % (
    statistics.median(d) * 1e6, statistics.median(g) * 1e6, len(T),
    {k: v for k, v in os.environ.items() if k.startswith("DEBUG_")}), file=sys.stderr)
