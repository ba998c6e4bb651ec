"""Kernel sequence of one replayed trainer call (tools/trace_replay.sh writes gpurun_out/trace9/compact.csv): start, duration,
idle gap in front, queue, kernel.  usage: trace_sequence.py compact.csv <kernels per call: 130 = D-step, 170 = G-step>
(rocprofv3 slows the host side of hipGraphLaunch to ~10 us per node: gaps and late starts at the head of a call are
larger than in an untraced run, where the feed rate is ~3.4 us per node -- tools/scratch/probe_launch.py)"""
import sys, re
rows = []
for line in open(sys.argv[1]):
    a = line.rstrip("\n").split(",", 3)
    rows.append((int(a[0]), int(a[1]), a[2], a[3]))
rows.sort()
steps, cur = [], []
for r in rows:
    cur.append(r)
    if "k_adam" in r[3] and "tick" not in r[3]:
        steps.append(cur); cur = []
want = int(sys.argv[2])
st = [s for s in steps[-8:] if len(s) == want][-1]
t0 = st[0][0]
prev_end = t0
def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    return n[:60]
for r in st:
    gap = r[0] - prev_end
    print("%8.1f %7.1f %s q%s %s" % ((r[0]-t0)/1e3, (r[1]-r[0])/1e3, ("gap %5.1f" % (gap/1e3)) if gap > 0 else "         ", r[2], short(r[3])))
    prev_end = max(prev_end, r[1])
