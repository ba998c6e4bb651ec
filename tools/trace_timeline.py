"""Timeline analysis of a rocprofv3 kernel trace of the replayed train step (tools/trace_replay.sh writes
gpurun_out/trace9/compact.csv: start, end, queue, kernel name; ns).  Steps are cut at the fused Adam kernel.
Per step kind (D / G): wall, device-busy union, idle, and -- per kernel family -- how long it ran ALONE on the device
(nothing else in flight) vs overlapped: the alone time is what a faster kernel would give back as wall time."""
import collections, re, sys

rows = []
for line in open(sys.argv[1]):
    a = line.rstrip("\n").split(",", 3)
    rows.append((int(a[0]), int(a[1]), a[2], a[3]))
rows.sort()


def fam(name):
    m = re.search(r"(k_[A-Za-z0-9_]+|__amd_rocclr_\w+|\w+elementwise\w*|\w+)", name.replace("void ", "").replace("(anonymous namespace)::", ""))
    return m.group(1) if m else name[:30]


# cut into steps at k_adam
steps, cur = [], []
for r in rows:
    cur.append(r)
    if "k_adam" in r[3]:
        steps.append(cur); cur = []
steps = steps[-8:]          # the last replayed calls
kinds = {}
for i, st in enumerate(steps):
    n = len(st)
    kinds.setdefault(n, []).append(st)
for n, group in sorted(kinds.items()):
    tot_wall = tot_busy = 0.0
    alone = collections.Counter(); overl = collections.Counter(); dur = collections.Counter(); cnt = collections.Counter()
    for st in group:
        t0 = min(r[0] for r in st); t1 = max(r[1] for r in st)
        ev = []
        for k, r in enumerate(st):
            ev.append((r[0], 1, k)); ev.append((r[1], -1, k))
        ev.sort()
        active = set(); last = t0; busy = 0
        for t, d, k in ev:
            if active:
                busy += t - last
                if len(active) == 1:
                    alone[fam(st[next(iter(active))][3])] += t - last
                else:
                    for a in active: overl[fam(st[a][3])] += (t - last) / len(active)
            last = t
            if d == 1: active.add(k)
            else: active.discard(k)
        tot_wall += t1 - t0; tot_busy += busy
        for r in st:
            dur[fam(r[3])] += r[1] - r[0]; cnt[fam(r[3])] += 1
    g = len(group)
    print("== step kind with %d kernels (%d samples): wall %.1f us, busy %.1f us, idle %.1f us, sum of kernels %.1f us" % (
        n, g, tot_wall / g / 1e3, tot_busy / g / 1e3, (tot_wall - tot_busy) / g / 1e3, sum(dur.values()) / g / 1e3))
    for f, v in sorted(alone.items(), key=lambda kv: -kv[1])[:28]:
        print("   %-34s x%-4.0f alone %7.1f us   overlapped-share %7.1f us   total kernel time %7.1f us" % (f, cnt[f] / g, v / g / 1e3, overl[f] / g / 1e3, dur[f] / g / 1e3))
# gaps between steps (host time between two trainer calls)
gaps = []
for a, b in zip(steps, steps[1:]):
    gaps.append(min(r[0] for r in b) - max(r[1] for r in a))
print("gaps between steps (us):", ["%.1f" % (x / 1e3) for x in gaps])
