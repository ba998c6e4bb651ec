"""Device time of the three kernels of the 256-group layer at the bench geometry (B = 64 rows x 3 scales)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import numpy as np, torch
from featuresynth._ops import lib as L, prims as P
rng = np.random.default_rng(0)
def dev(a): return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
B = 64
xs = [dev(rng.standard_normal((B, 1024, l))) for l in (128, 65, 33)]
w = dev(rng.standard_normal((1024, 4, 41)) * 0.08); b = dev(rng.standard_normal(1024) * 0.1)
d, _ = P.conv_desc(xs[0].shape, w.shape, stride=4, pad=20, groups=256, act=L.ACT_LRELU)
ys = P.conv1d_parts_fwd(xs, w, b, d)
gys = [dev(rng.standard_normal(tuple(y.shape))) for y in ys]
def run():
    ys = P.conv1d_parts_fwd(xs, w, b, d)
    gxs = P.conv1d_parts_bwd_data(gys, ys, w, d, [x.shape for x in xs])
    gw, gb = P.conv1d_parts_bwd_weight(xs, gys, ys, d, w.shape)
for _ in range(5): run()
L.profile_begin()
for _ in range(20): run()
rec = L.profile_end()
t = {}
for name, cost, ms in rec: t.setdefault(cost.get("kernel", name), []).append(ms)
print(os.environ.get("MSYNTH_LIB", "default").split("/")[-1], {k: "%.1f us" % (1e3 * float(np.median(v))) for k, v in t.items()})
