"""Is k_g4_wgrad deterministic (a) alone, (b) beside other work on a second stream?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import numpy as np, torch
from featuresynth._ops import lib as L, prims as P
rng = np.random.default_rng(0)
def dev(a): return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
B = 64
lens = (128, 65, 33)
xs = [dev(rng.standard_normal((B, 1024, l))) for l in lens]
w = dev(rng.standard_normal((1024, 4, 41)) * 0.08)
b = dev(rng.standard_normal(1024) * 0.1)
d, _ = P.conv_desc(xs[0].shape, w.shape, stride=4, pad=20, groups=256, act=L.ACT_LRELU)
ys = P.conv1d_parts_fwd(xs, w, b, d)
gys = [dev(rng.standard_normal(tuple(y.shape))) for y in ys]
torch.cuda.synchronize()
ref = None
for mode in ("alone", "beside"):
    side = torch.cuda.Stream()
    junk = torch.randn(4096, 4096, device="cuda")
    nbad = 0
    for it in range(12):
        if mode == "beside":
            with torch.cuda.stream(side):
                for _ in range(3): junk2 = junk @ junk
                gx_side = P.conv1d_parts_bwd_data(gys, ys, w, d, [x.shape for x in xs])
        gw, gb = P.conv1d_parts_bwd_weight(xs, gys, ys, d, w.shape)
        torch.cuda.synchronize()
        if ref is None: ref = (gw.clone(), gb.clone())
        if not (torch.equal(gw, ref[0]) and torch.equal(gb, ref[1])):
            nbad += 1
            dif = (gw - ref[0]).abs()
            idx = torch.nonzero(dif.flatten() > 0).flatten()
            print(mode, it, "differs: n =", int(idx.numel()), "max", float(dif.max()), "first idx", idx[:8].tolist())
    print(mode, "runs differing from the first:", nbad)
