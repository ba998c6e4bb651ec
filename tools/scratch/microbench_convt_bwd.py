"""Transposed-conv backward data: image kernel (csrc/convt_bwd_img.hip) vs the fp32 row-tile path, per layer shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import numpy as np, torch
from featuresynth._ops import prims as P
SHAPES = [("gen 512<-256 s8 L32", 32, 512, 32, 256, 8), ("gen 256<-128 s8 L256", 32, 256, 256, 128, 8),
          ("s1 2048<-512 W4", 128, 2048, 4, 512, 2), ("s1 1024<-256 W8", 256, 1024, 8, 256, 2), ("s1 512<-128 W16", 512, 512, 16, 128, 2),
          ("s1 256<-128 W32", 1024, 256, 32, 128, 2), ("s1 256<-64 W64", 2048, 256, 64, 64, 2), ("s1 192<-32 W128", 4096, 192, 128, 32, 2)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
for name, B, Cin, Lin, Cout, S in SHAPES:
    rng = np.random.default_rng(0)
    w = torch.from_numpy((rng.standard_normal((Cin, Cout, 2 * S)) * 0.05).astype(np.float32)).cuda()
    gy = torch.randn((B, Cout, Lin * S), device="cuda"); y = torch.randn((B, Cout, Lin * S), device="cuda")
    d, _ = P.convt_desc((B, Cin, Lin), w.shape, S, S // 2, act=1)
    flops = 2.0 * B * Lin * Cin * Cout * 2 * S
    os.environ.pop("MSYNTH_CONVTBWDIMG", None)
    t_new = timeit(lambda: P.convt1d_bwd_data(gy, y, w, d)) if P.convt_bwd_img_bytes(d) else float("nan")
    os.environ["MSYNTH_CONVTBWDIMG"] = "0"
    t_old = timeit(lambda: P.convt1d_bwd_data(gy, y, w, d))
    os.environ.pop("MSYNTH_CONVTBWDIMG", None)
    print("%-24s old %7.1f us (%5.1f TF/s)   image %7.1f us (%5.1f TF/s, pack included)" % (name, t_old, flops / t_old / 1e6, t_new, flops / t_new / 1e6))
