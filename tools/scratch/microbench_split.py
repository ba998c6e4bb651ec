"""Row-tile split-K: time the under-filled layers with MSYNTH_SPLIT_WGS = 0 (off) and the default,
and report the max difference between the two results.  GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
from featuresynth._ops import prims as P, lib as L


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): out = fn()
    if isinstance(out, tuple): out = out[0]
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, out


def both(tag, flops, fn, settings=("0", None)):
    res = []
    for v in settings:
        if v is None: os.environ.pop("MSYNTH_SPLIT_WGS", None)
        else: os.environ["MSYNTH_SPLIT_WGS"] = v
        us, out = timeit(fn)
        res.append((v, us, out))
    ref = res[0][2]
    msg = "  ".join("%s: %7.1f us %6.1f TF/s" % ("split<=" + (v or "256"), us, flops / us / 1e6) for v, us, _ in res)
    err = max(float((o - ref).abs().max()) for _, _, o in res[1:])
    print("%-44s %s   maxdiff %.2e (|ref| %.2e)" % (tag, msg, err, float(ref.abs().max())), flush=True)


torch.manual_seed(0)
for B in (64, 32):
    for Lg in (32, 17, 9):
        x = torch.randn(B, 1024, Lg, device="cuda"); w = torch.randn(1024, 1024, 5, device="cuda") * 0.02
        b = torch.randn(1024, device="cuda")
        d, lo = P.conv_desc(x.shape, w.shape, pad=2, act=1)
        fl = 2.0 * B * 1024 * Lg * 1024 * 5
        both("fwd k5 1024 B%d L%d" % (B, Lg), fl, lambda: P.conv1d_fwd(x, w, b, d, lo))
        gy = torch.randn(B, 1024, Lg, device="cuda"); ya = torch.randn(B, 1024, Lg, device="cuda")
        both("bwd_data k5 1024 B%d L%d" % (B, Lg), fl, lambda: P.conv1d_bwd_data(gy, ya, w, d))
for (Cin, Lin, Cout, K, S) in ((512, 32, 256, 16, 8), (256, 256, 128, 16, 8), (128, 2048, 64, 4, 2), (64, 4096, 32, 4, 2)):
    B = 32
    x = torch.randn(B, Cin, Lin, device="cuda"); w = torch.randn(Cin, Cout, K, device="cuda") * 0.02
    b = torch.randn(Cout, device="cuda")
    d, lo = P.convt_desc(x.shape, w.shape, S, S // 2, act=1)
    fl = 2.0 * B * Cin * Cout * K * Lin
    both("convT fwd %s" % ((Cin, Lin, Cout, K, S),), fl, lambda: P.convt1d_fwd(x, w, b, d, lo))
    gy = torch.randn(B, Cout, lo, device="cuda"); ya = torch.randn(B, Cout, lo, device="cuda")
    both("convT bwd_data %s" % ((Cin, Lin, Cout, K, S),), fl, lambda: P.convt1d_bwd_data(gy, ya, w, d))
for (C, Lg, dil) in ((512, 32, 1), (256, 256, 1), (256, 256, 9), (128, 2048, 3), (64, 4096, 1)):
    B = 32
    x = torch.randn(B, C, Lg, device="cuda"); w = torch.randn(C, C, 3, device="cuda") * 0.02; b = torch.randn(C, device="cuda")
    d, lo = P.conv_desc(x.shape, w.shape, pad=dil, dil=dil, act=1)
    fl = 2.0 * B * C * Lg * C * 3
    both("fwd k3 C%d L%d d%d" % (C, Lg, dil), fl, lambda: P.conv1d_fwd(x, w, b, d, lo), settings=("0", None, "512"))
x = torch.randn(32, 80, 32, device="cuda"); w = torch.randn(512, 80, 7, device="cuda") * 0.02; b = torch.randn(512, device="cuda")
d, lo = P.conv_desc(x.shape, w.shape, pad=3, pad_mode=L.PAD_REFLECT, act=1)
both("fwd k7 80->512 L32", 2.0 * 32 * 512 * 32 * 80 * 7, lambda: P.conv1d_fwd(x, w, b, d, lo))
