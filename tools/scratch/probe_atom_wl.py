"""(The MSYNTH_ATOM_WL variant this script drives -- template flag WL of k_atom_fwd: both weight images copied into LDS in the
prologue, load_a reading them with ds_read_b128 -- was removed from csrc/atom_fused.hip once measured: neutral.  DESIGN.md section 8.)
r05 probe: weights of the 32-channel atoms resident in LDS (MSYNTH_ATOM_WL, read per call) vs streamed from L2, interleaved
in one process; correctness of the resident form against torch."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
import torch.nn.functional as F
from featuresynth._ops import prims as P


def timeit(fn, n=60):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


torch.manual_seed(0)
B, C, Lg = 32, 32, 8192
for dil in (1, 3, 9):
    x = torch.randn(B, C, Lg, device="cuda")
    s = 1.0 / (3 * C) ** 0.5
    w0 = torch.randn(C, C, 3, device="cuda") * s; w1 = torch.randn(C, C, 3, device="cuda") * s
    b0 = torch.randn(C, device="cuda") * 0.1; b1 = torch.randn(C, device="cuda") * 0.1
    img = P.atom_image(C, x.device); P.atom_pack([(w0, w1, img)])
    imgb = P.atom_image(C, x.device); P.atom_pack([(w0, w1, imgb)], backward=True)
    t_ref = F.leaky_relu(F.conv1d(x, w0, b0, padding=dil, dilation=dil), 0.2)
    y_ref = x + F.leaky_relu(F.conv1d(t_ref, w1, b1, padding=1), 0.2)
    g = torch.randn_like(x) * 1e-3
    outs = {}
    for c in ("0", "1"):
        os.environ["MSYNTH_ATOM_WL"] = c
        y0 = P.atom_fwd(x, img, b0, b1, dil, False)[0]
        y1, t1, u1, aux = P.atom_fwd(x, img, b0, b1, dil, True, signs=True)
        gt, gx, _ = P.atom_bwd_data(g, u1, t1, imgb, dil, t_signs=aux.t_signs)
        outs[c] = (y0, y1, t1, gt, gx)
    print("dil=%d resident vs torch: y %.1e t %.1e | resident vs streamed bitwise: %s" % (
        dil, rel(outs["1"][1], y_ref), rel(outs["1"][2], t_ref), all(torch.equal(a, b) for a, b in zip(outs["0"], outs["1"]))), flush=True)
    fns = {"infer": lambda: P.atom_fwd(x, img, b0, b1, dil, False),
           "train": lambda: P.atom_fwd(x, img, b0, b1, dil, True, signs=True),
           "bwd": lambda: P.atom_bwd_data(g, u1, t1, imgb, dil, t_signs=aux.t_signs)}
    for mode, fn in fns.items():
        res = {"0": [], "1": []}
        for rep in range(5):
            for c in ("0", "1"):
                os.environ["MSYNTH_ATOM_WL"] = c
                res[c].append(timeit(fn))
        print("C=32 %-5s dil=%d | streamed median %6.1f us (min %6.1f) | resident median %6.1f us (min %6.1f)" % (
            mode, dil, statistics.median(res["0"]), min(res["0"]), statistics.median(res["1"]), min(res["1"])), flush=True)
os.environ["MSYNTH_ATOM_WL"] = "0"
