"""(The MSYNTH_ATOM_CFG experiment switch this script drives was removed from csrc/atom_fused.hip once measured: results in
DESIGN.md section 8; to rerun, re-add the instantiations named below to dispatch_atom.)
r05 probe, second pass: the three candidate changes of probe_atom_cfg.py measured INTERLEAVED in one process (the
switch is read per call), 5 repeats of 60 launches each, medians.
    C = 128 forward (inference / training): 96-column tiles (cfg 3) vs 64 (cfg 0)
    C = 64 training: 8 waves, four per SIMD (cfg 4) vs 4 waves (cfg 0)
    C = 32 backward: 256-column tiles (cfg 3) vs 128 (cfg 0)"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
from featuresynth._ops import prims as P


def timeit(fn, n=60):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


torch.manual_seed(0)
B = 32
if os.environ.get("PROBE_WIDE8"):      # 8-wave workgroups on double-width tiles (cfg 5: <128, 128, 8>, <64, 256, 8>) vs what runs
    cases = [(C, Lg, m, ("0", "5")) for (C, Lg) in ((128, 2048), (64, 4096)) for m in ("infer", "train", "bwd")]
else:
  cases = [(128, 2048, "infer", ("0", "3")), (128, 2048, "train", ("0", "3")), (128, 2048, "bwd", ("0", "3")),
           (64, 4096, "train", ("0", "4")), (64, 4096, "infer", ("0", "4")), (64, 4096, "bwd", ("0", "4")),
           (32, 8192, "bwd", ("0", "3")), (32, 8192, "train", ("0", "3")), (32, 8192, "infer", ("0", "3"))]
for C, Lg, mode, cfgs in cases:
    for dil in (1, 3, 9):
        x = torch.randn(B, C, Lg, device="cuda")
        s = 1.0 / (3 * C) ** 0.5
        w0 = torch.randn(C, C, 3, device="cuda") * s; w1 = torch.randn(C, C, 3, device="cuda") * s
        b0 = torch.randn(C, device="cuda") * 0.1; b1 = torch.randn(C, device="cuda") * 0.1
        img = P.atom_image(C, x.device); P.atom_pack([(w0, w1, img)])
        imgb = P.atom_image(C, x.device); P.atom_pack([(w0, w1, imgb)], backward=True)
        os.environ["MSYNTH_ATOM_CFG"] = "0"
        y1, t1, u1, aux = P.atom_fwd(x, img, b0, b1, dil, True, signs=True)
        g = torch.randn_like(x) * 1e-3
        fn = {"infer": lambda: P.atom_fwd(x, img, b0, b1, dil, False),
              "train": lambda: P.atom_fwd(x, img, b0, b1, dil, True, signs=True),
              "bwd": lambda: P.atom_bwd_data(g, u1, t1, imgb, dil, t_signs=aux.t_signs)}[mode]
        res = {c: [] for c in cfgs}
        for rep in range(5):
            for c in cfgs:
                os.environ["MSYNTH_ATOM_CFG"] = c
                res[c].append(timeit(fn))
        print("C=%-3d %-5s dil=%d | " % (C, mode, dil) + " | ".join("cfg %s median %6.1f us (min %6.1f)" % (c, statistics.median(v), min(v)) for c, v in res.items()), flush=True)
os.environ["MSYNTH_ATOM_CFG"] = "0"
