import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
from featuresynth._ops import prims as P, lib as L
torch.manual_seed(0)
def run(B, Cin, Lg, Cout, K, dil):
    x = torch.randn(B, Cin, Lg, device="cuda"); w = torch.randn(Cout, Cin, K, device="cuda") * 0.1; b = torch.randn(Cout, device="cuda")
    d, lo = P.conv_desc(x.shape, w.shape, pad=dil * (K - 1) // 2, dil=dil, act=1)
    gy = torch.randn(B, Cout, lo, device="cuda")
    outs = {}
    for mode in ("0", "1"):
        os.environ["MSYNTH_ROWS3"] = mode
        names = [L.load().ms_conv1d_kernel_name(d, k).decode() for k in (0, 1)]
        y, _ = P.conv1d_fwd(x, w, b, d, lo)
        gx = P.conv1d_bwd_data(gy, y if mode == "0" else outs["0"][0], w, d)
        torch.cuda.synchronize()
        outs[mode] = (y.clone(), gx.clone(), names)
    for i, nm in ((0, "fwd"), (1, "bwd")):
        a, c = outs["0"][i], outs["1"][i]
        print((B, Cin, Lg, Cout, K, dil), nm, outs["0"][2][i], "|", outs["1"][2][i], "rel", float((a - c).norm() / a.norm()),
              "max|ref|", float(a.abs().max()), "max|new|", float(c.abs().max()), flush=True)
run(11, 96, 9, 80, 3, 3)
run(11, 96, 9, 96, 3, 3)
run(11, 96, 9, 80, 3, 1)
run(7, 96, 12, 80, 3, 3)
run(9, 128, 17, 192, 5, 1)
