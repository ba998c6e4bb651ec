import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
from featuresynth._ops import prims as P, lib as L
torch.manual_seed(0)
def run(B, C, Lg, K, dil, env):
    for k, v in env.items(): os.environ[k] = v
    x = torch.randn(B, C, Lg, device="cuda"); w = torch.randn(C, C, K, device="cuda") * 0.05; b = torch.randn(C, device="cuda")
    d, lo = P.conv_desc(x.shape, w.shape, pad=dil * (K - 1) // 2, dil=dil, act=1)
    outs = {}
    for mode in ("fp32", "4w", "pair"):
        os.environ["MSYNTH_ROWS3"] = "0" if mode == "fp32" else "1"
        os.environ["MSYNTH_ROWS3P"] = "1" if mode == "pair" else "0"
        name = L.load().ms_conv1d_kernel_name(d, 0).decode()
        y, _ = P.conv1d_fwd(x, w, b, d, lo)
        torch.cuda.synchronize()
        outs[mode] = (y.clone(), name)
    ref = outs["fp32"][0]
    for mode in ("4w", "pair"):
        y, name = outs[mode]
        err = (y - ref).abs()
        # per (batch, 128-col tile) max error
        nt = (Lg + 127) // 128
        per = [[float(err[bb, :, t * 128:(t + 1) * 128].max()) for t in range(nt)] for bb in range(B)]
        bad = [(bb, t) for bb in range(B) for t in range(nt) if per[bb][t] > 1e-3]
        print(env, (B, C, Lg, K, dil), mode, name, "rel", float((y - ref).norm() / ref.norm()), "bad tiles", bad[:40], len(bad), flush=True)
os.environ["MSYNTH_R3P_MIN"] = "1"
run(2, 128, 2048, 3, 1, {"MSYNTH_SPLIT_WGS": "0"})
run(2, 128, 2048, 3, 1, {"MSYNTH_SPLIT_WGS": "192"})
run(32, 128, 2048, 3, 1, {"MSYNTH_SPLIT_WGS": "192"})
run(1, 128, 300, 3, 3, {"MSYNTH_SPLIT_WGS": "0"})
run(2, 64, 1028, 3, 1, {"MSYNTH_SPLIT_WGS": "0"})
