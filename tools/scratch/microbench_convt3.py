"""ConvTranspose1d forward of the generator's four upsampling layers: paired split-bf16 kernel (conv_rows3.hip, two-tap
form) vs the fp32-MFMA row kernel (MSYNTH_CONVT3=0), with and without the LeakyReLU in front, checked against float64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
import torch.nn.functional as F
from featuresynth._ops import prims as P, lib as L

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, (out[0] if isinstance(out, tuple) else out)

torch.manual_seed(0)
tot = [0.0, 0.0]
for (B, Cin, Lin, Cout, K, S) in ((32, 512, 32, 256, 16, 8), (32, 256, 256, 128, 16, 8), (32, 128, 2048, 64, 4, 2), (32, 64, 4096, 32, 4, 2),
                                  (3, 512, 36, 256, 16, 8), (5, 128, 132, 64, 4, 2), (1, 256, 260, 128, 16, 8)):
    x = torch.randn(B, Cin, Lin, device="cuda"); w = torch.randn(Cin, Cout, K, device="cuda") * 0.02
    bias = torch.randn(Cout, device="cuda")
    for ia in (1, 0):
        d, lo = P.convt_desc(x.shape, w.shape, S, S // 2, act=0, in_act=ia)
        xin = F.leaky_relu(x.double(), 0.2) if ia else x.double()
        ref = F.conv_transpose1d(xin, w.double(), bias.double(), stride=S, padding=S // 2)
        fl = 2.0 * B * Cin * Cout * K * Lin
        res = []
        for mode in ("0", "1"):
            os.environ["MSYNTH_CONVT3"] = mode
            us, out = timeit(lambda: P.convt1d_fwd(x, w, bias, d, lo))
            res.append((us, float((out.double() - ref).norm() / ref.norm()), L.load().ms_convt1d_kernel_name(d, 0).decode()))
        print("%-30s in_act %d fp32 %6.1f us %5.1f TF err %.1e | split %6.1f us %5.1f TF x%.2f err %.1e [%s]" % (
            (B, Cin, Lin, Cout, K, S), ia, res[0][0], fl / res[0][0] / 1e6, res[0][1], res[1][0], fl / res[1][0] / 1e6,
            res[0][0] / res[1][0], res[1][1], res[1][2]), flush=True)
        if B == 32 and ia: tot[0] += res[0][0]; tot[1] += res[1][0]
print("totals us (B = 32, LeakyReLU in front): fp32 %.0f split-bf16 %.0f" % tuple(tot))
