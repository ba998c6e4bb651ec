"""r04 probe: the fused atom kernel's operand schemes / schedules at the bench shapes (B = 32).
One process per setting (the switches are read once per process):
    MSYNTH_ATOM_NP=3                  bf16 x 3, six products (the r03 kernel)
    MSYNTH_ATOM_NP=2                  block-scaled fp16 x 2, three products (atom_fused.hip); MSYNTH_LIB=<other build> for A/Bs
Prints time per launch and the rel-L2 distance to the two row-tile launches (bf16 x 3, exact products).
    python3 tools/scratch/probe_atom_np.py            (driver)
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
    import torch
    from featuresynth._ops import graph as G, prims as P

    def timeit(fn, n=30):
        for _ in range(4): fn()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    def rel(a, b):
        a, b = a.double(), b.double()
        return float((a - b).norm() / b.norm())

    tag = "NP=%s %s %s" % (os.environ.get("MSYNTH_ATOM_NP", "2"), os.path.basename(os.environ.get("MSYNTH_LIB", "default")),
                           " ".join("%s=%s" % (k[12:], v) for k, v in sorted(os.environ.items()) if k.startswith("MSYNTH_ATOM_") and k != "MSYNTH_ATOM_NP"))
    torch.manual_seed(0)
    B = 32
    shapes = [(32, 8192), (64, 4096), (128, 2048), (256, 256)]
    only = os.environ.get("PROBE_C")
    for (C, Lg) in shapes:
        if only and int(only) != C:
            continue
        for dil in ((1, 3, 9) if not os.environ.get("PROBE_DIL") else (int(os.environ["PROBE_DIL"]),)):
            x = torch.randn(B, C, Lg, device="cuda")
            w0 = torch.randn(C, C, 3, device="cuda") * (1.0 / (3 * C) ** 0.5); w1 = torch.randn(C, C, 3, device="cuda") * (1.0 / (3 * C) ** 0.5)
            b0 = torch.randn(C, device="cuda") * 0.1; b1 = torch.randn(C, device="cuda") * 0.1
            img = P.atom_image(C, x.device); P.atom_pack([(w0, w1, img)])
            y2, rec2 = G.atom_forward(x, w0, b0, w1, b1, dil, True, image=None)
            out = ["%s C=%-3d L=%-4d dil=%d" % (tag, C, Lg, dil)]
            for save in (False, True):
                y, rec = G.atom_forward(x, w0, b0, w1, b1, dil, save, image=img)
                t = timeit(lambda: G.atom_forward(x, w0, b0, w1, b1, dil, save, image=img))
                e = rel(y, y2)
                if save:
                    e = max(e, rel(rec[3], rec2[3]), rel(rec[4], rec2[4]))
                out.append("%s %6.1f us err %.1e" % ("train" if save else "infer", t, e))
            if P.atom_bwd_supported(B, C, Lg, dil):
                d0, d1, _, tt, uu = rec2
                g = torch.randn_like(x) * 1e-6          # gradient-sized magnitudes
                imgb = P.atom_image(C, x.device); P.atom_pack([(w0, w1, imgb)], backward=True)
                gt, gx, _ = P.atom_bwd_data(g, uu, tt, imgb, dil)
                gt2 = P.conv1d_bwd_data(g, uu, w1, d1)
                gx2 = P.conv1d_bwd_data(gt2, tt, w0, d0, gx_add=g)
                t = timeit(lambda: P.atom_bwd_data(g, uu, tt, imgb, dil))
                out.append("bwd %6.1f us err %.1e / %.1e" % (t, rel(gt, gt2), rel(gx - g, gx2 - g)))
            print(" | ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker()
        sys.exit(0)
    for np_ in ("3", "2"):
        env = dict(os.environ, MSYNTH_ATOM_NP=np_)
        rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--worker"], env=env)
        if rc:
            print("setting NP=%s rc=%d" % (np_, rc), flush=True)
