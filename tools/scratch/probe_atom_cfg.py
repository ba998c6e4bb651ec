"""(The MSYNTH_ATOM_CFG experiment switch this script drives was removed from csrc/atom_fused.hip once measured: results in
DESIGN.md section 8; to rerun, re-add the instantiations named below to dispatch_atom.)
r05 probe: tile / wave configurations of the fused atom kernel (MSYNTH_ATOM_CFG, read once per process) at the bench
shapes (B = 32), in the three modes the train step runs: inference, training with sign words, backward data with sign words.
    python3 tools/scratch/probe_atom_cfg.py 0 1 2        (driver: one worker process per setting)
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
    import torch
    from featuresynth._ops import prims as P

    def timeit(fn, n=40):
        for _ in range(5): fn()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    def rel(a, b):
        a, b = a.double(), b.double()
        return float((a - b).norm() / b.norm())

    cfg = os.environ.get("MSYNTH_ATOM_CFG", "0")
    torch.manual_seed(0)
    B = 32
    tot = 0.0
    for (C, Lg) in [(32, 8192), (64, 4096), (128, 2048), (256, 256)]:
        for dil in (1, 3, 9):
            x = torch.randn(B, C, Lg, device="cuda")
            s = 1.0 / (3 * C) ** 0.5
            w0 = torch.randn(C, C, 3, device="cuda") * s; w1 = torch.randn(C, C, 3, device="cuda") * s
            b0 = torch.randn(C, device="cuda") * 0.1; b1 = torch.randn(C, device="cuda") * 0.1
            img = P.atom_image(C, x.device); P.atom_pack([(w0, w1, img)])
            imgb = P.atom_image(C, x.device); P.atom_pack([(w0, w1, imgb)], backward=True)
            ref = torch.nn.functional
            t_ref = ref.leaky_relu(ref.conv1d(x, w0, b0, padding=dil, dilation=dil), 0.2)
            y_ref = x + ref.leaky_relu(ref.conv1d(t_ref, w1, b1, padding=1), 0.2)
            y0 = P.atom_fwd(x, img, b0, b1, dil, False)[0]
            y1, t1, u1, aux = P.atom_fwd(x, img, b0, b1, dil, True, signs=True)
            g = torch.randn_like(x) * 1e-3
            gt, gx, _ = P.atom_bwd_data(g, u1, t1, imgb, dil, t_signs=aux.t_signs)
            ti = timeit(lambda: P.atom_fwd(x, img, b0, b1, dil, False))
            tt = timeit(lambda: P.atom_fwd(x, img, b0, b1, dil, True, signs=True))
            tb = timeit(lambda: P.atom_bwd_data(g, u1, t1, imgb, dil, t_signs=aux.t_signs))
            tot += ti + tt + tb
            print("cfg=%s C=%-3d L=%-4d dil=%d | infer %6.1f us err %.1e | train %6.1f us err %.1e / %.1e | bwd %6.1f us |gx|=%.3e"
                  % (cfg, C, Lg, dil, ti, rel(y0, y_ref), tt, rel(y1, y_ref), rel(t1, t_ref), tb, float(gx.double().norm())), flush=True)
    print("cfg=%s sum over the 36 launches: %.1f us" % (cfg, tot), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker()
        sys.exit(0)
    for c in (sys.argv[1:] or ["0"]):
        env = dict(os.environ, MSYNTH_ATOM_CFG=c)
        rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--worker"], env=env)
        if rc:
            print("setting cfg=%s rc=%d" % (c, rc), flush=True)
