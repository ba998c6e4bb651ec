"""r05: ablation builds of the fused atom kernel (what does each resource cost un-overlapped?).  Patches a COPY of
csrc/atom_fused.hip, compiles it and links it with the other objects of the normal build into build_ab/libmsynth_<variant>.so
(git-ignored; MSYNTH_LIB selects it).  Results of the variants are WRONG by construction -- timing only.
    python3 tools/scratch/atom_ablate.py build          (here, no GPU needed)
    python3 tools/scratch/atom_ablate.py run            (GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "music-synthesis_amd", "csrc")
OUT = os.path.join(ROOT, "build_ab")

PATCHES = {
    "base": [],
    # the weight stream: only the very first chunk of a workgroup's life is loaded, every MFMA reuses those registers
    "noweights": [("        const int conv = q / NC, chunk = q % NC;\n",
                   "        const int conv = q / NC, chunk = q % NC;\n        if (abl_loaded) return;\n        abl_loaded = true;\n"),
                  ("    u32x4 fa[2][TM][3][NP];\n", "    u32x4 fa[2][TM][3][NP];\n    bool abl_loaded = false;\n")],
    # no stores reach memory (out-of-range offsets: the instructions still issue)
    "nostores": [("const unsigned o_t = (MODE != 0 && inrow && col >= h2 && col < h2 + no) ? (unsigned)o_lane[j] : OOB;",
                  "const unsigned o_t = OOB;"),
                 ("            o_y[j] = (n < no && c0 + n < L) ? (unsigned)o_lane[j] : OOB;\n",
                  "            o_y[j] = (n < no && c0 + n < L) ? (unsigned)o_lane[j] : OOB;\n            oy_st[j] = OOB;\n"),
                 ("        unsigned o_y[TN];\n", "        unsigned o_y[TN], oy_st[TN];\n"),
                 ("                const unsigned oy = o_y[j];\n", "                const unsigned oy = oy_st[j];\n")],
    # no HBM reads of activations: window and residual loads out of range (return zeros)
    "noloads": [("            const unsigned goff = (t >= 0 && t < L) ? (unsigned)(u_goff[r] + 4 * cc0) : OOB;\n#pragma unroll\n            for (int cc = 0; cc < 4; ++cc)\n                rx[r][cc]",
                 "            const unsigned goff = OOB;\n#pragma unroll\n            for (int cc = 0; cc < 4; ++cc)\n                rx[r][cc]"),
                ("                            rsX, o_y[j], base + ((wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2)) * L4, 0));",
                 "                            rsX, OOB, base + ((wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2)) * L4, 0));")],
    # no matrix work: every MFMA triple becomes one cheap vector op that keeps both operands alive
    "nomfma": [("                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[i][j], 0, 0, 0);     // smallest products first\n"
                "                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[i][j], 0, 0, 0);\n"
                "                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i][j], 0, 0, 0);\n",
                "                            acc[i][j][0] += (float)(ah[0] + bl[1]) + (float)(al[2] + bh[3]);\n")],
}


def build():
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(CSRC, "atom_fused.hip")).read()
    objs = [os.path.join(CSRC, "build", f) for f in sorted(os.listdir(os.path.join(CSRC, "build"))) if f.endswith(".o") and f != "atom_fused.o"]
    for name, patches in PATCHES.items():
        s = src
        for a, b in patches:
            assert s.count(a) >= 1, (name, a[:60])
            s = s.replace(a, b)
        p = os.path.join(OUT, "atom_fused_%s.hip" % name)
        open(p, "w").write(s)
        o = os.path.join(OUT, "atom_fused_%s.o" % name)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                               "-I" + CSRC, "-Wno-unused-function", "-Wno-unused-variable", "-fno-gpu-rdc", "-c", p, "-o", o])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, "libmsynth_%s.so" % name), o] + objs + ["-ldl"])
        print("built", name, flush=True)


def worker():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
    import torch
    from featuresynth._ops import prims as P

    def timeit(fn, n=60):
        for _ in range(5): fn()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    torch.manual_seed(0)
    B, dil = 32, 3
    out = []
    for (C, Lg) in [(32, 8192), (64, 4096), (128, 2048), (256, 256)]:
        x = torch.randn(B, C, Lg, device="cuda")
        s = 1.0 / (3 * C) ** 0.5
        w0 = torch.randn(C, C, 3, device="cuda") * s; w1 = torch.randn(C, C, 3, device="cuda") * s
        b0 = torch.randn(C, device="cuda") * 0.1; b1 = torch.randn(C, device="cuda") * 0.1
        img = P.atom_image(C, x.device); P.atom_pack([(w0, w1, img)])
        imgb = P.atom_image(C, x.device); P.atom_pack([(w0, w1, imgb)], backward=True)
        y1, t1, u1, aux = P.atom_fwd(x, img, b0, b1, dil, True, signs=True)
        g = torch.randn_like(x) * 1e-3
        ti = min(timeit(lambda: P.atom_fwd(x, img, b0, b1, dil, False)) for _ in range(3))
        tt = min(timeit(lambda: P.atom_fwd(x, img, b0, b1, dil, True, signs=True)) for _ in range(3))
        tb = min(timeit(lambda: P.atom_bwd_data(g, u1, t1, imgb, dil, t_signs=aux.t_signs)) for _ in range(3))
        out.append("C=%-3d infer %5.1f train %5.1f bwd %5.1f" % (C, ti, tt, tb))
    print("%-10s | %s" % (os.environ["ABL"], " | ".join(out)), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "--worker":
        worker()
    else:
        for name in PATCHES:
            env = dict(os.environ, MSYNTH_LIB=os.path.join(OUT, "libmsynth_%s.so" % name), ABL=name)
            rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--worker"], env=env)
            if rc:
                print("%s rc=%d" % (name, rc), flush=True)
