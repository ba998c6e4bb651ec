"""r05 -> r06: ablation builds of the B = 1 transposed-conv kernel (k_convt_lanes, csrc/small_rows.hip): is its 2.3-2.6 us per
64-channel round the weight walk or the round's own serial chain at one wave per SIMD?
Patches a COPY of csrc/small_rows.hip, links it with the normal build's other objects into build_ab/libmsynth_ct_<variant>.so
(git-ignored; MSYNTH_LIB selects it).  Results are WRONG by construction -- timing only.
    python3 tools/scratch/convt_ablate.py build      (here: cross-compiles)
    python3 tools/scratch/convt_ablate.py run        (GPU box: tools/gfwd_b1.py --list per build, the k_convt_lanes lines)
Reading: `sameweights` ~ base  -> not the walk (every round's 32 KB come from the vector L1 / L2 after the first);
         `nofetch` ~ base      -> not the fetch at all;   `fewfma` ~ base -> not the arithmetic / LDS delivery either (barriers, launch).
`split4` / `wregs` are candidates, not ablations (see their comments)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "music-synthesis_amd", "csrc")
OUT = os.path.join(ROOT, "build_ab")
PATCHES = {
    "base": [],
    # every round reads the weights of input channels 0 .. 63: the same 32 KB per workgroup, no 16 KB-stride walk beyond round one
    "sameweights": [("*reinterpret_cast<const f32x4*>(w + ((size_t)(c0 + c) * p.Cout + co0) * K + part * 4)",
                     "*reinterpret_cast<const f32x4*>(w + ((size_t)c * p.Cout + co0) * K + part * 4)")],
    # only the first round is fetched; later rounds store the same registers again
    "nofetch": [("        if (c0 + CL_CR < p.Cin) fetch(c0 + CL_CR);\n        const int cb = wv * (CL_CR / 4);",
                 "        const int cb = wv * (CL_CR / 4);")],
    # timing of a candidate's main pass: layers of >= 256 input channels split the contraction over FOUR workgroups (grid z = 4 B;
    # all four write the same outputs here -- the real thing writes partial sums and adds an ordered second pass of ~ 1-2 us)
    "split4": [("    const int jt = blockIdx.x, co0 = blockIdx.y * CG, b = blockIdx.z;",
                "    const int NS = p.Cin >= 256 ? 4 : 1;\n    const int jt = blockIdx.x, co0 = blockIdx.y * CG, b = blockIdx.z / NS;\n"
                "    const int cfirst = (blockIdx.z % NS) * (p.Cin / NS), cend = cfirst + p.Cin / NS;"),
               ("    fetch(0);\n    for (int c0 = 0; c0 < p.Cin; c0 += CL_CR) {", "    fetch(cfirst);\n    for (int c0 = cfirst; c0 < cend; c0 += CL_CR) {"),
               ("        if (c0 + CL_CR < p.Cin) fetch(c0 + CL_CR);\n        const int cb", "        if (c0 + CL_CR < cend) fetch(c0 + CL_CR);\n        const int cb"),
               ("    const dim3 grid(ms_ceil_div(q.Lin, CL_JT), ms_ceil_div(q.Cout, 64 / d->stride), q.B);",
                "    const dim3 grid(ms_ceil_div(q.Lin, CL_JT), ms_ceil_div(q.Cout, 64 / d->stride), q.B * (q.Cin >= 256 ? 4 : 1));")],
    # NOT an ablation -- a candidate (same FMAs in the same order: results bitwise those of base): a round's 2 x 16 weights per lane
    # read into registers ahead of the channel loop, the loop fully unrolled, so that no FMA waits for the LAST LDS read issued
    "wregs": [("#pragma unroll 2\n        for (int c = cb; c < cb + CL_CR / 4; ++c) {\n            const float wa = wl[c * WROW + col * K + ka], wb = wl[c * WROW + col * K + kb];",
               "        float wav[CL_CR / 4], wbv[CL_CR / 4];\n#pragma unroll\n        for (int q = 0; q < CL_CR / 4; ++q) { wav[q] = wl[(cb + q) * WROW + col * K + ka]; wbv[q] = wl[(cb + q) * WROW + col * K + kb]; }\n"
               "#pragma unroll\n        for (int c = cb; c < cb + CL_CR / 4; ++c) {\n            const float wa = wav[c - cb], wb = wbv[c - cb];")],
    # one of a wave's sixteen channels per round: a sixteenth of the FMAs and LDS reads, the whole fetch / store / barrier chain
    "fewfma": [("        for (int c = cb; c < cb + CL_CR / 4; ++c) {\n            const float wa = wl[c * WROW + col * K + ka]",
                "        for (int c = cb; c < cb + 1; ++c) {\n            const float wa = wl[c * WROW + col * K + ka]")],
}


def build():
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(CSRC, "small_rows.hip")).read()
    objs = [os.path.join(CSRC, "build", f) for f in sorted(os.listdir(os.path.join(CSRC, "build"))) if f.endswith(".o") and f != "small_rows.o"]
    for name, patches in PATCHES.items():
        if len(sys.argv) > 2 and name not in sys.argv[2:]:
            continue
        s = src
        for a, b in patches:
            assert s.count(a) == 1, (name, a[:60], s.count(a))
            s = s.replace(a, b)
        p = os.path.join(OUT, "small_rows_%s.hip" % name)
        open(p, "w").write(s)
        o = os.path.join(OUT, "small_rows_%s.o" % name)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                               "-I" + CSRC, "-Wno-unused-function", "-Wno-unused-variable", "-fno-gpu-rdc", "-c", p, "-o", o])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, "libmsynth_ct_%s.so" % name), o] + objs + ["-ldl"])
        print("built", name, flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        for name in PATCHES:
            env = dict(os.environ, MSYNTH_LIB=os.path.join(OUT, "libmsynth_ct_%s.so" % name))
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gfwd_b1.py"), "--list"], env=env, capture_output=True, text=True)
            lines = [ln for ln in r.stdout.splitlines() if "k_convt_lanes" in ln or "graph replay" in ln]
            print("== %s" % name, flush=True)
            print("\n".join(lines) if lines else "failed: " + r.stderr[-300:], flush=True)
