"""r05: ablation builds of the batched atom weight gradient (k_wgrad_rows3, four-wave form): which part of its operand
preparation costs what?  Patches a COPY of csrc/wgrad_rows.hip, links it with the normal build's other objects into
build_ab/libmsynth_w_<variant>.so (git-ignored; MSYNTH_LIB selects it).  Results are WRONG by construction -- timing only.
    python3 tools/scratch/wgrad_ablate.py build      (here)
    python3 tools/scratch/wgrad_ablate.py run        (GPU box: the bench's per-family kernel time of each build)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "music-synthesis_amd", "csrc")
OUT = os.path.join(ROOT, "build_ab")
SPLIT = "w3_split_quad_np<NP>(e, o3);"
# (the upper halves of the fp32 words as fp16 bit patterns: finite for scaled operands, two instructions per pair)
CHEAP = ("{ const unsigned q0 = (__builtin_bit_cast(unsigned, e[0]) >> 16) | (__builtin_bit_cast(unsigned, e[1]) & 0xffff0000u),"
         " q1 = (__builtin_bit_cast(unsigned, e[2]) >> 16) | (__builtin_bit_cast(unsigned, e[3]) & 0xffff0000u);"
         " o3[0] = make_uint2(q0, q1); o3[NP - 1] = make_uint2(q0, q1); }")
PATCHES = {
    "base": [],
    # tap windows taken from the first aligned vector as they are: no v_perm funnel
    "nofunnel": [("                    switch (o & 7) {\n                        case 0: b[pp] = w3_funnel<0>(lo, hi); break;",
                  "                    switch (0) {\n                        case 0: b[pp] = w3_funnel<0>(lo, hi); break;")],
    # every sign bit reads as positive: no decode, no select
    "nosigns": [("                    pos = ((((i & 1) ? pair >> 16 : pair) >> gm_sh) & 1u) != 0;", "                    pos = true;")],
    # operands stored as raw bits: no conversions, no residual
    "nosplit": [(SPLIT, CHEAP)],
    # no operand reads from memory
    "noloads": [("        const bool live = ch < c_end && b < p.B;", "        const bool live = false;")],
}


def build():
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(CSRC, "wgrad_rows.hip")).read()
    objs = [os.path.join(CSRC, "build", f) for f in sorted(os.listdir(os.path.join(CSRC, "build"))) if f.endswith(".o") and f != "wgrad_rows.o"]
    for name, patches in PATCHES.items():
        if len(sys.argv) > 2 and name not in sys.argv[2:]:
            continue
        s = src
        for a, b in patches:
            assert s.count(a) >= 1, (name, a[:60])
            s = s.replace(a, b)
        p = os.path.join(OUT, "wgrad_rows_%s.hip" % name)
        open(p, "w").write(s)
        o = os.path.join(OUT, "wgrad_rows_%s.o" % name)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                               "-I" + CSRC, "-Wno-unused-function", "-Wno-unused-variable", "-fno-gpu-rdc", "-c", p, "-o", o])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, "libmsynth_w_%s.so" % name), o] + objs + ["-ldl"])
        print("built", name, flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        for name in PATCHES:
            env = dict(os.environ, MSYNTH_LIB=os.path.join(OUT, "libmsynth_w_%s.so" % name))
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "6", "--no-cpu-baseline", "--no-exact",
                                "--no-dp-overhead", "--no-gforward"], env=env, capture_output=True, text=True)
            try:
                d = json.loads(r.stdout.strip().splitlines()[-1])
                fam = d["roofline"]["families"].get("k_wgrad_rows3", {})
                print("%-9s k_wgrad_rows3 %.4f ms per D+G pair (3 launches) | step %.4f ms" % (name, fam.get("ms_per_DG_pair", float("nan")), d["ms_per_step"]), flush=True)
            except Exception as e:
                print(name, "failed", e, r.stderr[-300:], flush=True)
