"""Kernels defined in csrc/*.hip that nothing dispatched in tools/trace_dispatch.sh's traces (the whole GPU test suite + the
three bench models): python3 tools/dead_kernels.py [gpurun_out/dispatch/kernels.txt]"""
import glob, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
seen = set(open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out/dispatch/kernels.txt")).read().split())
for f in sorted(glob.glob(os.path.join(ROOT, "music-synthesis_amd/csrc/*.hip"))):
    src = open(f).read()
    defined = sorted(set(re.findall(r"__global__[^;{]*?\bvoid\s+(k_\w+)\s*\(", src, re.S)))
    dead = [k for k in defined if k not in seen]
    if dead:
        print("%-22s %s" % (os.path.basename(f), " ".join(dead)))
