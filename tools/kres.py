"""Kernel resource usage of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
usage: python tools/kres.py music-synthesis_amd/csrc/atom_fused.hip [name filter regex]"""
import os, re, subprocess, sys
src = sys.argv[1]
flt = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(root, "include"),
       "-fno-gpu-rdc", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: (?:[^ ]+ )?\s*(Function Name|VGPRs|AGPRs|VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = subprocess.run(["/usr/bin/c++filt", v], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::", "", cur).split("(")[0].replace("void ", "")
        rows[cur] = {}
    elif cur:
        rows[cur][k] = v
for n, r in rows.items():
    if flt and not flt.search(n):
        continue
    print("%-44s VGPR %4s AGPR %4s spill %3s scratch %4s occ %s" % (n[:44], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"),
                                                                  r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]")))
