"""Static look at the hot loops of compiled kernels: python3 tools/isa_loops.py file.s [name filter ...]
(file.s from `hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S csrc/x.hip`).  Per kernel and innermost loop: instruction
counts by class -- MFMAs, other vector ALU, accumulation-file moves (v_accvgpr_*: accumulators shuffled between the two
register files), v_mov, scratch (spills), LDS, vector memory.  Found r05's 48 + 48 accumulator moves per 18 MFMAs in
k_wgrad_rows3 this way."""
import collections, re, sys
src = open(sys.argv[1]).read()
flt = sys.argv[2:]
for m in re.finditer(r"^(_Z\S+):\s*; @\S+\n(.*?)\n\s*s_endpgm", src, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt and not all(f in name for f in flt):
        continue
    lines = body.split("\n")
    heads = [i for i, l in enumerate(lines) if "Inner Loop Header" in l]
    if not heads:
        continue
    print(name[:150])
    for h in heads:
        label = lines[h].split(":")[0].strip()
        end = None
        for j in range(h + 1, len(lines)):
            if re.search(r"s_cbranch\S*\s+%s\b" % re.escape(label), lines[j]) or re.search(r"s_branch\s+%s\b" % re.escape(label), lines[j]):
                end = j
        if end is None:
            continue
        c = collections.Counter()
        for l in lines[h:end + 1]:
            l = l.strip()
            if not l or l.startswith((";", ".")) or l.endswith(":"):
                continue
            op = l.split()[0]
            if op.startswith("v_mfma"): c["mfma"] += 1
            elif op.startswith("v_accvgpr"): c["accvgpr"] += 1
            elif op.startswith("v_mov"): c["v_mov"] += 1
            elif op.startswith("v_"): c["valu"] += 1
            elif op.startswith("scratch"): c["scratch"] += 1
            elif op.startswith("ds_"): c["lds"] += 1
            elif op.startswith(("buffer_", "global_", "flat_")): c["vmem"] += 1
            elif op.startswith("s_waitcnt"): c["waitcnt"] += 1
            elif op.startswith("s_barrier"): c["barrier"] += 1
            elif op.startswith("s_"): c["salu"] += 1
        if c["mfma"] or c["valu"] > 40:
            print("   loop %-10s %4d lines: " % (label, end - h) + "  ".join("%s %d" % (k, c[k]) for k in ("mfma", "valu", "v_mov", "accvgpr", "scratch", "lds", "vmem", "salu", "waitcnt", "barrier") if c[k]))
