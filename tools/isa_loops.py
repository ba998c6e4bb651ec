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
    # LLVM annotates basic blocks: ".LBBn_m:  ; =>This Inner Loop Header: Depth=d" and ".LBBn_k:  ;   in Loop: Header=BBn_m Depth=d"
    blocks, cur = {}, None
    owner = {}
    for l in lines:
        m2 = re.match(r"^(\.LBB\d+_\d+):(.*)$", l)
        if m2:
            cur = m2.group(1)
            blocks[cur] = []
            c = m2.group(2)
            if "Inner Loop Header" in c:
                owner[cur] = cur
            else:
                h = re.search(r"in Loop: Header=(BB\d+_\d+)", c)
                if h:
                    owner[cur] = ".L" + h.group(1)
            continue
        if cur is not None:
            blocks[cur].append(l)
    loops = collections.OrderedDict()
    for blk, hdr in owner.items():
        loops.setdefault(hdr, []).extend(blocks[blk])
    inner = [h for h in loops if h in owner and owner[h] == h]
    if not inner:
        continue
    print(name[:150])
    for hdr in inner:
        c = collections.Counter()
        for l in loops[hdr]:
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
            print("   loop %-10s " % hdr + "  ".join("%s %d" % (k, c[k]) for k in ("mfma", "valu", "v_mov", "accvgpr", "scratch", "lds", "vmem", "salu", "waitcnt", "barrier") if c[k]))
