"""Derived ratios from tools/pmc_one.sh listings: MFMA-pipe busy share, vector-ALU share, LDS bank-conflict rate."""
import re, sys
for path in sys.argv[1:]:
    kern, vals = None, {}
    out = []
    for line in open(path):
        if not line.startswith("   ") and line.strip():
            if kern and vals:
                out.append((kern, vals))
            kern, vals = line.strip(), {}
        else:
            m = re.match(r"\s+(\S+)\s+n=(\d+)\s+mean (\S+)", line)
            if m:
                vals[m.group(1)] = float(m.group(3))
    if kern and vals:
        out.append((kern, vals))
    for kern, v in out:
        if "SQ_BUSY_CYCLES" not in v or "SQ_INSTS_MFMA" not in v or v.get("SQ_INSTS_MFMA", 0) == 0:
            continue
        # SQ_BUSY_CYCLES is summed over the 32 shader engines x XCDs sampled; per-SIMD MFMA-busy share = MFMA busy cycles
        # / (kernel cycles x 1024 SIMDs); kernel cycles = SQ_BUSY_CYCLES / 32
        kcycles = v["SQ_BUSY_CYCLES"] / 32.0
        mfma_share = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (kcycles * 1024) if kcycles else 0
        waves = v.get("SQ_WAVES", 1)
        print("%s\n   kernel ~%.0f cycles; MFMA-pipe busy %.1f %% of SIMD cycles; per wave: %.0f MFMA, %.0f other VALU, %.0f SALU, "
              "%.0f LDS, %.0f VMEM-rd, %.0f VMEM-wr instructions; LDS bank-conflict cycles %.1f %% of LDS-active cycles; "
              "waves waiting (any) %.1f %% of wave cycles"
              % (kern, kcycles, 100 * mfma_share, v["SQ_INSTS_MFMA"] / waves,
                 (v.get("SQ_INSTS_VALU", 0) - v["SQ_INSTS_MFMA"]) / waves, v.get("SQ_INSTS_SALU", 0) / waves,
                 v.get("SQ_INSTS_LDS", 0) / waves, v.get("SQ_INSTS_VMEM_RD", 0) / waves, v.get("SQ_INSTS_VMEM_WR", 0) / waves,
                 100 * v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1),
                 100 * v.get("SQ_WAIT_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1)))
