#!/usr/bin/env python3
"""Per-kernel HBM traffic from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv output.

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counter unit = KiB;
on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads -> doubled; WRITE_SIZE exact."""
import csv
import glob
import json
import sys
from collections import defaultdict


def load(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                a = acc[row["Kernel_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    return acc


fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(k, [0.0, 0])
    w, nw = write.get(k, [0.0, 0])
    n = max(nf, nw, 1)
    out[k] = {"dispatches": n,
              "fetch_bytes_per_launch": 2.0 * 1024.0 * f / max(nf, 1),     # x2: gfx950 correction
              "write_bytes_per_launch": 1024.0 * w / max(nw, 1)}
    out[k]["hbm_bytes_per_launch"] = out[k]["fetch_bytes_per_launch"] + out[k]["write_bytes_per_launch"]
json.dump(out, sys.stdout, indent=1)
