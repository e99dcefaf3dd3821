#!/usr/bin/env python3
"""HBM traffic of the dominant kernel from rocprofv3 PMC passes (tools/pmc_krig.sh), corrected as
MI355X_MICROARCH.md section HBM prescribes: FETCH_SIZE (KiB units) under-counts wide coalesced reads by exactly 2x
on gfx950 (16 B/lane loads), WRITE_SIZE is exact.  Writes a small JSON that bench.py attaches as roofline.traffic.

usage: tools/pmc_traffic.py <pmc dir> <kernel substring> <out.json>"""
import csv, glob, json, os, sys
d, sub, out = sys.argv[1], sys.argv[2], sys.argv[3]
tot = {}
cnt = {}
for f in sorted(glob.glob(os.path.join(d, "pass*", "*counter_collection.csv"))):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if sub in row["Kernel_Name"] and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                cnt[row["Counter_Name"]] = cnt.get(row["Counter_Name"], 0) + 1
fetch = tot.get("FETCH_SIZE", 0.0) * 1024 * 2.0      # KiB -> B, gfx950 wide-read correction
write = tot.get("WRITE_SIZE", 0.0) * 1024
res = {"kernel": sub, "source": d, "dispatches": cnt.get("FETCH_SIZE", 0), "fetch_bytes_corrected": fetch,
       "write_bytes": write, "hbm_bytes_per_step": fetch + write,
       "note": "sum over the kernel's dispatches of ONE bench step (1e6 points); FETCH_SIZE x1024 x2 (gfx950), WRITE_SIZE x1024"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
