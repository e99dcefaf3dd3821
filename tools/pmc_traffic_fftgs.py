#!/usr/bin/env python3
"""HBM traffic of one FFTGS realisation from rocprofv3 PMC passes (tools/pmc_run.sh ... tools/fftgs_one.py), corrected
as MI355X_MICROARCH.md section HBM prescribes: FETCH_SIZE (KiB) x 1024 x 2 on gfx950 (16 B per lane loads counted at
half their bytes), WRITE_SIZE (KiB) x 1024.  Per kernel: total over its dispatches / number of realisations (= the
dispatches of the last pass, ff_x_inv: the strided passes run as several slab launches per realisation).  Writes
the JSON bench.py attaches as fftgs.roofline.traffic.

usage: tools/pmc_traffic_fftgs.py <pmc dir> <out.json>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d, out = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "pass*", "*counter_collection.csv"))):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE") and ("ff_x_" in k or "ff_axis" in k):
                name = k.split("(")[0].replace("void ", "").replace("gss::", "")
                acc[name][row["Counter_Name"]].append((int(row.get("Dispatch_Id", 0)), float(row["Counter_Value"])))
# the spectrum build runs the forward y and z passes once each over the whole buffer (the first two dispatches of the
# mode-0 strided kernel): not part of a realisation
for name, cs in acc.items():
    for cn in cs:
        v = sorted(cs[cn])
        if "ff_axis" in name and "<0," in name:
            v = v[2:]
        cs[cn] = [x for _, x in v]
per = {}
nreal = max(len(cs.get("FETCH_SIZE", [])) for name, cs in acc.items() if name.startswith("ff_x_inv"))
for name, cs in acc.items():
    if name.startswith("ff_x_fwd") and "<2," in name:      # covariance-source instance: spectrum build, not a realisation
        continue
    if "ff_axis" in name and len(cs.get("FETCH_SIZE", [])) < nreal:   # the forward y / z passes of the spectrum build
        continue
    nf = len(cs.get("FETCH_SIZE", []))
    fetch = sum(cs.get("FETCH_SIZE", [])) / nreal * 1024 * 2.0
    write = sum(cs.get("WRITE_SIZE", [])) / nreal * 1024
    per[name] = {"fetch_bytes_corrected": fetch, "write_bytes": write, "dispatches": nf, "per_realisation": nf / nreal}
tot = sum(v["fetch_bytes_corrected"] + v["write_bytes"] for v in per.values())
res = {"source": d, "kernels": per, "hbm_bytes_per_realisation": tot, "algorithmic_bytes": 32.0 * 512 ** 3,
       "realisations": nreal,
       "note": "FETCH_SIZE x1024 x2 (gfx950) + WRITE_SIZE x1024, totals / realisations, summed over the five passes"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
