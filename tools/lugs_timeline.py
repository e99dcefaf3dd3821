#!/usr/bin/env python3
"""Kernels longer than 60 us of the LAST gss_lugs_create in a rocprofv3 --kernel-trace CSV of tools/lugs_one.py, with
their queue: start, end, duration (us).  usage: tools/lugs_timeline.py <kernel_trace.csv> [min_us]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
lim = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
cp = [i for i, r in enumerate(rows) if "cov_pairwise" in r["Kernel_Name"]]
i0 = cp[-3]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    nm = r["Kernel_Name"].replace("void ", "").replace("gss::", "")[:34]
    st = (int(r["Start_Timestamp"]) - t0) / 1e3
    en = (int(r["End_Timestamp"]) - t0) / 1e3
    if en - st > lim:
        print(f"{st:9.1f} {en:9.1f} {en - st:8.1f} q{r['Queue_Id']} {nm}")
