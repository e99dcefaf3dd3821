#!/usr/bin/env python3
"""Summarises rocprofv3 counter_collection CSVs per kernel: mean counter value per dispatch and mean duration.
usage: tools/pmc_summary.py <dir with pass*/k_counter_collection.csv> [kernel substring]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in sorted(glob.glob(os.path.join(d, "pass*", "*counter_collection.csv"))):
    acc = defaultdict(lambda: defaultdict(list))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0][:60]
            if flt in k:
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("==", f)
    for k, cs in acc.items():
        print("  ", k, {c: (round(sum(v) / len(v), 1), len(v)) for c, v in cs.items()})
