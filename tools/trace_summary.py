#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 --kernel-trace CSV: per kernel name calls / total / mean, plus the idle gaps
between consecutive kernels in the window [t0, t1] of the LAST `marker` occurrences.
usage: tools/trace_summary.py <kernel_trace.csv> [skip_first_fraction]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * frac):]          # second half = the measured (warm) repetition
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
busy = 0
gaps = []
prev_end = None
acc = defaultdict(lambda: [0, 0])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    name = r["Kernel_Name"].split("(")[0][-60:]
    acc[name][0] += 1
    acc[name][1] += e - s
    if prev_end is not None:
        gaps.append(max(0, s - prev_end))
    prev_end = max(prev_end or 0, e)
print("window %.3f ms, kernels %d, busy %.3f ms, sum of gaps %.3f ms (mean gap %.2f us)" %
      ((t1 - t0) / 1e6, len(rows), busy / 1e6, sum(gaps) / 1e6, (sum(gaps) / max(len(gaps), 1)) / 1e3))
for name, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%-62s %5d calls %9.3f ms  mean %8.1f us" % (name, n, t / 1e6, t / n / 1e3))
