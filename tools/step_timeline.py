#!/usr/bin/env python3
"""GPU timeline (us) around the last-but-one launch of a kernel whose name contains <substring> in a rocprofv3
--kernel-trace CSV: tools/step_timeline.py <kernel_trace.csv> <substring> [before] [after]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sub = sys.argv[2]
before = int(sys.argv[3]) if len(sys.argv) > 3 else 12
after = int(sys.argv[4]) if len(sys.argv) > 4 else 25
hits = [i for i, r in enumerate(rows) if sub in r['Kernel_Name']]
i = hits[-2] if len(hits) > 1 else hits[-1]
t0 = int(rows[i]['Start_Timestamp'])
for r in rows[max(0, i - before):i + after]:
    nm = r['Kernel_Name'].replace('void ', '').replace('gss::', '')[:40]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - t0) / 1e3:9.1f} q{r.get('Queue_Id')} {nm}")
