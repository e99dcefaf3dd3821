"""Prints the GPU timeline (us, relative to the last krig_rhs_kernel) of the kernels around one bench step from a
rocprofv3 --kernel-trace CSV: python3 tools/trace_step.py <k_kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rh = [r for r in rows if 'krig_rhs_kernel' in r['Kernel_Name']]
t0 = int(rh[-1]['Start_Timestamp'])
win = [r for r in rows if int(r['End_Timestamp']) > t0 - 3_500_000 and int(r['Start_Timestamp']) < t0 + 6_000_000]
win.sort(key=lambda r: int(r['Start_Timestamp']))
n = 0
for r in win:
    nm = r['Kernel_Name'].replace('void ', '').replace('gss::', '')[:34]
    s = (int(r['Start_Timestamp']) - t0) / 1e3; e = (int(r['End_Timestamp']) - t0) / 1e3
    big = any(k in nm for k in ('quadform', 'rhs', 'cov_pairwise', 'wd_row', 'finish'))
    n += 1
    if big or n % 12 == 0:
        print(f"{s:9.1f} {e:9.1f} q{r.get('Queue_Id')} {nm}")
