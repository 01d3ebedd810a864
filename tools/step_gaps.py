"""
Per-step GPU timeline from a rocprofv3 --kernel-trace CSV: the trace is cut into steps at the Adam kernel, and for every
position in the step the kernel name, mean duration and mean idle gap before it (start - previous end on the device,
clamped at 0 for overlapping launches) over the last steps are printed.
    python tools/step_gaps.py k_kernel_trace.csv
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
steps, cur = [], []
for e in ev:
    cur.append(e)
    if "adam_kernel" in e[0]:
        steps.append(cur)
        cur = []
steps = steps[-20:]
lens = sorted(len(s) for s in steps)
n = lens[len(lens) // 2]
steps = [s for s in steps if len(s) == n]
print(f"{len(steps)} steps of {n} launches")
tot_d = tot_g = 0.0
prev_end = None
for i in range(n):
    d = sum(s[i][2] - s[i][1] for s in steps) / len(steps) / 1e3
    g = 0.0
    if i > 0:
        g = sum(max(0, s[i][1] - max(x[2] for x in s[:i])) for s in steps) / len(steps) / 1e3
    tot_d += d
    tot_g += g
    print(f"{i:3d} gap {g:8.1f} us  dur {d:8.1f} us  {steps[0][i][0][:90]}")
span = sum(s[-1][2] - s[0][1] for s in steps) / len(steps) / 1e3
print(f"sum of durations {tot_d:.1f} us, sum of gaps {tot_g:.1f} us, first start -> last end {span:.1f} us")
if len(steps) > 1:
    # gap between steps
    print("(the idle time between one step's Adam kernel and the next step's first kernel is not included)")
