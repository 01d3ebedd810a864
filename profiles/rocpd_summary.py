"""
Summarise a rocprofv3 rocpd database (--kernel-trace) into a per-kernel table:
    python profiles/rocpd_summary.py results.db [out.csv]
Kernels that the train step launches twice with different sizes (coarse pass = 64 samples per ray, fine pass
= 192) are bimodal; the last four columns split the launches at the midpoint between the fastest and the
slowest one, so the fine-pass average can be compared with the per-family HIP-event timings of bench.py.
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, end - start from kernels").fetchall()
by_name = {}
for name, dur in rows:
    by_name.setdefault(name, []).append(dur)
total = sum(sum(v) for v in by_name.values())
lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,LowCalls,LowAverageNs,HighCalls,HighAverageNs"]
for name, durs in sorted(by_name.items(), key=lambda kv: -sum(kv[1])):
    mid = (min(durs) + max(durs)) / 2
    low = [d for d in durs if d <= mid]
    high = [d for d in durs if d > mid]
    lines.append(f"\"{name}\",{len(durs)},{sum(durs)},{sum(durs) / len(durs):.1f},{min(durs)},{max(durs)},"
                 f"{100.0 * sum(durs) / total:.2f},{len(low)},{sum(low) / max(len(low), 1):.1f},"
                 f"{len(high)},{sum(high) / max(len(high), 1):.1f}")
text = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
else:
    sys.stdout.write(text)
