"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel table: python rocpd_summary.py x.db [out.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                  "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
for r in rows:
    lines.append(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]:.1f},{r[4]},{r[5]},{100.0 * r[2] / tot:.2f}")
text = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
else:
    sys.stdout.write(text)
