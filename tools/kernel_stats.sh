#!/bin/bash
# rocprofv3 kernel-trace statistics of one bench workload on the GPU box -> gpurun_out/<tag>_kernel_stats.csv
# Usage: bash tools/kernel_stats.sh <workload> <tag> [extra bench args]
set -e
WL=$1; TAG=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/_kt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_kt -o k -- python3 bench.py --workload $WL --steps 50 --warmup 5 --no-cpu-baseline --no-other-workloads "$@" > gpurun_out/${TAG}_kt.json 2> gpurun_out/${TAG}_kt.err
find gpurun_out/_kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
rm -rf gpurun_out/_kt
python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/${TAG}_kernel_stats.csv")))
for r in rows[:16]:
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {r['Percentage']}")
PY
