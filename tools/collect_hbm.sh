#!/bin/bash
# HBM bytes per step of one bench workload (GPU box): FETCH_SIZE and WRITE_SIZE passes only, then the summary.
# Usage: bash tools/collect_hbm.sh <workload> <tag>
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
WL=$1; TAG=$2
ARGS="bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads"
rm -rf gpurun_out/_pf gpurun_out/_pw
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/_pf -o f -- python3 $ARGS > /dev/null 2> gpurun_out/${TAG}_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/_pw -o w -- python3 $ARGS > /dev/null 2> gpurun_out/${TAG}_write.err
F=$(find gpurun_out/_pf -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/_pw -name "*counter_collection.csv" | head -1)
python3 profiles/pmc_summary_generic.py "$F" "$W" 4 gpurun_out/${TAG}_hbm.json
rm -rf gpurun_out/_pf gpurun_out/_pw
