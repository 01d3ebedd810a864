#!/bin/bash
# GPU timeline of one train step from a rocprofv3 kernel trace: per kernel, mean duration and mean idle gap before it.
# Usage: bash tools/step_gaps.sh <workload> <tag> [extra bench args]   -> gpurun_out/<tag>_step_gaps.txt
set -e
WL=$1; TAG=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/_gt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/_gt -o k -- python3 bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads --no-kernel-timers "$@" > gpurun_out/${TAG}_gt.json 2> gpurun_out/${TAG}_gt.err
F=$(find gpurun_out/_gt -name "*kernel_trace.csv" | head -1)
python3 tools/step_gaps.py "$F" > gpurun_out/${TAG}_step_gaps.txt
rm -rf gpurun_out/_gt
cat gpurun_out/${TAG}_step_gaps.txt
