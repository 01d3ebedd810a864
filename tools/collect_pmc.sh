#!/bin/bash
# Counter passes of the bench command (run on the GPU box): separate rocprofv3 --pmc runs, csv output, no tracing domains.
# Usage: bash tools/collect_pmc.sh [workload] -> gpurun_out/r2_pmc_<workload>_{fetch,write,mfma}/
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
WL=${1:-nerf}
ARGS="bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2_pmc_${WL}_fetch -o f -- python3 $ARGS > gpurun_out/r2_pmc_${WL}_fetch.json 2> gpurun_out/r2_pmc_${WL}_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r2_pmc_${WL}_write -o w -- python3 $ARGS > gpurun_out/r2_pmc_${WL}_write.json 2> gpurun_out/r2_pmc_${WL}_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/r2_pmc_${WL}_mfma -o m -- python3 $ARGS > gpurun_out/r2_pmc_${WL}_mfma.json 2> gpurun_out/r2_pmc_${WL}_mfma.err
find gpurun_out/r2_pmc_${WL}_fetch gpurun_out/r2_pmc_${WL}_write gpurun_out/r2_pmc_${WL}_mfma -name "*counter_collection.csv" | head
