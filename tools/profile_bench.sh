#!/bin/bash
# rocprofv3 evidence for one bench workload (run on the GPU box): kernel-trace statistics, HBM bytes (FETCH_SIZE / WRITE_SIZE
# in separate counter passes) and MFMA-pipe utilisation, all of `python3 bench.py --workload <wl> ...`.
#   bash tools/profile_bench.sh <workload> <tag> [mlp kernel-name substrings for the time-weighted MFMA figure ...]
# -> gpurun_out/<tag>_kernel_stats.csv, <tag>_bench_under_rocprof.json, <tag>_pmc_summary.json, <tag>_mfma_util.json
set -e
WL=$1; TAG=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/_kt gpurun_out/_pf gpurun_out/_pw gpurun_out/_pm
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_kt -o k -- python3 bench.py --workload $WL --steps 50 --warmup 5 --no-cpu-baseline --no-other-workloads > gpurun_out/${TAG}_bench_under_rocprof.json 2> gpurun_out/${TAG}_kt.err
find gpurun_out/_kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
echo "kernel trace done"
ARGS="bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads --no-kernel-timers"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/_pf -o f -- python3 $ARGS > /dev/null 2> gpurun_out/${TAG}_fetch.err
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/_pw -o w -- python3 $ARGS > /dev/null 2> gpurun_out/${TAG}_write.err
echo "write pass done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/_pm -o m -- python3 $ARGS > /dev/null 2> gpurun_out/${TAG}_mfma.err
echo "mfma pass done"
F=$(find gpurun_out/_pf -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/_pw -name "*counter_collection.csv" | head -1)
M=$(find gpurun_out/_pm -name "*counter_collection.csv" | head -1)
python3 profiles/pmc_summary_generic.py "$F" "$W" 4 gpurun_out/${TAG}_pmc_summary.json
python3 profiles/mfma_util_generic.py "$M" gpurun_out/${TAG}_mfma_util.json "$@" | tail -12
rm -rf gpurun_out/_kt gpurun_out/_pf gpurun_out/_pw gpurun_out/_pm
head -14 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-200
