#!/bin/bash
# Counter passes of the default bench workload and their summaries (GPU box):
#   gpurun_out/<tag>_pmc_summary.json (HBM bytes per launch, profiles/pmc_summary.py)
#   gpurun_out/<tag>_mfma_util.json   (MFMA pipe utilisation per kernel, profiles/mfma_util_summary.py)
# Usage: bash tools/pmc_nerf.sh <tag>
set -e
TAG=$1
bash tools/collect_pmc.sh nerf > gpurun_out/${TAG}_pmc_files.txt 2>&1 || true
F=gpurun_out/r2_pmc_nerf_fetch/f_counter_collection.csv
W=gpurun_out/r2_pmc_nerf_write/w_counter_collection.csv
M=gpurun_out/r2_pmc_nerf_mfma/m_counter_collection.csv
F=$(find gpurun_out/r2_pmc_nerf_fetch -name "*counter_collection.csv" -print -quit)
W=$(find gpurun_out/r2_pmc_nerf_write -name "*counter_collection.csv" -print -quit)
M=$(find gpurun_out/r2_pmc_nerf_mfma -name "*counter_collection.csv" -print -quit)
python3 profiles/pmc_summary.py "$F" "$W" gpurun_out/${TAG}_pmc_summary.json > /dev/null
python3 profiles/mfma_util_summary.py "$M" gpurun_out/${TAG}_mfma_util.json > /dev/null
rm -rf gpurun_out/r2_pmc_nerf_fetch gpurun_out/r2_pmc_nerf_write gpurun_out/r2_pmc_nerf_mfma
echo done
