#!/bin/bash
# A/B of one environment switch on the GPU box.  Usage: bash tools/ab_env.sh VAR workload [steps]
set -e
cd "$GRAFT_REPO_ROOT"
VAR=$1; WL=$2; STEPS=${3:-100}
for val in 1 0 1 0; do
  env $VAR=$val python bench.py --workload $WL --steps $STEPS --warmup 10 --no-cpu-baseline --no-other-workloads \
    > gpurun_out/ab_${VAR}_${val}.json 2> gpurun_out/ab_${VAR}_${val}.err
  python - <<PY
import json
j = json.loads([x for x in open("gpurun_out/ab_${VAR}_${val}.json") if x.startswith("{")][-1])
print("$WL $VAR=$val", round(j["ms_per_step"], 4), {k: v["ms"] for k, v in j.get("kernels", {}).items() if "hashgrid" in k or "$WL" == "nerf"})
PY
done
