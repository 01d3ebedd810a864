#!/bin/bash
# A/B of the side-stream coarse backward on the GPU box: bench lines for each workload with LNRF_OVERLAP_BACKWARD=1/0.
set -e
cd "$GRAFT_REPO_ROOT"
for wl in ${1:-ngp nerf refnerf}; do
  for ov in 1 0; do
    LNRF_OVERLAP_BACKWARD=$ov python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --no-other-workloads \
      > gpurun_out/ov_${wl}_${ov}.json 2> gpurun_out/ov_${wl}_${ov}.err
    python - <<PY
import json
j = json.loads([x for x in open("gpurun_out/ov_${wl}_${ov}.json") if x.startswith("{")][-1])
print("$wl overlap=$ov", round(j["ms_per_step"], 4), {k: v["ms"] for k, v in j.get("kernels", {}).items()})
PY
  done
done
