"""
MFMA-pipe utilisation per kernel (generic: every lnrf:: kernel, all launches of the run) from a rocprofv3 counter pass
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv ...
    python profiles/mfma_util_generic.py counter_collection.csv out.json [kernel-name substrings of the "MLP" set ...]
utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (elapsed cycles x 1024 SIMDs), elapsed cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 sums
GRBM_GUI_ACTIVE over the 8 XCDs; the busy counter counts 32 cycles per 32x32x16 bf16 MFMA, summed over all SIMDs).
`_mlp_time_weighted` = sum of busy cycles / (sum of elapsed cycles x 1024) over the kernels whose name contains one of the
given substrings: the time-weighted utilisation of the coarse + fine MLP work of a step.
"""
import collections
import csv
import json
import re
import sys

disp = collections.defaultdict(dict)
name_of = {}
for row in csv.DictReader(open(sys.argv[1])):
    name = re.split(r"[(]", row["Kernel_Name"].replace("void ", ""))[0].strip()
    if not name.startswith("lnrf::"):
        continue
    name_of[row["Dispatch_Id"]] = name
    d = disp[row["Dispatch_Id"]]
    d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
by = collections.defaultdict(list)
for d, c in disp.items():
    by[name_of[d]].append(c)
out = {}
mlp_keys = sys.argv[3:]
mlp_busy = mlp_cyc = 0.0
for name, lst in by.items():
    n = len(lst)
    cyc = sum(c.get("GRBM_GUI_ACTIVE", 0.0) for c in lst) / 8
    busy = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for c in lst)
    wave = sum(c.get("SQ_WAVE_CYCLES", 0.0) for c in lst)
    if cyc <= 0:
        continue
    out[name] = dict(launches=n, elapsed_cycles_per_launch=round(cyc / n), mfma_pipe_utilisation=round(busy / (cyc * 1024), 4),
                     valu_active_per_wave_cycle=round(sum(c.get("SQ_ACTIVE_INST_VALU", 0.0) for c in lst) / max(wave, 1), 4),
                     wait_inst_per_wave_cycle=round(sum(c.get("SQ_WAIT_INST_ANY", 0.0) for c in lst) / max(wave, 1), 4))
    if any(k in name for k in mlp_keys):
        mlp_busy += busy
        mlp_cyc += cyc
out = dict(sorted(out.items(), key=lambda kv: -kv[1]["elapsed_cycles_per_launch"] * kv[1]["launches"]))
if mlp_cyc > 0:
    out["_mlp_time_weighted"] = dict(kernels=mlp_keys, mfma_pipe_utilisation=round(mlp_busy / (mlp_cyc * 1024), 4))
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in out.items():
    print(f"{k[:70]:70s} {v}")
