"""
MFMA-pipe utilisation per kernel from a rocprofv3 counter pass (csv) collected with
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
        --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (elapsed cycles x 1024 SIMDs), elapsed cycles = GRBM_GUI_ACTIVE / 8
(rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES counts 32 cycles per 32x32x16 bf16
MFMA summed over all SIMDs; SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles per wave).
    python profiles/mfma_util_summary.py counter_collection.csv [out.json]
"""
import collections
import csv
import json
import sys

NAMES = {"nerf_fwd_kernel<true": "fwd_save", "nerf_fwd_kernel<false": "fwd_inference", "nerf_bwd_chain_kernel": "bwd_chain",
         "nerf_wgrad_kernel": "bwd_weights"}
disp = collections.defaultdict(dict)
fam_of = {}
for row in csv.DictReader(open(sys.argv[1])):
    for key, fam in NAMES.items():
        if key in row["Kernel_Name"]:
            fam_of[row["Dispatch_Id"]] = fam
            disp[row["Dispatch_Id"]][row["Counter_Name"]] = disp[row["Dispatch_Id"]].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
by = collections.defaultdict(list)
for d, c in disp.items():
    by[fam_of[d]].append(c)
out = {}
for fam, lst in by.items():
    # split coarse / fine launches by elapsed cycles
    mid = (min(c["GRBM_GUI_ACTIVE"] for c in lst) + max(c["GRBM_GUI_ACTIVE"] for c in lst)) / 2
    for lvl, sel in (("coarse", [c for c in lst if c["GRBM_GUI_ACTIVE"] <= mid]), ("fine", [c for c in lst if c["GRBM_GUI_ACTIVE"] > mid])):
        if not sel:
            continue
        n = len(sel)
        cyc = sum(c["GRBM_GUI_ACTIVE"] for c in sel) / n / 8
        mfma = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"] for c in sel) / n
        wave = sum(c.get("SQ_WAVE_CYCLES", 0.0) for c in sel) / n
        out[f"{lvl}_{fam}"] = dict(launches=n, elapsed_cycles=round(cyc), mfma_pipe_utilisation=round(mfma / (cyc * 1024), 4),
                                   valu_active_per_wave_cycle=round(sum(c.get("SQ_ACTIVE_INST_VALU", 0.0) for c in sel) / n / max(wave, 1), 4),
                                   wait_inst_per_wave_cycle=round(sum(c.get("SQ_WAIT_INST_ANY", 0.0) for c in sel) / n / max(wave, 1), 4))
text = json.dumps(out, indent=1)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
