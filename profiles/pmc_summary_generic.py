"""
HBM bytes per step and per kernel from two rocprofv3 counter passes (separate runs: --pmc FETCH_SIZE, --pmc WRITE_SIZE,
csv output; tools/collect_pmc.sh) of a bench.py run with S timed + warm-up steps:
    python profiles/pmc_summary_generic.py fetch.csv write.csv steps out.json
bytes = FETCH_SIZE * 1024 * 2 (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md) + WRITE_SIZE * 1024.
Kernel names are cut at the first '(' / '<'; per kernel: launches per step, mean bytes per launch, bytes per step.
"""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    by = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            name = re.split(r"[(<]", row["Kernel_Name"].replace("void ", ""))[0].strip()
            by[name][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return by


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3])
out, total = {}, 0.0
for name in sorted(set(fetch) | set(write)):
    f = sum(fetch.get(name, {}).values()) * 1024 * 2
    w = sum(write.get(name, {}).values()) * 1024
    n = max(len(fetch.get(name, {})), len(write.get(name, {})))
    if not name.startswith("lnrf::") or (f + w) / steps < 1e6:
        continue
    out[name] = dict(launches_per_step=round(n / steps, 2), fetch_bytes_per_step=f / steps, write_bytes_per_step=w / steps,
                     hbm_bytes_per_step=(f + w) / steps, hbm_bytes_per_launch=(f + w) / max(n, 1))
    # a kernel launched k times per step (e.g. coarse pass, then fine pass): mean bytes of the i-th launch of a step
    k = n // steps
    if k >= 2 and k * steps == n:
        fd = [v for _, v in sorted(fetch.get(name, {}).items(), key=lambda kv: int(kv[0]))]
        wd = [v for _, v in sorted(write.get(name, {}).items(), key=lambda kv: int(kv[0]))]
        if len(fd) == n and len(wd) == n:
            out[name]["hbm_bytes_by_launch_in_step"] = [
                sum(fd[i::k]) / steps * 1024 * 2 + sum(wd[i::k]) / steps * 1024 for i in range(k)]
    total += (f + w) / steps
out = dict(sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_step"]))
out["_total_hbm_bytes_per_step"] = total
out["_method"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (no tracing domains); FETCH x 1024 x 2 "
                  "(gfx950 correction), WRITE x 1024; all launches of the run divided by its step count (warm-up included)")
json.dump(out, open(sys.argv[4], "w"), indent=1)
for k, v in out.items():
    if isinstance(v, dict):
        print(f"{k[:48]:48s} {v['launches_per_step']:5.1f}/step {v['hbm_bytes_per_step'] / 1e9:7.3f} GB/step")
print(f"total {total / 1e9:.2f} GB/step")
