"""
Turn two rocprofv3 counter passes (separate runs: `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, csv output) of
`bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads` (tools/collect_pmc.sh) into
profiles/rNN_pmc_summary.json:
    python profiles/pmc_summary.py fetch_counter_collection.csv write_counter_collection.csv out.json
HBM bytes per launch = FETCH_SIZE * 1024 * 2 (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH guide)
+ WRITE_SIZE * 1024.  The train step launches each fused kernel twice per step (coarse pass: 64 samples per ray,
fine pass: 192), coarse first: launches are split by dispatch order.
"""
import collections
import csv
import json
import sys

FAMILIES = {"nerf_fwd_kernel<true": "fwd", "nerf_bwd_chain_kernel": "bwd_chain", "nerf_wgrad_kernel": "bwd_weights"}


def per_launch(path, counter):
    by = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        for key, fam in FAMILIES.items():
            if key in row["Kernel_Name"]:
                by[fam][row["Dispatch_Id"]] += float(row["Counter_Value"])
    # launches in dispatch order: within a step the coarse pass (64 samples per ray) runs before the fine pass (192)
    return {fam: [v[k] for k in sorted(v, key=int)] for fam, v in by.items()}


def split(values):
    """(coarse, fine) means.  The two passes alternate launch by launch; a kernel whose counter does not depend on the
    pass size (the weight-gradient launch writes the same slabs either way) cannot be split by value."""
    low, high = values[0::2], values[1::2]
    return sum(low) / len(low), sum(high) / len(high)


fetch = per_launch(sys.argv[1], "FETCH_SIZE")
write = per_launch(sys.argv[2], "WRITE_SIZE")
out = {}
for fam in ("fwd", "bwd_chain", "bwd_weights"):
    f_lo, f_hi = split(fetch[fam])
    w_lo, w_hi = split(write[fam])
    for lvl, f, w in (("coarse", f_lo, w_lo), ("fine", f_hi, w_hi)):
        out[f"{lvl}_{fam}"] = dict(fetch_bytes_corrected=f * 1024 * 2, write_bytes=w * 1024,
                                   hbm_bytes_per_launch=f * 1024 * 2 + w * 1024)
out["_method"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (no tracing domains); "
                  "FETCH_SIZE*1024*2 (gfx950 correction), WRITE_SIZE*1024; launches split coarse/fine by dispatch order (coarse, fine, coarse, ...); "
                  "bench.py --workload nerf --steps 3 --warmup 1 (tools/collect_pmc.sh)")
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v.get("hbm_bytes_per_launch") for k, v in out.items() if isinstance(v, dict)}, indent=1))
