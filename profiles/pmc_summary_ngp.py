"""
HBM traffic of the hash-grid kernels of `bench.py --workload ngp` from two rocprofv3 counter passes
(`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, csv), per train step:
    python profiles/pmc_summary_ngp.py fetch.csv write.csv steps_profiled out.json
bytes = FETCH_SIZE * 1024 * 2 (gfx950 correction) + WRITE_SIZE * 1024, summed over the launches of a kernel and
divided by the number of steps; the bin / reduce / gather kernels run once per model (coarse, fine) and step.
"""
import collections
import csv
import json
import sys

KERNELS = ("hashgrid_fwd_kernel", "hashgrid_bin_kernel", "hashgrid_reduce_kernel", "ngp_mlp_kernel", "ngp_wgrad_kernel")


def total(path, counter):
    out = collections.defaultdict(float)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            for k in KERNELS:
                if k in row["Kernel_Name"]:
                    out[k] += float(row["Counter_Value"])
    return out


steps = float(sys.argv[3])
f, w = total(sys.argv[1], "FETCH_SIZE"), total(sys.argv[2], "WRITE_SIZE")
out = {k: dict(fetch_bytes_per_step=f[k] * 2048 / steps, write_bytes_per_step=w[k] * 1024 / steps,
               hbm_bytes_per_step=(f[k] * 2048 + w[k] * 1024) / steps) for k in KERNELS}
out["fine_hashgrid_bwd"] = dict(hbm_bytes_per_launch=0.75 * (out["hashgrid_bin_kernel"]["hbm_bytes_per_step"] +
                                                             out["hashgrid_reduce_kernel"]["hbm_bytes_per_step"]),
                                note="bin + reduce of the fine model = 786,432 x 16 of the 786,432 x 16 + 262,144 x 6 "
                                     "sample-levels per step (0.89); 0.75 by evaluations is used as the lower bound")
out["_method"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH*1024*2, WRITE*1024; per step"
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out, indent=1))
