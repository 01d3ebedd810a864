"""Throughput of the generic dense kernels (lnrf_dense_fwd / _bwd_input / _bwd_weight) at MLP-layer shapes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learn-nerf_amd"))
from learn_nerf import _lib as L  # noqa: E402
from learn_nerf import ops  # noqa: E402

m, k, n = 786432, 256, 256
x = torch.randn(m, k, device="cuda")
w = torch.randn(k, n, device="cuda") / 16
b = torch.zeros(n, device="cuda")
gy = torch.randn(m, n, device="cuda")
y = torch.empty(m, n, device="cuda")
gx = torch.empty(m, k, device="cuda")
gw = torch.zeros(k, n, device="cuda")


def timed(fn, reps=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


flop = 2.0 * m * k * n
for prec in ("fp32", "bf16"):
    with ops.dense_precision(prec):
        t1 = timed(lambda: ops.dense_fwd(x, w, b, L.ACT_RELU, out=y))
        t2 = timed(lambda: ops.dense_bwd_input(gy, w, out=gx))
        t3 = timed(lambda: ops.dense_bwd_weight(x, gy, gw, None))
    print(f"{prec}: fwd {t1:.3f} ms ({flop / t1 / 1e9:.0f} TF)  dgrad {t2:.3f} ms ({flop / t2 / 1e9:.0f} TF)  "
          f"wgrad {t3:.3f} ms ({flop / t3 / 1e9:.0f} TF)")
