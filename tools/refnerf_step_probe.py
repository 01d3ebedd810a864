"""Train-step time of the Ref-NeRF models (exact-fp32 dense path) at the bench's batch shape."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learn-nerf_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from learn_nerf import _prof  # noqa: E402
from learn_nerf.ref_nerf import RefNERFModel  # noqa: E402
from learn_nerf.rng import Key  # noqa: E402
from learn_nerf.train import TrainLoop  # noqa: E402

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
loop = TrainLoop(RefNERFModel(), RefNERFModel(), init_rng=0, lr=1e-4, coarse_ts=64, fine_ts=128, device=dev)
step = loop.step_fn(bench.BBOX_MIN, bench.BBOX_MAX)
batch = bench.synthetic_batch(n, 1000, dev)
for i in range(2):
    step(Key(i), batch)
torch.cuda.synchronize()
_prof.enable(True)
reps = 5
t0 = time.perf_counter()
for i in range(reps):
    step(Key(10 + i), batch)
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / reps
print(f"ref-nerf step, {n} rays x 192: {ms:.2f} ms/step = {n * 192 / ms * 1e3:.3e} ray-samples/s")
for name, (cnt, t) in sorted(_prof.summary().items(), key=lambda kv: -kv[1][1] * kv[1][0]):
    print(f"  {name:32s} {cnt / reps:5.1f} x {t:8.3f} ms")
