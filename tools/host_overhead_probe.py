"""How long does the host take to enqueue one train step?  (GPU time per step is ~5 ms; the host must stay ahead.)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learn-nerf_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from learn_nerf.model import NeRFModel  # noqa: E402
from learn_nerf.rng import Key  # noqa: E402
from learn_nerf.train import TrainLoop  # noqa: E402

dev = torch.device("cuda", 0)
loop = TrainLoop(NeRFModel(), NeRFModel(), init_rng=0, lr=1e-4, coarse_ts=64, fine_ts=128, device=dev)
step = loop.step_fn(bench.BBOX_MIN, bench.BBOX_MAX)
batch = bench.synthetic_batch(4096, 1000, dev)
for i in range(5):
    step(Key(i), batch)
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for i in range(n):
    step(Key(10 + i), batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, total {1e3 * (t2 - t0) / n:.3f} ms/step")
