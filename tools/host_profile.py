"""
Host-side cost of one train step (cProfile over steps at a small ray count, where the GPU is idle most of the time):
    python tools/host_profile.py [ngp|nerf|refnerf] [rays]
"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learn-nerf_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from learn_nerf.rng import Key  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "ngp"
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda", 0)
loop = bench.build_loop(wl, "bf16", 19, dev)
step = loop.step_fn(bench.BBOX_MIN, bench.BBOX_MAX)
batch = bench.synthetic_batch(rays, 1, dev)
for i in range(10):
    step(Key(i), batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200):
    step(Key(i), batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{wl} rays={rays}: host {1e3 * (t1 - t0) / 200:.3f} ms/step enqueue, {1e3 * (t2 - t0) / 200:.3f} ms/step incl. drain")
pr = cProfile.Profile()
pr.enable()
for i in range(200):
    step(Key(i), batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
