"""Per-step GPU time of a bench workload over a long back-to-back run (HIP events between steps):
python tools/step_series.py <workload> <steps>   -> prints the series in blocks of 10 (ms)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learn-nerf_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from learn_nerf.rng import Key

wl, steps = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda", 0)
loop = bench.build_loop(wl, "bf16", 19, dev)
step = loop.step_fn(bench.BBOX_MIN, bench.BBOX_MAX)
batch = bench.synthetic_batch(4096, 1000, dev)
marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
torch.cuda.synchronize()
marks[0].record()
for i in range(steps):
    step(Key(i), batch)
    marks[i + 1].record()
torch.cuda.synchronize()
g = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
for k in range(0, steps, 10):
    print(f"steps {k:4d}..: " + " ".join(f"{v:6.2f}" for v in g[k:k + 10]))
