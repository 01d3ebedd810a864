"""Explore the resolution of tests/test_gpu_psnr_parity.py: per-seed bf16 - fp32 held-out PSNR differences for a schedule.
python tools/psnr_explore.py <n_seeds> <tc> <tf> <batch> <steps1,steps2,steps3>"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learn-nerf_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_psnr_parity as T

n_seeds, tc, tf, batch = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
steps = [int(v) for v in sys.argv[5].split(",")]
T.TC, T.TF, T.BATCH = tc, tf, batch
T.SCHEDULE = tuple(zip(steps, (5e-4, 1e-4, 2e-5)))
T.EVAL_EVERY = max(1, sum(steps[1:]) // 16)
T.EVAL_POINTS = 8
train_rays = torch.cat(T.cube_views(24, seed=0), dim=0)
test_views = T.cube_views(8, seed=1234)
deltas = []
t0 = time.time()
for s in range(n_seeds):
    a, _ = T.train("bf16", train_rays, test_views, init_seed=100 + s)
    ta = time.time()
    b, _ = T.train("fp32", train_rays, test_views, init_seed=100 + s)
    deltas.append(a - b)
    print(f"seed {s}: bf16 {a:.3f} fp32 {b:.3f} delta {a - b:+.3f}  ({time.time() - t0:.0f} s so far)", flush=True)
n = len(deltas)
mean = sum(deltas) / n
std = math.sqrt(sum((d - mean) ** 2 for d in deltas) / (n - 1))
print(f"config tc={tc} tf={tf} batch={batch} steps={steps}: mean {mean:+.3f} std {std:.3f} se {std / math.sqrt(n):.3f} over {n} seeds, {time.time() - t0:.0f} s")
