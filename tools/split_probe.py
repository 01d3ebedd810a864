"""Times the fine-pass NeRFModel forward without activation save: plain bf16 kernel vs the split-precision
(bf16x3) render kernel, 4096 rays x 192 samples.  python tools/split_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learn-nerf_amd")); sys.path.insert(0, ROOT)
import torch
from learn_nerf import ops
from learn_nerf.model import NeRFModel
from bench import synthetic_batch, BBOX_MIN, BBOX_MAX

dev = torch.device("cuda", 0)
batch = synthetic_batch(4096, 1000, dev)
_, _, _, ts = ops.ray_aabb_stratified(batch, BBOX_MIN, BBOX_MAX, 192, seed=1)
for rp in ("bf16", "bf16x3"):
    m = NeRFModel(render_precision=rp)
    flat = m.flat(m.init(dict(params=0))["params"])
    for _ in range(3):
        m.forward_rays(flat, batch, ts, save=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        m.forward_rays(flat, batch, ts, save=False)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{rp}: {ms:.3f} ms per 786,432 evaluations = {786432 * 2 * 591488 / ms / 1e9:.0f} TFLOP/s algorithmic")
