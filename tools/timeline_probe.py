"""
In-kernel timeline of the saving forward, the inference forward and the backward chain (debug library built by tools/build_timeline.sh):
    LNRF_LIB=learn-nerf_amd/lib/liblnrf_timeline.so python tools/timeline_probe.py
One workgroup in the middle of the grid stamps s_memtime (100 MHz constant clock on gfx9: REFCLK; converted with the
measured kernel duration) at: kernel start, after the ring prologue, and per 32-row out tile: [stage-barrier
arrive, release,] MFMAs issued, epilogue issued.
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learn-nerf_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from learn_nerf import _lib as L  # noqa: E402
from learn_nerf import ops  # noqa: E402
from learn_nerf.model import NeRFModel  # noqa: E402

dev = torch.device("cuda", 0)
from learn_nerf.rng import Key  # noqa: E402
from learn_nerf.train import TrainLoop  # noqa: E402

lib = L.lib()
lib.lnrf_debug_set_timeline.restype = ctypes.c_int32
lib.lnrf_debug_set_timeline.argtypes = [ctypes.c_void_p]
tl = torch.zeros(3 * 8 * 1024, dtype=torch.int64, device=dev)
loop = TrainLoop(NeRFModel(), NeRFModel(), init_rng=0, lr=1e-4, coarse_ts=64, fine_ts=128, device=dev)
step = loop.step_fn(bench.BBOX_MIN, bench.BBOX_MAX)
batch = bench.synthetic_batch(4096, 1, dev)
for i in range(3):
    step(Key(i), batch)
_, _, _, ts = ops.ray_aabb_stratified(batch, bench.BBOX_MIN, bench.BBOX_MAX, 192, seed=1)
torch.cuda.synchronize()
L.check(lib.lnrf_debug_set_timeline(ctypes.c_void_p(tl.data_ptr())), "set_timeline")
step(Key(9), batch)                                            # saving forward + chain (the fine pass overwrites the coarse one)
loop.fine.forward_rays(loop._slices(loop.flat)[1], batch, ts, save=False)  # inference forward
torch.cuda.synchronize()
t = tl.cpu().view(3, 8, 1024)
for name, k in (("saving forward", 0), ("inference forward", 1), ("backward chain", 2)):
    for w in (0, 4):
        s = t[k, w]
        n = int((s != 0).sum())
        s = s[:n] - s[0]
        d = (s[1:] - s[:-1]).tolist()
        print(f"{name}, wave {w}: {n} stamps, {int(s[-1])} cycles; deltas 200..232: {d[200:232]}")
