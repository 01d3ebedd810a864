"""
In-kernel timeline of the inference forward (debug library built by tools/build_timeline.sh):
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
model = NeRFModel()
flat = model.flat(model.init(dict(params=0))["params"])
batch = bench.synthetic_batch(4096, 1, dev)
_, _, _, ts = ops.ray_aabb_stratified(batch, bench.BBOX_MIN, bench.BBOX_MAX, 192, seed=1)
packed = model.packed_weights(flat)
shape = model._shape_struct()
m = ts.numel()
density = torch.empty(m, device=dev)
rgb = torch.empty(m, 3, device=dev)
tl = torch.zeros(8 * 1024, dtype=torch.int64, device=dev)


def run():
    L.check(L.lib().lnrf_nerf_mlp_fwd(ctypes.byref(shape), L.ptr(packed, torch.uint8), None, None, L.ptr(batch), 9,
                                      L.ptr(ts), ts.shape[1], m, L.ptr(density), L.ptr(rgb),
                                      ctypes.c_void_p(tl.data_ptr()), L.stream()), "fwd")


for _ in range(3):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
print(f"kernel {e0.elapsed_time(e1):.3f} ms")
t = tl.cpu().view(8, 1024)
for w in (0, 4, 3):
    s = t[w]
    n = int((s != 0).sum())
    s = s[:n] - s[0]
    print(f"wave {w}: {n} stamps, total {int(s[-1])} ticks")
    print("  first 40 deltas:", [int(v) for v in (s[1:41] - s[0:40])])
    print("  stamps 200..240 deltas:", [int(v) for v in (s[201:241] - s[200:240])])
