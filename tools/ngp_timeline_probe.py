"""
In-kernel timeline of the fused InstantNGP MLP backward (debug library built by tools/build_timeline.sh):
    LNRF_LIB=learn-nerf_amd/lib/liblnrf_timeline.so python tools/ngp_timeline_probe.py
The middle workgroup stamps s_memtime (100 MHz constant clock) during its third group of 256 evaluations; stamp
order: group start, top barrier, encoding loaded, d / gradients requested, barrier, ring prologue, then per chain layer
and out tile [stage-barrier arrive, release,] MFMAs issued, epilogue issued, and per weight-gradient layer: staged,
released, MFMAs issued, released.
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
from learn_nerf.rng import Key  # noqa: E402

dev = torch.device("cuda", 0)
lib = L.lib()
lib.lnrf_debug_set_ngp_timeline.restype = ctypes.c_int32
lib.lnrf_debug_set_ngp_timeline.argtypes = [ctypes.c_void_p]
tl = torch.zeros(8 * 1024, dtype=torch.int64, device=dev)
loop = bench.build_loop("ngp", "bf16", 19, dev)
step = loop.step_fn(bench.BBOX_MIN, bench.BBOX_MAX)
batch = bench.synthetic_batch(4096, 1, dev)
for i in range(3):
    step(Key(i), batch)
torch.cuda.synchronize()
L.check(lib.lnrf_debug_set_ngp_timeline(ctypes.c_void_p(tl.data_ptr())), "set_ngp_timeline")
step(Key(9), batch)  # the fine model's backward runs first; the coarse one overwrites it: keep only what the last launch wrote
torch.cuda.synchronize()
t = tl.cpu().view(8, 1024)
for w in (0, 3, 4, 7):
    s = t[w]
    n = int((s != 0).sum())
    s = s[:n] - s[0]
    print(f"wave {w}: {n} stamps, {int(s[-1])} ticks of 10 ns; cumulative: {s.tolist()}")
