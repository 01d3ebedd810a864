"""
In-kernel timeline of the layer-stationary backward (debug library built by tools/build_timeline.sh):
    LNRF_LIB=learn-nerf_amd/lib/liblnrf_timeline.so python tools/ls_timeline_probe.py [m]
The four waves of pipeline 0's stages 0, 3 and 7 stamp s_memtime (shader clock cycles) at 7 points of 16 consecutive
tiles: 0 iteration start | 1 input known ready | 2 phase A issued (input-gradient MFMAs + poll + DMA pieces) | 3 first half of
phase B issued | 4 phase B issued (weight-gradient MFMAs + epilogue pieces + stores) | 5 counted vmcnt wait passed | 6 barrier passed.
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learn-nerf_amd"))
sys.path.insert(0, ROOT)
from learn_nerf import _lib as L  # noqa: E402
from learn_nerf.model import NeRFModel, ls_status  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 786432
lib = L.lib()
lib.lnrf_debug_set_ls_timeline.restype = ctypes.c_int32
lib.lnrf_debug_set_ls_timeline.argtypes = [ctypes.c_void_p]
gen = torch.Generator().manual_seed(0)
x = (torch.rand(m, 3, generator=gen) * 2 - 1).cuda()
d = torch.randn(m, 3, generator=gen); d = (d / d.norm(dim=-1, keepdim=True)).cuda()
gd = torch.randn(m, generator=gen).cuda(); gr = torch.randn(m, 3, generator=gen).cuda()
model = NeRFModel()
model.backward_kernel = "ls"
flat = model.flat(model.init(dict(params=1))["params"])
dens, rgb, _, ctx = model.forward_points(flat, x, d, save=True)
g = torch.zeros_like(flat)
for _ in range(3):
    model.backward(ctx, gd, gr, None, g)
torch.cuda.synchronize()
tl = torch.zeros(3 * 4 * 16 * 8, dtype=torch.int64, device="cuda")
L.check(lib.lnrf_debug_set_ls_timeline(ctypes.c_void_p(tl.data_ptr())), "set_ls_timeline")
model.backward(ctx, gd, gr, None, g)
torch.cuda.synchronize()
print("status", ls_status(ctx))
t = tl.cpu().view(3, 4, 16, 8)
names = ["ready", "phaseA", "phaseB1", "phaseB2", "vmcnt", "barrier"]
for si, stage in enumerate((0, 3, 7)):
    for w in (0, 3):
        s = t[si, w]
        per = (s[1:, 0] - s[:-1, 0]).float()
        seg = (s[:, 1:7] - s[:, 0:6]).float()
        print(f"stage {stage} wave {w}: tile period mean {per.mean():.0f} min {per.min():.0f} max {per.max():.0f} cycles; "
              + " ".join(f"{n} {seg[:, k].mean():.0f}" for k, n in enumerate(names)))
    print("   periods:", (t[si, 0, 1:, 0] - t[si, 0, :-1, 0]).tolist())
