"""Does a backward depend on what its scratch lease held before the call?  Prefills the lease with a byte pattern before every run."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learn-nerf_amd")); sys.path.insert(0, ROOT)
import torch
from learn_nerf.model import NeRFModel
from learn_nerf import _ws, _lib as L

m = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
gen = torch.Generator().manual_seed(11)
x = (torch.rand(m, 3, generator=gen) * 2 - 1).cuda()
d = torch.randn(m, 3, generator=gen); d = (d / d.norm(dim=-1, keepdim=True)).cuda()
gd = torch.randn(m, generator=gen).cuda(); gr = torch.randn(m, 3, generator=gen).cuda()
model = NeRFModel()
flat = model.flat(model.init(dict(params=1))["params"])
shape = model._shape_struct()
sizes = {"split": ("nerf_bwd", L.lib().lnrf_nerf_bwd_scratch_bytes(ctypes.byref(shape), m)),
         "ls": ("nerf_bwd_ls", L.lib().lnrf_nerf_bwd_ls_scratch_bytes(ctypes.byref(shape), m))}
for kind in ("split", "ls"):
    res = []
    for pat in (0, 0, 255, 255, 0x3c, None, None):
        if pat is not None:
            lease = _ws.lease(sizes[kind][0], sizes[kind][1], flat.device)
            lease.buf.fill_(pat)
            lease.release()
        model.backward_kernel = kind
        _, _, _, ctx = model.forward_points(flat, x, d, save=True)
        g = torch.zeros_like(flat)
        model.backward(ctx, gd, gr, None, g)
        torch.cuda.synchronize()
        res.append((pat, g.clone()))
    base = res[0][1]
    print(kind, "prefill pattern -> equal to the first run / NaNs:", [(p, bool(torch.equal(g, base)), int(torch.isnan(g).sum())) for p, g in res])
