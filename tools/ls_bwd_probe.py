"""Timing / first-light probe of the layer-stationary backward (lnrf_nerf_mlp_bwd_ls) against the two-launch backward.
python tools/ls_bwd_probe.py [m]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learn-nerf_amd")); sys.path.insert(0, ROOT)
import torch
from learn_nerf.model import NeRFModel, ls_status

m = int(sys.argv[1]) if len(sys.argv) > 1 else 786432
gen = torch.Generator().manual_seed(0)
x = (torch.rand(m, 3, generator=gen) * 2 - 1).cuda()
d = torch.randn(m, 3, generator=gen); d = (d / d.norm(dim=-1, keepdim=True)).cuda()
gd = torch.randn(m, generator=gen).cuda(); gr = torch.randn(m, 3, generator=gen).cuda()
model = NeRFModel()
flat = model.flat(model.init(dict(params=1))["params"])
grads = {}
for kind in ("split", "ls"):
    model.backward_kernel = kind
    dens, rgb, _, ctx = model.forward_points(flat, x, d, save=True)
    g = torch.zeros_like(flat)
    model.backward(ctx, gd, gr, None, g)
    torch.cuda.synchronize()
    print(kind, "status", ls_status(ctx), "grad norm", float(g.norm()), flush=True)
    grads[kind] = g.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        model.backward(ctx, gd, gr, None, g)
    e1.record(); torch.cuda.synchronize()
    print(f"{kind}: {e0.elapsed_time(e1) / 10:.3f} ms per backward of {m} evaluations, status {ls_status(ctx)}", flush=True)
rel = ((grads["ls"] - grads["split"]).norm() / grads["split"].norm()).item()
print(f"ls vs split gradient: rel L2 {rel:.3e}, max abs {float((grads['ls'] - grads['split']).abs().max()):.3e}")
