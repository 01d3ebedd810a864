"""First-light / timing probe of lnrf_nerf_mlp_bwd_fused against the split backward (chain + weight-gradient launches).
python tools/fused_bwd_probe.py [m] [producer_permille] [ring_buffers]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learn-nerf_amd")); sys.path.insert(0, ROOT)
import torch
from learn_nerf import _lib as L
from learn_nerf.model import NeRFModel, fused_bwd_status

m = int(sys.argv[1]) if len(sys.argv) > 1 else 786432
if len(sys.argv) > 3:
    L.check(L.lib().lnrf_nerf_bwd_fused_tune(int(sys.argv[2]), int(sys.argv[3])))
if len(sys.argv) > 4:
    L.check(L.lib().lnrf_nerf_bwd_fused_debug(int(sys.argv[4])))
gen = torch.Generator().manual_seed(0)
x = (torch.rand(m, 3, generator=gen) * 2 - 1).cuda()
d = torch.randn(m, 3, generator=gen); d = (d / d.norm(dim=-1, keepdim=True)).cuda()
gd = torch.randn(m, generator=gen).cuda(); gr = torch.randn(m, 3, generator=gen).cuda()
model = NeRFModel()
flat = model.flat(model.init(dict(params=1))["params"])
grads = {}
for kind in ("split", "fused"):
    model.backward_kernel = kind
    dens, rgb, _, ctx = model.forward_points(flat, x, d, save=True)
    g = torch.zeros_like(flat)
    model.backward(ctx, gd, gr, None, g)
    torch.cuda.synchronize()
    print(kind, "status", fused_bwd_status(), "grad norm", float(g.norm()), flush=True)
    grads[kind] = g.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        model.backward(ctx, gd, gr, None, g)
    e1.record(); torch.cuda.synchronize()
    print(f"{kind}: {e0.elapsed_time(e1) / 5:.3f} ms per backward of {m} evaluations, status {fused_bwd_status()}", flush=True)
rel = ((grads["fused"] - grads["split"]).norm() / grads["split"].norm()).item()
print(f"fused vs split gradient: rel L2 {rel:.3e}, max abs {float((grads['fused'] - grads['split']).abs().max()):.3e}")
