"""Stage-by-stage comparison of the fused Ref-NeRF trunk kernels (refnerf_fused.hip) with the dense GEMM path (both with
bf16 operands), plus timing.  python tools/refnerf_fused_probe.py [m]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learn-nerf_amd")); sys.path.insert(0, ROOT)
import torch
from learn_nerf.ref_nerf import RefNERFModel

m = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
gen = torch.Generator().manual_seed(0)
x = (torch.rand(m, 3, generator=gen) * 2 - 1).cuda()
d = torch.randn(m, 3, generator=gen); d = (d / d.norm(dim=-1, keepdim=True)).cuda()
g_d = torch.randn(m, generator=gen).cuda(); g_c = torch.randn(m, 3, generator=gen).cuda()
g_a = {"normal_mse": torch.rand(m, generator=gen).cuda(), "neg_normal": torch.rand(m, generator=gen).cuda()}
res = {}
for kind in ("dense", "fused"):
    model = RefNERFModel(spatial_kernel=kind)
    flat = model.flat(model.init(dict(params=1))["params"])
    off = 0
    g2 = torch.Generator().manual_seed(7)
    for fi, fo in model.layer_dims():
        off += fi * fo
        flat[off:off + fo] += (torch.randn(fo, generator=g2) * 0.1).cuda()
        off += fo
    dens, rgb, aux, ctx = model.forward_points(flat, x, d, save=True)
    grad = torch.zeros_like(flat)
    model.backward(ctx, g_d, g_c, g_a, grad)
    torch.cuda.synchronize()
    res[kind] = dict(z=ctx["dir_in"][:, :256].clone(), nraw=ctx["nraw"].clone(), dens=dens.clone(), rgb=rgb.clone(),
                     aux={k: v.clone() for k, v in aux.items()}, grad=grad.clone(), model=model)
    for _ in range(2):
        dens, rgb, aux, ctx = model.forward_points(flat, x, d, save=True); model.backward(ctx, g_d, g_c, g_a, grad)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        dens, rgb, aux, ctx = model.forward_points(flat, x, d, save=True); model.backward(ctx, g_d, g_c, g_a, grad)
    e1.record(); torch.cuda.synchronize()
    print(f"{kind}: {e0.elapsed_time(e1) / 3:.3f} ms per forward+backward of {m} evaluations", flush=True)
a, b = res["dense"], res["fused"]
def rel(u, v): return float((u - v).norm() / v.norm())
print("z         rel", rel(b["z"], a["z"]), "max", float((b["z"] - a["z"]).abs().max()))
print("nraw      rel", rel(b["nraw"], a["nraw"]), "max", float((b["nraw"] - a["nraw"]).abs().max()), "scale", float(a["nraw"].abs().mean()))
print("density   rel", rel(b["dens"], a["dens"]), " rgb max", float((b["rgb"] - a["rgb"]).abs().max()))
for k in a["aux"]:
    print("aux", k, "max", float((b["aux"][k] - a["aux"][k]).abs().max()))
off = 0
for i, (fi, fo) in enumerate(a["model"].layer_dims()):
    for name, n in (("kernel", fi * fo), ("bias", fo)):
        ga, gb = a["grad"][off:off + n], b["grad"][off:off + n]
        print(f"Dense_{i}.{name}: fused vs dense rel {rel(gb, ga):.3e} (|dense| {float(ga.norm()):.3e})")
        off += n
