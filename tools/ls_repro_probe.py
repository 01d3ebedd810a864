"""Runs the layer-stationary backward several times on the same inputs and reports, per Dense layer, where runs differ.
python tools/ls_repro_probe.py [m] [runs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learn-nerf_amd")); sys.path.insert(0, ROOT)
import torch
from learn_nerf.model import NeRFModel, ls_status

m = int(sys.argv[1]) if len(sys.argv) > 1 else 70000
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
gen = torch.Generator().manual_seed(11)
x = (torch.rand(m, 3, generator=gen) * 2 - 1).cuda()
d = torch.randn(m, 3, generator=gen); d = (d / d.norm(dim=-1, keepdim=True)).cuda()
gd = torch.randn(m, generator=gen).cuda(); gr = torch.randn(m, 3, generator=gen).cuda()
model = NeRFModel()
flat = model.flat(model.init(dict(params=1))["params"])
grads = []
for r in range(runs):
    model.backward_kernel = "split" if r == 0 else "ls"
    _, _, _, ctx = model.forward_points(flat, x, d, save=True)
    g = torch.zeros_like(flat)
    model.backward(ctx, gd, gr, None, g)
    torch.cuda.synchronize()
    grads.append(g.clone())
    if r and os.environ.get("LS_POISON"):  # what the next run finds in the scratch: 0xFF bytes (NaN as bf16 / fp32) instead of this run's data
        ctx["ls_scratch"].fill_(int(os.environ["LS_POISON"]))
        torch.cuda.synchronize()
    print("run", r, model.backward_kernel, "status", ls_status(ctx) if r else "-", flush=True)
for r in range(2, runs):
    off = 0
    line = []
    for i, (fi, fo) in enumerate(model.layer_dims()):
        for name, n in (("k", fi * fo), ("b", fo)):
            dlt = (grads[r][off:off + n] - grads[1][off:off + n]).abs()
            if float(dlt.max()) > 0:
                nz = int((dlt > 0).sum())
                idx = int(dlt.argmax())
                line.append(f"Dense_{i}.{name}: {nz} differ, max {float(dlt.max()):.3e} at {idx} (row {idx // fo if name == 'k' else '-'}, col {idx % fo})")
            off += n
    print(f"run {r} vs run 1:", "identical" if not line else "; ".join(line)[:200], "| NaNs", int(torch.isnan(grads[r]).sum()))
for r in range(1, runs):
    off = 0
    rels = []
    for i, (fi, fo) in enumerate(model.layer_dims()):
        n = fi * fo + fo
        a, b = grads[r][off:off + n], grads[0][off:off + n]
        rels.append(f"{float((a - b).norm() / b.norm()):.1e}")
        off += n
    print(f"ls run {r} vs split, rel per Dense layer:", " ".join(rels))
