"""
Generates tests/golden/nerf_hot_path_v1.npz from the oracle (float64 torch-CPU restatement of the
reference; the reference itself cannot run here: JAX/Flax absent, SURVEY.md section 8c — these are
*oracle* vectors, "parity unpinned" with respect to the JAX implementation).

    python tests/golden/make_golden.py

Contents: 16 rays x (16 coarse + 32 fine) samples, the default NeRFModel (one parameter vector used for
both the coarse and the fine model) with weights exactly representable in bf16 (stored as bf16 bit
patterns, 1.2 MB), explicit uniforms, and every intermediate of NeRFRenderer.render_rays plus the loss
and per-layer gradient norms of TrainLoop.losses.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import model as OM  # noqa: E402
from oracle import render as OR  # noqa: E402
from oracle import train as OT  # noqa: E402

F64 = torch.float64
N, TC, TF = 16, 16, 32


def main():
    gen = torch.Generator().manual_seed(20240101)
    dims = OM.nerf_layer_dims()
    flat = OM.lecun_normal_init(dims, gen, dtype=torch.float32)
    off = 0
    for fi, fo in dims:  # non-zero biases; a density head that makes the medium partly opaque
        off += fi * fo
        flat[off:off + fo] = torch.randn(fo, generator=gen) * 0.1
        off += fo
    w9 = OM.param_count(dims[:9])
    flat[w9:w9 + 256] *= 6.0
    flat[w9 + 256] += 1.5
    bits = flat.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)  # bf16-exact weights
    flat64 = torch.from_numpy(bits.view(np.int16).copy()).view(torch.bfloat16).to(F64)

    o = torch.randn(N, 3, generator=gen)
    o = 4 * o / o.norm(dim=-1, keepdim=True)
    d = -o + (torch.rand(N, 3, generator=gen) - 0.5) * 0.6
    d[:2] = torch.tensor([[0.0, 0.0, 1.0], [1.0, 0.0, 0.0]])  # two rays that may miss the box
    d = d / d.norm(dim=-1, keepdim=True)
    c = torch.rand(N, 3, generator=gen) * 2 - 1
    batch = torch.stack([o, d, c], 1).float()
    uc = torch.rand(N, TC, generator=gen).float()
    uf = torch.rand(N, TF, generator=gen).float()
    bg = torch.tensor([0.25, -0.5, 0.75])
    bmin, bmax = torch.tensor([-1.0, -1.0, -1.0], dtype=F64), torch.tensor([1.0, 1.0, 1.0], dtype=F64)

    out = {}
    for tag, rnd in (("exact", None), ("bf16", OM.bf16_round)):
        fn = OM.make_nerf_fn(flat64, rnd)
        r = OR.render_hierarchy(fn, fn, bg.double(), bmin, bmax, batch[:, :2].double(), TC, TF, uc.double(),
                                uf.double())
        for lvl in ("coarse", "fine"):
            for k in ("outputs", "rgbs", "densities", "alphas", "coords"):
                out[f"{tag}_{lvl}_{k}"] = r[lvl][k].numpy()
            out[f"{tag}_{lvl}_ts"] = r[f"{lvl}_ts"].ts.numpy()
        if tag == "exact":
            out["t_min"] = r["coarse_ts"].t_min.numpy()
            out["t_max"] = r["coarse_ts"].t_max.numpy()
            out["mask"] = r["coarse_ts"].mask.numpy()
            out["coarse_probs"] = r["coarse_ts"].termination_probs(r["coarse"]["densities"]).numpy()
        p = [flat64.clone().requires_grad_(True), flat64.clone().requires_grad_(True),
             bg.double().clone().requires_grad_(True)]
        total, ld, _ = OT.losses(OM.make_nerf_fn(p[0], rnd), OM.make_nerf_fn(p[1], rnd), p[2], bmin, bmax,
                                 batch.double(), TC, TF, uc.double(), uf.double())
        grads = torch.autograd.grad(total, p)
        out[f"{tag}_loss_coarse"] = ld["coarse"].detach().numpy()
        out[f"{tag}_loss_fine"] = ld["fine"].detach().numpy()
        norms = []
        for g in grads[:2]:
            off = 0
            for fi, fo in dims:
                norms.append(float(g[off:off + fi * fo].norm()))
                off += fi * fo
                norms.append(float(g[off:off + fo].norm()))
                off += fo
        out[f"{tag}_grad_layer_norms"] = np.array(norms)
        out[f"{tag}_grad_background"] = grads[2].numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nerf_hot_path_v1.npz")
    np.savez_compressed(path, weights_bf16_bits=bits, batch=batch.numpy(), u_coarse=uc.numpy(), u_fine=uf.numpy(),
                        background=bg.numpy(), bbox=np.array([[-1.0, -1, -1], [1, 1, 1]], dtype=np.float32),
                        coarse_ts=np.int32(TC), fine_ts=np.int32(TF), **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
