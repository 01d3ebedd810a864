"""
Generates tests/golden/ngp_refnerf_v1.npz from the float64 oracle (oracle/instant_ngp.py, oracle/ref_nerf.py;
"parity unpinned" with respect to the JAX implementation, which cannot run here: SURVEY.md section 8c).

    python tests/golden/make_golden_ngp_refnerf.py

Contents (small on purpose, 100 KB): one InstantNGPModel (4 levels, dense and hashed, T = 2^10) and one
RefNERFModel (hidden 32, colour 16, sh_degree 3), each with its flat parameter vector, 48 points / directions,
the model outputs, and the gradient of a fixed linear functional of the outputs wrt the parameters.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import instant_ngp as ON  # noqa: E402
from oracle import model as OM  # noqa: E402
from oracle import ref_nerf as ORF  # noqa: E402

F64 = torch.float64
M = 48
NGP = dict(table_sizes=[2 ** 10] * 4, grid_sizes=[4, 8, 16, 64], bbox_min=(-1.0, -0.5, -2.0), bbox_max=(1.0, 1.5, 0.5))
REF = dict(sh_degree=3, hidden_dim=32, color_layer_dim=16)


def main():
    gen = torch.Generator().manual_seed(20240202)
    out = {}
    lo, hi = torch.tensor(NGP["bbox_min"]), torch.tensor(NGP["bbox_max"])
    x = (torch.rand(M, 3, generator=gen) * (hi - lo) * 1.1 + lo - 0.05 * (hi - lo)).float()
    x[:3] = torch.stack([lo, hi, (lo + hi) / 2])
    d = torch.randn(M, 3, generator=gen)
    d = (d / d.norm(dim=-1, keepdim=True)).float()
    g_d = torch.randn(M, generator=gen).float()
    g_c = torch.randn(M, 3, generator=gen).float()
    out.update(x=x.numpy(), d=d.numpy(), g_density=g_d.numpy(), g_rgb=g_c.numpy())

    # ---- InstantNGPModel
    rows, dims = ON.ngp_spec(NGP["table_sizes"], NGP["grid_sizes"])
    nt = sum(r * 2 for r in rows)
    flat = torch.cat([(torch.rand(nt, generator=gen) * 2 - 1) * 0.5, OM.lecun_normal_init(dims, gen)]).float()
    off = nt
    for fi, fo in dims:
        off += fi * fo
        flat[off:off + fo] = torch.randn(fo, generator=gen) * 0.1
        off += fo
    f64 = flat.double().requires_grad_(True)
    dens, rgb, _ = ON.ngp_model(f64, x.double(), d.double(), NGP["table_sizes"], NGP["grid_sizes"], NGP["bbox_min"],
                                NGP["bbox_max"])
    (grad,) = torch.autograd.grad((dens[:, 0] * g_d.double()).sum() + (rgb * g_c.double()).sum(), f64)
    out.update(ngp_flat=flat.numpy(), ngp_density=dens.detach().numpy(), ngp_rgb=rgb.detach().numpy(),
               ngp_grad=grad.numpy(), ngp_table_floats=np.int64(nt))

    # ---- RefNERFModel
    rdims = ORF.ref_nerf_layer_dims(hidden_dim=REF["hidden_dim"], color_layer_dim=REF["color_layer_dim"],
                                    sh_degree=REF["sh_degree"])
    rflat = OM.lecun_normal_init(rdims, gen).float()
    off = 0
    for fi, fo in rdims:
        off += fi * fo
        rflat[off:off + fo] = torch.randn(fo, generator=gen) * 0.1
        off += fo
    xr = (torch.rand(M, 3, generator=gen) * 2 - 1).float()
    g_a = {k: torch.rand(M, generator=gen).float() for k in ("normal_mse", "neg_normal")}
    r64 = rflat.double().requires_grad_(True)
    rd, rr, raux = ORF.ref_nerf_model(r64, xr.double(), d.double(), **REF)
    loss = (rd[:, 0] * g_d.double()).sum() + (rr * g_c.double()).sum() + sum((raux[k] * g_a[k].double()).sum()
                                                                             for k in g_a)
    (rgrad,) = torch.autograd.grad(loss, r64)
    out.update(ref_flat=rflat.numpy(), ref_x=xr.numpy(), ref_density=rd.detach().numpy(), ref_rgb=rr.detach().numpy(),
               ref_normal_mse=raux["normal_mse"].detach().numpy(), ref_neg_normal=raux["neg_normal"].detach().numpy(),
               ref_g_normal_mse=g_a["normal_mse"].numpy(), ref_g_neg_normal=g_a["neg_normal"].numpy(),
               ref_grad=rgrad.numpy())
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ngp_refnerf_v1.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
