"""
Golden vectors for the hash-grid and Ref-NeRF paths (tests/golden/ngp_refnerf_v1.npz, produced by
tests/golden/make_golden_ngp_refnerf.py from the float64 oracle; "parity unpinned" with respect to JAX):
  * CPU: the oracle reproduces them (regression pin) and they satisfy the structural identities of the models;
  * GPU: InstantNGPModel (exact-fp32 and fused bf16 MLP) and RefNERFModel, through the C ABI, reproduce them.
"""
import os

import numpy as np
import pytest
import torch

from oracle import instant_ngp as ON
from oracle import ref_nerf as ORF

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "ngp_refnerf_v1.npz"))
NGP = dict(table_sizes=[2 ** 10] * 4, grid_sizes=[4, 8, 16, 64], bbox_min=(-1.0, -0.5, -2.0), bbox_max=(1.0, 1.5, 0.5))
REF = dict(sh_degree=3, hidden_dim=32, color_layer_dim=16)


def T(name, dtype=torch.float64):
    return torch.from_numpy(G[name]).to(dtype)


def test_golden_structure():
    rows, dims = ON.ngp_spec(NGP["table_sizes"], NGP["grid_sizes"])
    assert rows == [64, 512, 1024, 1024]  # 4^3, 8^3 dense; 16^3, 64^3 hashed into 2^10 rows (instant_ngp.py:178)
    assert int(G["ngp_table_floats"]) == 2 * sum(rows) and G["ngp_flat"].size == ON.ngp_param_count(NGP["table_sizes"], NGP["grid_sizes"])
    assert (G["ngp_density"] > 0).all() and np.abs(G["ngp_rgb"]).max() < 1  # exp / tanh ranges
    assert (G["ref_density"] >= 0).all() and G["ref_rgb"].min() >= -1.0 - 1e-9  # softplus; sRGB*2-1 with leaky clip
    assert (G["ref_normal_mse"] >= 0).all() and (G["ref_neg_normal"] >= 0).all()
    # untouched table rows have exactly zero gradient; touched ones do not
    nt = int(G["ngp_table_floats"])
    gt = G["ngp_grad"][:nt]
    assert (gt == 0).any() and (gt != 0).any()


def test_oracle_reproduces_golden():
    dens, rgb, _ = ON.ngp_model(T("ngp_flat"), T("x"), T("d"), NGP["table_sizes"], NGP["grid_sizes"], NGP["bbox_min"],
                                NGP["bbox_max"])
    assert np.allclose(dens.numpy(), G["ngp_density"], atol=1e-12) and np.allclose(rgb.numpy(), G["ngp_rgb"], atol=1e-12)
    rd, rr, aux = ORF.ref_nerf_model(T("ref_flat"), T("ref_x"), T("d"), **REF)
    assert np.allclose(rd.detach().numpy(), G["ref_density"], atol=1e-10)
    assert np.allclose(rr.detach().numpy(), G["ref_rgb"], atol=1e-10)
    assert np.allclose(aux["normal_mse"].detach().numpy(), G["ref_normal_mse"], atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_hip_ngp_reproduces_golden(precision):
    from learn_nerf.instant_ngp import InstantNGPModel

    model = InstantNGPModel(precision=precision, **NGP)
    assert model._use_fused() == (precision == "bf16")
    flat = T("ngp_flat", torch.float32).cuda()
    x, d = T("x", torch.float32).cuda(), T("d", torch.float32).cuda()
    dens, rgb, _, ctx = model.forward_points(flat, x, d, save=True)
    tol = 1e-4 if precision == "fp32" else 2e-2  # bf16 operands; exact float64 golden values
    assert np.abs(rgb.cpu().double().numpy() - G["ngp_rgb"]).max() < tol
    assert (np.abs(dens.cpu().double().numpy() - G["ngp_density"][:, 0]) / (1 + G["ngp_density"][:, 0])).max() < tol
    grad = torch.zeros_like(flat)
    model.backward(ctx, T("g_density", torch.float32).cuda(), T("g_rgb", torch.float32).cuda(), None, grad)
    ref = G["ngp_grad"]
    rel = np.linalg.norm(grad.cpu().double().numpy() - ref) / np.linalg.norm(ref)
    print(f"ngp {precision}: gradient rel L2 err vs golden {rel:.2e}")
    assert rel < (1e-4 if precision == "fp32" else 6e-2)
    nt = int(G["ngp_table_floats"])
    assert ((grad[:nt].cpu().numpy() == 0) == (ref[:nt] == 0)).all()  # same set of touched table entries


@pytest.mark.gpu
def test_hip_ref_nerf_reproduces_golden():
    from learn_nerf.ref_nerf import RefNERFModel

    model = RefNERFModel(precision="fp32", **REF)
    flat = T("ref_flat", torch.float32).cuda()
    x, d = T("ref_x", torch.float32).cuda(), T("d", torch.float32).cuda()
    dens, rgb, aux, ctx = model.forward_points(flat, x, d, save=True)
    assert np.abs(rgb.cpu().double().numpy() - G["ref_rgb"]).max() < 1e-4
    assert np.abs(dens.reshape(-1).cpu().double().numpy() - G["ref_density"][:, 0]).max() < 1e-4 * (1 + G["ref_density"].max())
    assert np.abs(aux["normal_mse"].cpu().double().numpy() - G["ref_normal_mse"]).max() < 2e-3
    assert np.abs(aux["neg_normal"].cpu().double().numpy() - G["ref_neg_normal"]).max() < 2e-3
    grad = torch.zeros_like(flat)
    g_aux = {"normal_mse": T("ref_g_normal_mse", torch.float32).cuda(), "neg_normal": T("ref_g_neg_normal", torch.float32).cuda()}
    model.backward(ctx, T("g_density", torch.float32).cuda(), T("g_rgb", torch.float32).cuda(), g_aux, grad)
    rel = np.linalg.norm(grad.cpu().double().numpy() - G["ref_grad"]) / np.linalg.norm(G["ref_grad"])
    print(f"ref-nerf: gradient rel L2 err vs golden {rel:.2e}")
    assert rel < 1e-3
