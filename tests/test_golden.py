"""
Golden vectors (tests/golden/nerf_hot_path_v1.npz, produced by tests/golden/make_golden.py from the
float64 oracle):
  * CPU: the oracle reproduces them (regression pin of the oracle) and they satisfy the analytic
    identities of SURVEY.md section 8c;
  * GPU: the HIP path (exact-fp32 and bf16-MFMA), called through the C ABI, reproduces them.
"""
import os

import numpy as np
import pytest
import torch

from oracle import model as OM
from oracle import render as OR

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "nerf_hot_path_v1.npz"))
F64 = torch.float64


def weights64():
    bits = G["weights_bf16_bits"]
    return torch.from_numpy(bits.view(np.int16).copy()).view(torch.bfloat16).to(F64)


def test_golden_identities():
    n, tc, tf = G["batch"].shape[0], int(G["coarse_ts"]), int(G["fine_ts"])
    assert G["exact_fine_ts"].shape == (n, tc + tf)
    assert np.allclose(G["coarse_probs"].sum(1), 1.0, atol=1e-12)  # identity (2)
    miss = ~G["mask"]
    assert miss.any() and (~miss).any()
    assert np.allclose(G["exact_fine_outputs"][miss], G["background"][None].astype(np.float64))  # identity (4)
    assert (G["exact_fine_alphas"][miss] == 0).all()
    assert (np.diff(G["exact_fine_ts"], axis=1) >= 0).all()  # sorted (8)
    for i in range(n):
        assert np.isin(G["exact_coarse_ts"][i], G["exact_fine_ts"][i]).all()  # coarse ts are a subset (8)
    a = G["exact_fine_alphas"][:, 0]
    assert a.max() > 0.5 and a.min() < 0.05


@pytest.mark.parametrize("tag", ["exact", "bf16"])
def test_oracle_reproduces_golden(tag):
    flat = weights64()
    rnd = OM.bf16_round if tag == "bf16" else None
    fn = OM.make_nerf_fn(flat, rnd)
    batch = torch.from_numpy(G["batch"]).double()
    r = OR.render_hierarchy(fn, fn, torch.from_numpy(G["background"]).double(), torch.tensor([-1.0] * 3, dtype=F64),
                            torch.tensor([1.0] * 3, dtype=F64), batch[:, :2], int(G["coarse_ts"]), int(G["fine_ts"]),
                            torch.from_numpy(G["u_coarse"]).double(), torch.from_numpy(G["u_fine"]).double())
    for lvl in ("coarse", "fine"):
        assert np.allclose(r[lvl]["outputs"].numpy(), G[f"{tag}_{lvl}_outputs"], atol=1e-10)
        assert np.allclose(r[f"{lvl}_ts"].ts.numpy(), G[f"{tag}_{lvl}_ts"], atol=1e-10)


def _loop(precision):
    """precision: "fp32" dense path | "bf16" fused path with the default split-precision render kernel |
    "bf16-plain" fused path rendering with the plain bf16 kernel (training-forward arithmetic)."""
    from learn_nerf.model import NeRFModel
    from learn_nerf.train import TrainLoop

    tc, tf = int(G["coarse_ts"]), int(G["fine_ts"])
    kw = dict(precision="fp32") if precision == "fp32" else dict(
        render_precision="bf16" if precision == "bf16-plain" else "bf16x3")
    loop = TrainLoop(NeRFModel(**kw), NeRFModel(**kw), init_rng=0, lr=1e-3, coarse_ts=tc, fine_ts=tf)
    w = weights64().float().cuda()
    c, f, bg = loop._slices(loop.flat)
    c.copy_(w)
    f.copy_(w)
    bg.copy_(torch.from_numpy(G["background"]).cuda())
    loop._params_changed()
    return loop


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16-plain"])
def test_hip_path_reproduces_golden_render(precision):
    from learn_nerf.render import NeRFRenderer
    from learn_nerf.rng import Uniforms

    loop = _loop(precision)
    p = loop.state.params
    renderer = NeRFRenderer(coarse=loop.coarse, fine=loop.fine, coarse_params=p["coarse"], fine_params=p["fine"],
                            background=p["background"], bbox_min=G["bbox"][0], bbox_max=G["bbox"][1],
                            coarse_ts=loop.coarse_ts, fine_ts=loop.fine_ts)
    key = (Uniforms(torch.from_numpy(G["u_coarse"]).cuda()), Uniforms(torch.from_numpy(G["u_fine"]).cuda()))
    out = renderer.render_rays(key, torch.from_numpy(G["batch"][:, :2].copy()).cuda())
    # north_star gate (1e-3 vs the exact vectors) for the dense fp32 path and for the fused render kernel
    # (split precision); the plain bf16 kernel is checked against the bf16-operand vectors at 4e-3
    tag = "bf16" if precision == "bf16-plain" else "exact"
    tol = 4e-3 if precision == "bf16-plain" else 1e-3
    for lvl in ("coarse", "fine"):
        got = out[lvl]["outputs"].cpu().double().numpy()
        err = np.abs(got - G[f"{tag}_{lvl}_outputs"]).max()
        print(f"{precision} {lvl}: max |rgb - golden| = {err:.2e}")
        assert err < tol
        assert np.abs(out[lvl]["alphas"].cpu().double().numpy() - G[f"{tag}_{lvl}_alphas"]).max() < tol
    t_min, t_max, mask = renderer.t_range(torch.from_numpy(G["batch"][:, :2].copy()).cuda())
    assert np.array_equal(mask.cpu().numpy(), G["mask"])
    assert np.allclose(t_min.cpu().numpy(), G["t_min"], atol=1e-5) and np.allclose(t_max.cpu().numpy(), G["t_max"], atol=1e-5)
    if precision != "bf16-plain":
        assert np.abs(out["coarse"]["densities"].cpu().double().numpy() - G["exact_coarse_densities"]).max() < 1e-3
        assert np.abs(out["fine"]["coords"].cpu().double().numpy() - G["exact_fine_coords"]).max() < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_hip_path_reproduces_golden_loss_and_gradient(precision):
    from learn_nerf.rng import Uniforms

    loop = _loop(precision)
    tag = "exact" if precision == "fp32" else "bf16"
    key = (("unused", "unused"), None)
    # explicit uniforms: (render_key, density_key) with render_key = (coarse, fine)
    render_key = (Uniforms(torch.from_numpy(G["u_coarse"]).cuda()), Uniforms(torch.from_numpy(G["u_fine"]).cuda()))
    loop.grad.zero_()
    ld, _ = loop._forward_backward((render_key, 0), tuple(G["bbox"][0]), tuple(G["bbox"][1]),
                                   torch.from_numpy(G["batch"]).cuda(), loop.flat, loop.grad, True)
    ltol = 1e-5 if precision == "fp32" else 2e-3
    # with want_grad the two mean squared errors stay in the step's accumulators (lnrf_step_log formats them)
    inv = 1.0 / (3.0 * G["batch"].shape[0])
    assert abs(float(loop._scalars[0]) * inv - float(G[f"{tag}_loss_coarse"])) < ltol
    assert abs(float(loop._scalars[1]) * inv - float(G[f"{tag}_loss_fine"])) < ltol
    c, f, bg = loop._slices(loop.grad)
    norms = []
    for g in (c, f):
        off = 0
        for fi, fo in loop.coarse.layer_dims():
            norms.append(float(g[off:off + fi * fo].norm()))
            off += fi * fo
            norms.append(float(g[off:off + fo].norm()))
            off += fo
    ref = G[f"{tag}_grad_layer_norms"]
    rel = np.abs(np.array(norms) - ref) / ref
    print(f"{precision}: max rel error of per-layer gradient norms {rel.max():.2e}")
    assert rel.max() < (2e-3 if precision == "fp32" else 3e-2)
    assert np.allclose(bg.cpu().double().numpy(), G[f"{tag}_grad_background"], rtol=3e-2, atol=1e-6)
