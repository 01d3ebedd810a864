"""
Edge cases and error behaviour of the C ABI (through the ctypes wrappers): empty and ragged inputs,
single-sample rays, all-miss batches, argument errors reported through return codes + lnrf_last_error
(never an abort), and the loud failure on CPU tensors.
"""
import ctypes

import pytest
import torch

from oracle import render as OR

pytestmark = pytest.mark.gpu
F64 = torch.float64
BMIN, BMAX = (-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)


def test_empty_batches_are_noops():
    from learn_nerf import ops

    rays = torch.zeros((0, 2, 3), device="cuda")
    t_min, t_max, mask, ts = ops.ray_aabb_stratified(rays, BMIN, BMAX, 8)
    assert t_min.shape == (0,) and ts.shape == (0, 8)
    out = ops.fine_sample(ts, t_min, t_max, torch.zeros((0, 8), device="cuda"), 4)
    assert out.shape == (0, 12)
    o, a, c, s = ops.composite_fwd(rays, ts, t_min, t_max, mask, torch.zeros((0, 8), device="cuda"),
                                   torch.zeros((0, 8, 3), device="cuda"), torch.zeros(3, device="cuda"))
    assert o.shape == (0, 3) and a.shape == (0,)


def test_single_sample_and_all_miss_rays():
    from learn_nerf import ops

    rays = torch.tensor([[[0.0, 0.0, -3.0], [0.0, 0.0, 1.0]], [[5.0, 5.0, 5.0], [0.0, 1.0, 0.0]],
                         [[9.0, 0.0, 0.0], [0.0, 0.0, 1.0]]], device="cuda")
    u = torch.full((3, 1), 0.25, device="cuda")
    t_min, t_max, mask, ts = ops.ray_aabb_stratified(rays, BMIN, BMAX, 1, u=u)
    assert mask.tolist() == [1, 0, 0]
    assert abs(ts[0, 0].item() - 2.5) < 1e-6  # t_min 2 + 0.25 * (4 - 2)
    assert abs(ts[1, 0].item() - 0.25e-3) < 1e-9  # null range [0, 1e-3] (render.py:387-389)
    dens = torch.tensor([[0.7], [3.0], [3.0]], device="cuda")
    rgb = torch.tensor([[[0.2, 0.4, -0.6]]] * 3, device="cuda")
    bg = torch.tensor([-1.0, 0.0, 1.0], device="cuda")
    out, alpha, coords, _ = ops.composite_fwd(rays, ts, t_min, t_max, mask, dens, rgb, bg)
    a0 = 1 - torch.exp(torch.tensor(-0.7 * 2.0))
    assert torch.allclose(alpha.cpu(), torch.tensor([a0.item(), 0.0, 0.0]), atol=1e-6)
    assert torch.allclose(out[0].cpu(), a0 * torch.tensor([0.2, 0.4, -0.6]) + (1 - a0) * bg.cpu(), atol=1e-6)
    assert torch.equal(out[1:].cpu(), bg.cpu().expand(2, 3)) and torch.equal(coords[1:].cpu(), torch.zeros(2, 3))
    # backward on masked rays: only the background receives gradient
    g_bg = torch.zeros(3, device="cuda")
    gd, gc, _ = ops.composite_bwd(ts, t_min, t_max, mask, dens, rgb, bg, g_bg, g_out=torch.ones(3, 3, device="cuda"))
    assert (gd[1:] == 0).all() and (gc[1:] == 0).all()
    assert torch.allclose(g_bg.cpu(), torch.full((3,), 2.0 + (1 - a0.item())), atol=1e-6)


def test_ragged_sizes_match_oracle():
    """ray counts that are not multiples of the 4-waves-per-block / 32-column tiling"""
    from learn_nerf import ops

    for n, t in ((1, 3), (5, 65), (63, 129)):
        gen = torch.Generator().manual_seed(n)
        o = torch.randn(n, 3, generator=gen)
        o = 4 * o / o.norm(dim=-1, keepdim=True)
        d = -o / o.norm(dim=-1, keepdim=True)
        rays = torch.stack([o, d], 1).float()
        u = torch.rand(n, t, generator=gen)
        bbox = torch.tensor([BMIN, BMAX], dtype=F64)
        tm, tx, mk = OR.ray_t_range(bbox, rays.double())
        s = OR.RaySamples(tm, tx, mk, OR.stratified_ts(tm, tx, t, u.double()))
        dens = (torch.rand(n, t, generator=gen) * 3).float()
        rgb = (torch.rand(n, t, 3, generator=gen) * 2 - 1).float()
        bg = torch.tensor([0.1, 0.2, 0.3])
        t_min, t_max, mask, ts = ops.ray_aabb_stratified(rays.cuda(), BMIN, BMAX, t, u=u.cuda())
        out, _, _, _ = ops.composite_fwd(rays.cuda(), ts, t_min, t_max, mask, dens.cuda(), rgb.cuda(), bg.cuda())
        ref = s.render_rays(dens.double(), rgb.double(), bg.double())
        assert (out.cpu().double() - ref).abs().max().item() < 5e-6


def test_argument_errors_return_codes_and_messages():
    from learn_nerf import _lib as L

    lib = L.lib()
    buf = torch.zeros(64, device="cuda")
    p = ctypes.c_void_p(buf.data_ptr())
    rc = lib.lnrf_dense_fwd(p, 0, p, None, 0, p, 8, 4, 8, 8, None)  # ldx < k
    assert rc == -1 and b"bad sizes" in lib.lnrf_last_error()
    rc = lib.lnrf_adam_step(None, p, p, p, 4, 1e-3, 0.9, 0.999, 1e-7, 1, 1.0, None)
    assert rc == -1 and b"null pointer" in lib.lnrf_last_error()
    rc = lib.lnrf_adam_step(p, p, p, p, 4, 1e-3, 0.9, 0.999, 1e-7, 0, 1.0, None)  # step counts from 1
    assert rc == -1
    shape = L.NerfShape(5, 4, 128, 128, 10, 4)
    rc = lib.lnrf_nerf_pack_weights(ctypes.byref(shape), p, p, None)
    assert rc == -3 and b"default NeRFModel shape" in lib.lnrf_last_error()  # LNRF_ERR_UNSUPPORTED
    rc = lib.lnrf_integrated_directional_encoding(9, p, None, 4, p, None)
    assert rc == -1
    with pytest.raises(RuntimeError, match="lnrf error"):
        L.check(rc, "ide")


def test_non_default_nerf_shape_uses_dense_path():
    from learn_nerf.model import NeRFModel
    from oracle import model as OM

    model = NeRFModel(hidden_dim=64, color_layer_dim=32, x_freqs=6, d_freqs=2, input_layers=3, mid_layers=2,
                      precision="fp32")
    assert not model.fused_supported()
    params = model.init(dict(params=0))["params"]
    gen = torch.Generator().manual_seed(0)
    x = (torch.rand(100, 3, generator=gen) * 2 - 1).float()
    d = torch.randn(100, 3, generator=gen).float()
    dens, rgb, _ = model.apply(dict(params=params), x.cuda(), d.cuda())  # exact dense path
    rd, rr, _ = OM.nerf_mlp(model.flat(params).cpu().double(), x.double(), d.double(), input_layers=3, mid_layers=2,
                            hidden_dim=64, color_layer_dim=32, x_freqs=6, d_freqs=2)
    assert (rgb.cpu().double() - rr).abs().max().item() < 2e-5
    # precision="bf16" on a shape without a fused kernel: bf16 operands on the same dense path
    model16 = NeRFModel(hidden_dim=64, color_layer_dim=32, x_freqs=6, d_freqs=2, input_layers=3, mid_layers=2)
    _, rgb16, _ = model16.apply(dict(params=params), x.cuda(), d.cuda())
    _, rr16, _ = OM.nerf_mlp(model.flat(params).cpu().float(), x, d, input_layers=3, mid_layers=2, hidden_dim=64,
                             color_layer_dim=32, x_freqs=6, d_freqs=2, operand_round=OM.bf16_round)
    assert (rgb16.cpu() - rr16).abs().max().item() < 2e-3
    assert (rgb16.cpu().double() - rr).abs().max().item() > 1e-5


def test_cpu_tensor_is_rejected_loudly():
    from learn_nerf import ops

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.stratified(torch.zeros(4), torch.ones(4), 8)
