"""
GPU parity of NeRFModel (learn_nerf/model.py:43-62): fused bf16-MFMA kernels and the exact-fp32
dense path, forward and backward, against the oracle on identical weights and points.

Tolerances
  fp32 path  vs float64 oracle: 2e-5 abs on rgb/density (well inside the 1e-3 gate of north_star).
  bf16x3     (split-precision render kernel, the default for every forward without a backward) vs the float64
             oracle: 5e-5 abs on rgb, 5e-5 relative on density — the reference's fp32 arithmetic to ~16 bits.
  bf16 path  vs the bf16-operand oracle (same roundings, float64 accumulate): 1e-3 abs on rgb.
  bf16 path  vs the float64 oracle: reported, loosely bounded (bf16 has an 8-bit significand).
  gradients  bf16 path vs float64 autograd: relative L2 error per Dense layer < 3e-2.
"""
import pytest
import torch

from oracle import model as OM

pytestmark = pytest.mark.gpu
F64 = torch.float64


def make_points(m, seed=0):
    gen = torch.Generator().manual_seed(seed)
    x = (torch.rand(m, 3, generator=gen) * 2 - 1).float()
    d = torch.randn(m, 3, generator=gen)
    d = (d / d.norm(dim=-1, keepdim=True)).float()
    return x, d, gen


def make_model(precision, seed=1, bias_scale=0.1, render_precision="bf16x3"):
    from learn_nerf.model import NeRFModel

    model = NeRFModel(precision=precision, render_precision=render_precision)
    params = model.init(dict(params=seed))["params"]
    flat = model.flat(params)
    # Flax initialises biases to zero; perturb them so that bias handling is exercised
    gen = torch.Generator().manual_seed(seed + 100)
    noise = torch.zeros(flat.numel())
    off = 0
    for fi, fo in model.layer_dims():
        off += fi * fo
        noise[off:off + fo] = torch.randn(fo, generator=gen) * bias_scale
        off += fo
    flat.add_(noise.cuda())
    return model, params, flat


@pytest.mark.parametrize("m", [1, 777, 4096])
def test_fp32_dense_path_forward(m):
    model, params, flat = make_model("fp32")
    x, d, _ = make_points(m)
    dens, rgb, aux = model.apply(dict(params=params), x.cuda(), d.cuda())
    rd, rr, _ = OM.nerf_mlp(flat.cpu().double(), x.double(), d.double())
    assert aux == {} and dens.shape == (m, 1) and rgb.shape == (m, 3)
    assert (rgb.cpu().double() - rr).abs().max().item() < 2e-5
    assert torch.allclose(dens.cpu().double(), rd, atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize("m", [1, 31, 127, 777, 4096, 10000])
def test_fused_split_precision_forward(m):
    """lnrf_nerf_mlp_fwd_split (bf16 hi+lo operands, 3 MFMAs per product) against the exact float64 oracle."""
    model, params, flat = make_model("bf16")
    x, d, _ = make_points(m, seed=m)
    dens, rgb, aux = model.apply(dict(params=params), x.cuda(), d.cuda())
    rd, rr, _ = OM.nerf_mlp(flat.cpu().double(), x.double(), d.double())
    assert aux == {} and dens.shape == (m, 1) and rgb.shape == (m, 3)
    err = (rgb.cpu().double() - rr).abs().max().item()
    derr = ((dens.cpu().double() - rd).abs() / (1 + rd.abs())).max().item()
    print(f"m={m}: split-precision rgb max|d| vs fp64 oracle {err:.2e}, density rel {derr:.2e}")
    assert err < 5e-5 and derr < 5e-5


def test_split_precision_rays_mode_and_repack():
    """rays mode == points mode; the packed split stream follows parameter updates (cache keyed on version)."""
    from learn_nerf import ops

    model, params, flat = make_model("bf16")
    gen = torch.Generator().manual_seed(5)
    n, t = 77, 19
    o = torch.randn(n, 3, generator=gen)
    dd = torch.randn(n, 3, generator=gen)
    dd = dd / dd.norm(dim=-1, keepdim=True)
    batch = torch.stack([o, dd], 1).float().cuda()
    ts = (torch.rand(n, t, generator=gen) * 2).float().cuda()
    d1, c1, _, _ = model.forward_rays(flat, batch, ts, save=False)
    pts, dirs = ops.ray_points(batch, ts)
    d2, c2, _, _ = model.forward_points(flat, pts.view(-1, 3), dirs.view(-1, 3), save=False)
    assert torch.allclose(d1.reshape(-1), d2, atol=1e-6, rtol=1e-6) and torch.allclose(c1.reshape(-1, 3), c2, atol=1e-6)
    flat.mul_(1.01)  # in-place torch op: bumps the version counter, so the next forward repacks
    d3, c3, _, _ = model.forward_points(flat, pts.view(-1, 3), dirs.view(-1, 3), save=False)
    rd, rr, _ = OM.nerf_mlp(flat.cpu().double(), pts.view(-1, 3).cpu().double(), dirs.view(-1, 3).cpu().double())
    assert (c3.cpu().double() - rr).abs().max().item() < 5e-5
    assert not torch.equal(c3, c2)


def test_apply_with_two_plain_dict_param_sets():
    """model.apply with plain (non-ParamTree) dicts builds temporary flat buffers; a recycled address must not
    hit the pack cache (stale weights)."""
    model, params, flat = make_model("bf16")
    x, d, _ = make_points(300, seed=9)
    outs = []
    for scale in (1.0, 0.5):
        plain = {k: {kk: (vv * scale).clone() for kk, vv in v.items()} for k, v in params.items()}
        _, rgb, _ = model.apply(dict(params=plain), x.cuda(), d.cuda())
        _, rr, _ = OM.nerf_mlp(flat.cpu().double() * scale, x.double(), d.double())
        assert (rgb.cpu().double() - rr).abs().max().item() < 5e-5
        outs.append(rgb)
        del plain
    assert not torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("m", [1, 31, 777, 4096, 10000])
def test_fused_bf16_forward(m):
    model, params, flat = make_model("bf16", render_precision="bf16")  # the arithmetic of the training forward
    x, d, _ = make_points(m, seed=m)
    dens, rgb, _ = model.apply(dict(params=params), x.cuda(), d.cuda())
    f64 = flat.cpu().double()
    rd_b, rr_b, _ = OM.nerf_mlp(f64, x.double(), d.double(), operand_round=OM.bf16_round)
    rd, rr, _ = OM.nerf_mlp(f64, x.double(), d.double())
    diff_b = (rgb.cpu().double() - rr_b).abs()
    diff_f = (rgb.cpu().double() - rr).abs()
    derr_b = ((dens.cpu().double() - rd_b).abs() / (1 + rd_b.abs()))
    print(f"m={m}: rgb |kernel - bf16 oracle| max {diff_b.max():.2e} mean {diff_b.mean():.2e}; "
          f"vs fp64 oracle max {diff_f.max():.2e} mean {diff_f.mean():.2e}; density rel max {derr_b.max():.2e}")
    # Same operand roundings as the oracle; the two differ only where fp32-vs-fp64 accumulation
    # order flips a bf16 rounding of a hidden activation (rare, 1 bf16 ulp = 2^-8 relative).
    # A layout or indexing bug would move the MEAN, which is pinned tightly.
    assert diff_b.mean().item() < 5e-5, "bf16 kernel vs bf16-operand oracle (mean)"
    assert diff_b.max().item() < 5e-3, "bf16 kernel vs bf16-operand oracle (max)"
    assert derr_b.max().item() < 5e-3
    # vs the exact oracle: bf16 operand rounding only (north_star's 1e-3 gate is met by the
    # precision="fp32" path; the bf16 deviation is reported, SURVEY.md section 7 "bf16 vs parity gate")
    assert diff_f.mean().item() < 1e-3 and diff_f.max().item() < 3e-2


def test_fused_rays_mode_matches_points_mode():
    from learn_nerf import ops

    model, params, flat = make_model("bf16", render_precision="bf16")
    gen = torch.Generator().manual_seed(5)
    n, t = 100, 17
    o = torch.randn(n, 3, generator=gen)
    dd = torch.randn(n, 3, generator=gen)
    dd = dd / dd.norm(dim=-1, keepdim=True)
    col = torch.rand(n, 3, generator=gen)
    batch = torch.stack([o, dd, col], 1).float().cuda()  # [N,3,3] stride 9
    ts = (torch.rand(n, t, generator=gen) * 2).float().cuda()
    d1, c1, _, _ = model.forward_rays(flat, batch, ts, save=False)
    pts, dirs = ops.ray_points(batch, ts)
    d2, c2, _, _ = model.forward_points(flat, pts.view(-1, 3), dirs.view(-1, 3), save=False)
    assert torch.allclose(d1.reshape(-1), d2, atol=1e-5, rtol=1e-5)
    assert torch.allclose(c1.reshape(-1, 3), c2, atol=1e-5)


def oracle_grads(flat64, x, d, g_dens, g_rgb, operand_round=None):
    p = flat64.clone().requires_grad_(True)
    dens, rgb, _ = OM.nerf_mlp(p, x.double(), d.double(), operand_round=operand_round)
    loss = (dens[:, 0] * g_dens.double()).sum() + (rgb * g_rgb.double()).sum()
    (g,) = torch.autograd.grad(loss, p)
    return g


def per_layer_rel_err(model, got, ref):
    out, off = [], 0
    for i, (fi, fo) in enumerate(model.layer_dims()):
        for name, n in (("kernel", fi * fo), ("bias", fo)):
            a, b = got[off:off + n], ref[off:off + n]
            out.append((f"Dense_{i}.{name}", ((a - b).norm() / (b.norm() + 1e-30)).item(), b.norm().item()))
            off += n
    return out


@pytest.mark.parametrize("m", [100, 2048 + 5])
def test_fp32_dense_path_backward(m):
    model, params, flat = make_model("fp32")
    x, d, gen = make_points(m, seed=3)
    g_dens = torch.randn(m, generator=gen).float()
    g_rgb = torch.randn(m, 3, generator=gen).float()
    dens, rgb, _, ctx = model.forward_points(flat, x.cuda(), d.cuda(), save=True)
    grad = torch.zeros_like(flat)
    model.backward(ctx, g_dens.cuda(), g_rgb.cuda(), None, grad)
    ref = oracle_grads(flat.cpu().double(), x, d, g_dens, g_rgb)
    errs = per_layer_rel_err(model, grad.cpu().double(), ref)
    for name, err, nrm in errs:
        print(f"{name}: rel L2 err {err:.3e} (|ref| {nrm:.3e})")
    # fp32 kernel vs float64 oracle: ~1e-6, except when a ReLU pre-activation lies within fp32
    # rounding of zero and the mask flips (expected a few times per ~5M units); bound that too.
    tol = 1e-5 if m <= 100 else 5e-3
    for name, err, nrm in errs:
        assert err < tol, (name, err, nrm)


@pytest.mark.parametrize("m", [32, 1000, 8192 + 17])
def test_fused_bf16_backward(m):
    model, params, flat = make_model("bf16")
    x, d, gen = make_points(m, seed=7)
    g_dens = torch.randn(m, generator=gen).float()
    g_rgb = torch.randn(m, 3, generator=gen).float()
    dens, rgb, _, ctx = model.forward_points(flat, x.cuda(), d.cuda(), save=True)
    grad = torch.zeros_like(flat)
    model.backward(ctx, g_dens.cuda(), g_rgb.cuda(), None, grad)
    torch.cuda.synchronize()
    # Oracle for the implementation: same bf16 operand roundings in the forward (so the ReLU masks
    # agree), float64 autograd backward.  The kernel additionally rounds dy_l and W^T to bf16.
    ref = oracle_grads(flat.cpu().double(), x, d, g_dens, g_rgb, operand_round=OM.bf16_round)
    exact = oracle_grads(flat.cpu().double(), x, d, g_dens, g_rgb)
    errs = per_layer_rel_err(model, grad.cpu().double(), ref)
    errs_exact = per_layer_rel_err(model, grad.cpu().double(), exact)
    for (name, err, nrm), (_, err_x, _) in zip(errs, errs_exact):
        print(f"{name}: rel L2 err vs bf16-operand oracle {err:.3e}, vs exact fp64 gradient {err_x:.3e} "
              f"(|ref| {nrm:.3e})")
    for name, err, nrm in errs:
        assert err < 3e-2, (name, err, nrm)
    # vs the exact gradient the deviation is dominated by ReLU units whose sign differs between the
    # bf16 and the exact forward (a property of bf16 training, reported above); bound it loosely.
    for name, err, nrm in errs_exact:
        assert err < 0.25, (name, err, nrm)
    # accumulate semantics: a second backward doubles the gradient (atomics: tiny reordering noise)
    model.backward(ctx, g_dens.cuda(), g_rgb.cuda(), None, grad)
    ref2 = 2 * ref
    tot = ((grad.cpu().double() - ref2).norm() / ref2.norm()).item()
    assert tot < 3e-2


@pytest.mark.parametrize("m", [32, 8192 + 17, 70000])
def test_backward_is_bit_reproducible(m):
    """The weight-gradient launch leaves its partial sums as slabs that a second launch folds in a fixed order
    (fused_chain.h, slab epilogue), so two backward passes over the same inputs give bit-identical gradients — which
    fp32 atomics (LNRF_WGRAD_ATOMICS=1, the older epilogue) do not.  Kernel and bias gradients, ragged and multi-workgroup
    sizes."""
    model, _, flat = make_model("bf16")
    x, d, gen = make_points(m, seed=3)
    g_dens = torch.randn(m, generator=gen).float().cuda()
    g_rgb = torch.randn(m, 3, generator=gen).float().cuda()
    grads = []
    for _ in range(3):
        dens, rgb, _, ctx = model.forward_points(flat, x.cuda(), d.cuda(), save=True)
        g = torch.zeros_like(flat)
        model.backward(ctx, g_dens, g_rgb, None, g)
        grads.append(g)
    assert grads[0].abs().max().item() > 0
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])


@pytest.mark.parametrize("m", [32, 8192 + 17, 70000, 262144])
def test_layer_stationary_backward_equals_two_launch_backward(m):
    """lnrf_nerf_mlp_bwd_ls (head launch + one persistent launch in which every CU owns one Dense layer and the tiles
    pass from CU to CU) must give the gradients of lnrf_nerf_mlp_bwd_chain + _bwd_weights: the same bf16 operands and the
    same fp32 MFMA accumulation per 32-evaluation tile, so the pre-activation gradients are identical and only the order
    of the fp32 partial sums differs.  Per Dense layer; the hand-off status word must stay 0; and the result must be
    bit-reproducible (fixed-order fold of the per-pipeline partial sums), also on a second call with the same scratch."""
    from learn_nerf.model import ls_status

    model, params, flat = make_model("bf16")
    x, d, gen = make_points(m, seed=11)
    g_dens = torch.randn(m, generator=gen).float().cuda()
    g_rgb = torch.randn(m, 3, generator=gen).float().cuda()
    grads = {}
    for kind in ("split", "ls", "ls_again"):
        model.backward_kernel = kind.split("_")[0]
        _, _, _, ctx = model.forward_points(flat, x.cuda(), d.cuda(), save=True)
        g = torch.zeros_like(flat)
        model.backward(ctx, g_dens, g_rgb, None, g)
        torch.cuda.synchronize()
        if kind != "split":
            assert ls_status(ctx) == 0
        grads[kind] = g
    off = 0
    for i, (fi, fo) in enumerate(model.layer_dims()):
        for name, n in (("kernel", fi * fo), ("bias", fo)):
            a, b = grads["ls"][off:off + n], grads["split"][off:off + n]
            rel = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
            assert rel < 1e-5, (f"Dense_{i}.{name}", rel)
            off += n
    assert torch.equal(grads["ls"], grads["ls_again"])


def test_forward_for_the_layer_stationary_backward():
    """lnrf_nerf_mlp_fwd_ls (the saving forward without the hidden layers' ReLU-mask slots) must return bit-identical
    density / rgb to lnrf_nerf_mlp_fwd, and a save it wrote must be refused by the chain backward (which reads those
    slots) instead of producing wrong gradients."""
    model, params, flat = make_model("bf16")
    m = 4096 + 5
    x, d, gen = make_points(m, seed=5)
    outs = {}
    for kind in ("split", "ls"):
        model.backward_kernel = kind
        dens, rgb, _, ctx = model.forward_points(flat, x.cuda(), d.cuda(), save=True)
        outs[kind] = (dens.clone(), rgb.clone(), ctx)
    assert torch.equal(outs["ls"][0], outs["split"][0]) and torch.equal(outs["ls"][1], outs["split"][1])
    assert outs["ls"][2]["hidden_masks"] is False and outs["split"][2]["hidden_masks"] is True
    model.backward_kernel = "split"
    g = torch.zeros_like(flat)
    with pytest.raises(ValueError, match="layer-stationary"):
        model.backward(outs["ls"][2], torch.zeros(m, device="cuda"), torch.zeros(m, 3, device="cuda"), None, g)
