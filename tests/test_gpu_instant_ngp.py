"""
GPU parity of the hash-grid path (instant_ngp.py:34-54, 92-224): gather kernel, scatter-add kernel and
InstantNGPModel forward / backward / train step against the oracle on identical tables and points.
fp32 kernels vs float64 oracle; scatter-add sums fp32 atomics in arbitrary order (tolerance stated).
"""
import pytest
import torch

from oracle import instant_ngp as ON
from oracle import philox
from oracle import train as OT

pytestmark = pytest.mark.gpu
F64 = torch.float64
BMIN, BMAX = (-1.0, -0.5, -2.0), (1.0, 1.5, 0.5)


def points(m, seed=0):
    gen = torch.Generator().manual_seed(seed)
    lo, hi = torch.tensor(BMIN), torch.tensor(BMAX)
    x = torch.rand(m, 3, generator=gen) * (hi - lo) * 1.1 + lo - 0.05 * (hi - lo)  # some just outside the box
    x[:3] = torch.stack([lo, hi, (lo + hi) / 2])  # corners / centre exactly
    d = torch.randn(m, 3, generator=gen)
    return x.float().contiguous(), (d / d.norm(dim=-1, keepdim=True)).float().contiguous(), gen


@pytest.mark.parametrize("smooth", [False, True])
@pytest.mark.parametrize("cfg", [([2 ** 10] * 4, [4, 8, 16, 64]), ([2 ** 14] * 6, [16, 16, 32, 32, 64, 64])])
def test_hashgrid_fwd_bwd(cfg, smooth):
    from learn_nerf.instant_ngp import MultiresHashTableEncoding

    ts, gs = cfg
    enc = MultiresHashTableEncoding(ts, gs, BMIN, BMAX, 2, smooth)
    m = 3000
    x, _, gen = points(m)
    tables = (torch.rand(enc.num_table_floats(), generator=gen) * 2 - 1).float()
    out = enc.apply(tables.cuda(), x.cuda())
    assert out.shape == (m, 2 * len(gs))
    t64 = tables.double().requires_grad_(True)
    off, feats = 0, []
    for r, t, g in zip(enc.rows(), ts, gs):
        feats.append(ON.hash_table_encoding(x.double(), t64[off:off + 2 * r].reshape(r, 2), g, t,
                                            torch.tensor(BMIN, dtype=F64), torch.tensor(BMAX, dtype=F64), smooth))
        off += 2 * r
    ref = torch.cat(feats, 1)
    assert (out.cpu().double() - ref).abs().max().item() < 2e-5  # fp32 cell coordinates at G = 64
    g_enc = torch.randn(m, 2 * len(gs), generator=gen).float()
    (g_ref,) = torch.autograd.grad((ref * g_enc.double()).sum(), t64)
    from learn_nerf import ops

    g_tab = torch.zeros_like(tables).cuda()
    ops.hashgrid_bwd(enc.desc(), x.cuda(), g_enc.t().contiguous().cuda(), g_tab)
    err = (g_tab.cpu().double() - g_ref).abs().max().item()
    assert err < 2e-4 * max(1.0, g_ref.abs().max().item())


def make_model(levels, table, hidden=64, seed=2, precision="fp32"):
    from learn_nerf.instant_ngp import InstantNGPModel

    model = InstantNGPModel(table_sizes=[table] * levels, grid_sizes=[2 ** (4 + i // 2) for i in range(levels)],
                            bbox_min=BMIN, bbox_max=BMAX, hidden_dim=hidden, precision=precision)
    params = model.init(dict(params=seed))["params"]
    flat = model.flat(params)
    nt = model.encoding().num_table_floats()
    gen = torch.Generator().manual_seed(seed)
    flat[:nt] = ((torch.rand(nt, generator=gen) * 2 - 1) * 0.5).cuda()  # U(-1e-4,1e-4) init would hide gather errors
    return model, params, flat


@pytest.mark.parametrize("levels,table", [(6, 2 ** 12), (16, 2 ** 14)])
def test_ngp_model_forward_backward(levels, table):
    model, params, flat = make_model(levels, table)
    assert model.num_params() == ON.ngp_param_count(model.table_sizes, model.grid_sizes)
    m = 2500
    x, d, gen = points(m, seed=levels)
    dens, rgb, aux = model.apply(dict(params=params), x.cuda(), d.cuda())
    f64 = flat.cpu().double().requires_grad_(True)
    rd, rr, _ = ON.ngp_model(f64, x.double(), d.double(), model.table_sizes, model.grid_sizes, BMIN, BMAX)
    assert aux == {} and dens.shape == (m, 1)
    assert (rgb.cpu().double() - rr).abs().max().item() < 1e-4
    assert ((dens.cpu().double() - rd).abs() / (1 + rd.abs())).max().item() < 1e-4
    g_d = torch.randn(m, generator=gen).float()
    g_c = torch.randn(m, 3, generator=gen).float()
    (g_ref,) = torch.autograd.grad((rd[:, 0] * g_d.double()).sum() + (rr * g_c.double()).sum(), f64)
    _, _, _, ctx = model.forward_points(flat, x.cuda(), d.cuda(), save=True)
    grad = torch.zeros_like(flat)
    model.backward(ctx, g_d.cuda(), g_c.cuda(), None, grad)
    nt = model.encoding().num_table_floats()
    got = grad.cpu().double()
    # The reference arithmetic is fp32: at G = 2048 the fp32 cell coordinate (G-1)*frac carries ~1e-4 of a
    # cell, so a float64 evaluation puts a few samples into neighbouring cells.  The table gradient is
    # therefore compared with the oracle run in float32 (same cells), the float64 figure is printed.
    f32 = flat.cpu().float().requires_grad_(True)
    rd32, rr32, _ = ON.ngp_model(f32, x, d, model.table_sizes, model.grid_sizes, BMIN, BMAX)
    (g_ref32,) = torch.autograd.grad((rd32[:, 0] * g_d).sum() + (rr32 * g_c).sum(), f32)
    for name, a, b, b32 in (("tables", got[:nt], g_ref[:nt], g_ref32[:nt].double()),
                            ("mlp", got[nt:], g_ref[nt:], g_ref32[nt:].double())):
        rel = ((a - b).norm() / b.norm()).item()
        rel32 = ((a - b32).norm() / b32.norm()).item()
        print(f"L={levels} {name}: rel L2 err vs float64 oracle {rel:.2e}, vs float32 oracle {rel32:.2e}")
        assert min(rel, rel32) < 2e-3, (name, rel, rel32)  # ReLU sign flips of the 64-wide MLP are the floor
        assert rel < 2e-2


@pytest.mark.parametrize("levels,table,m", [(6, 2 ** 12, 2500), (16, 2 ** 14, 4133), (3, 2 ** 10, 31)])
def test_ngp_fused_mlp_vs_bf16_oracle(levels, table, m, monkeypatch):
    """
    The backward's scratch buffer (partial weight-gradient rows of the fused kernel) is poisoned with NaN bytes first:
    every word the reduce launch folds must have been written by the backward kernel.
    lnrf_ngp_mlp_fwd / lnrf_ngp_mlp_bwd (bf16 MFMA operands, fp32 accumulate) against the oracle with the same
    operand rounding.  Forward: 2e-3 (bf16 rounding boundaries of intermediate activations); gradients: 3e-2
    relative L2 (ReLU mask flips + bf16 dy operands), as for the fused NeRFModel.  The distance to the exact
    float64 model is printed.  m = 4133 / 31 exercise ragged tiles and partially filled workgroups.
    """
    from oracle.model import bf16_round

    import learn_nerf.instant_ngp as NGP

    monkeypatch.setattr(NGP, "POISON_SCRATCH", True)
    model, params, flat = make_model(levels, table, precision="bf16")
    assert model._use_fused()
    x, d, gen = points(m, seed=levels)
    _, _, _, ctx = model.forward_points(flat, x.cuda(), d.cuda(), save=True)
    model.render_precision = "bf16"  # the plain-operand kernel (= the training forward) is what this test compares
    dens, rgb, _, _ = model.forward_points(flat, x.cuda(), d.cuda(), save=False)
    model.render_precision = "bf16x3"
    assert ctx["kind"] == "fused"
    f32 = flat.cpu().float().requires_grad_(True)
    rd, rr, _ = ON.ngp_model(f32, x, d, model.table_sizes, model.grid_sizes, BMIN, BMAX, operand_round=bf16_round)
    ed, er, _ = ON.ngp_model(flat.cpu().double(), x.double(), d.double(), model.table_sizes, model.grid_sizes, BMIN, BMAX)
    e_rgb = (rgb.cpu() - rr).abs().max().item()
    e_den = ((dens.cpu() - rd[:, 0]).abs() / (1 + rd[:, 0].abs())).max().item()
    x_rgb = (rgb.cpu().double() - er).abs().max().item()
    print(f"L={levels} m={m}: fused vs bf16 oracle rgb {e_rgb:.2e} density {e_den:.2e}; vs exact rgb {x_rgb:.2e}")
    assert e_rgb < 2e-3 and e_den < 2e-3
    assert x_rgb < 3e-2
    g_d = torch.randn(m, generator=gen).float()
    g_c = torch.randn(m, 3, generator=gen).float()
    (g_ref,) = torch.autograd.grad((rd[:, 0] * g_d).sum() + (rr * g_c).sum(), f32)
    grad = torch.zeros_like(flat)
    model.backward(ctx, g_d.cuda(), g_c.cuda(), None, grad)
    nt = model.encoding().num_table_floats()
    got = grad.cpu()
    assert torch.isfinite(got).all()
    for name, a, b in (("tables", got[:nt], g_ref[:nt]), ("mlp", got[nt:], g_ref[nt:])):
        rel = ((a - b).norm() / b.norm()).item()
        print(f"   {name}: rel L2 err vs bf16-operand oracle {rel:.2e}")
        assert rel < 3e-2, (name, rel)
    # per-layer check so that a misplaced block cannot hide in the norm of a larger one
    off = nt
    for i, (fi, fo) in enumerate(model.dense_dims()):
        for kind, n in (("kernel", fi * fo), ("bias", fo)):
            a, b = got[off:off + n], g_ref[off:off + n]
            rel = ((a - b).norm() / (b.norm() + 1e-12)).item()
            assert rel < 6e-2, (f"Dense_{i}/{kind}", rel)
            off += n
    # a second backward into the same buffer accumulates
    model.backward(ctx, g_d.cuda(), g_c.cuda(), None, grad)
    assert ((grad.cpu() - 2 * got).norm() / got.norm()).item() < 1e-3


@pytest.mark.parametrize("levels,table,m", [(6, 2 ** 12, 2500), (16, 2 ** 14, 4133), (3, 2 ** 10, 31), (16, 2 ** 14, 70000)])
def test_ngp_fused_split_forward_matches_exact(levels, table, m):
    """lnrf_ngp_mlp_fwd_split — the default for every forward without a backward (rendering, evaluation, model.apply): bf16
    hi + lo operand pairs, three MFMAs per product, fp32 accumulate — against the EXACT float64 model
    (instant_ngp.py:38-54 restated in oracle/instant_ngp.py): 5e-5 absolute on rgb, 5e-5 relative on density."""
    model, params, flat = make_model(levels, table, precision="bf16")
    assert model._use_fused() and model.render_precision == "bf16x3"
    x, d, _ = points(m, seed=levels + 1)
    dens, rgb, aux = model.apply(dict(params=params), x.cuda(), d.cuda())
    ed, er, _ = ON.ngp_model(flat.cpu().double(), x.double(), d.double(), model.table_sizes, model.grid_sizes, BMIN, BMAX)
    e_rgb = (rgb.cpu().double() - er).abs().max().item()
    e_den = ((dens.cpu().double() - ed).abs() / (1e-3 + ed.abs())).max().item()
    print(f"L={levels} m={m}: split-precision forward vs exact: rgb {e_rgb:.2e}, density rel {e_den:.2e}")
    assert aux == {} and dens.shape == (m, 1) and rgb.shape == (m, 3)
    assert e_rgb < 5e-5 and e_den < 5e-5


def test_ngp_renderer_meets_the_1e3_gate_on_the_fused_path():
    """north_star: rendered RGB within 1e-3 of the reference's fp32 arithmetic on identical rays.  NeRFRenderer over two
    InstantNGPModels in their DEFAULT configuration (precision "bf16": fused kernels; the renderer's forwards carry no
    backward, so the MLP runs lnrf_ngp_mlp_fwd_split) at 256 rays x (64 + 128) samples against the exact float64 oracle."""
    from learn_nerf.instant_ngp import InstantNGPModel
    from learn_nerf.render import NeRFRenderer
    from learn_nerf.rng import Key, split
    from learn_nerf.train import TrainLoop
    from oracle import render as OR

    def mk(levels):
        return InstantNGPModel(table_sizes=[2 ** 14] * levels, grid_sizes=[2 ** (4 + i // 2) for i in range(levels)],
                               bbox_min=(-1.0,) * 3, bbox_max=(1.0,) * 3)

    n, tc, tf = 256, 64, 128
    loop = TrainLoop(mk(6), mk(16), init_rng=4, lr=1e-2, coarse_ts=tc, fine_ts=tf)
    gen = torch.Generator().manual_seed(0)
    for mdl, sl in ((loop.coarse, loop._slices(loop.flat)[0]), (loop.fine, loop._slices(loop.flat)[1])):
        nt = mdl.encoding().num_table_floats()
        sl[:nt] = ((torch.rand(nt, generator=gen) * 2 - 1) * 0.5).cuda()
        # density = exp(logit): lift the logit so that some rays are opaque and compositing matters
        off = nt
        dims = mdl.dense_dims()
        off += dims[0][0] * dims[0][1] + dims[0][1] + dims[1][0] * dims[1][1]
        sl[off] += 2.0  # Dense_1 bias[0]
    loop._params_changed()
    o = torch.randn(n, 3, generator=gen)
    o = 4 * o / o.norm(dim=-1, keepdim=True)
    dd = -o + (torch.rand(n, 3, generator=gen) - 0.5) * 0.6
    dd = dd / dd.norm(dim=-1, keepdim=True)
    rays = torch.stack([o, dd], 1).float().contiguous()
    p = loop.state.params
    renderer = NeRFRenderer(coarse=loop.coarse, fine=loop.fine, coarse_params=p["coarse"], fine_params=p["fine"],
                            background=p["background"], bbox_min=(-1.0,) * 3, bbox_max=(1.0,) * 3, coarse_ts=tc,
                            fine_ts=tf)
    key = Key(321)
    out = renderer.render_rays(key, rays.cuda())
    ck, fk = split(key, 2)
    uc = torch.from_numpy(philox.ray_uniforms(ck.seed, 0, 0, n, tc)).double()
    uf = torch.from_numpy(philox.ray_uniforms(fk.seed, 1, 0, n, tf)).double()
    cf, ff, bg = [t.cpu().double() for t in loop._slices(loop.flat)]

    def make_fn(model, fl):
        return lambda x, d: ON.ngp_model(fl, x, d, model.table_sizes, model.grid_sizes, (-1.0,) * 3, (1.0,) * 3)

    ref = OR.render_hierarchy(make_fn(loop.coarse, cf), make_fn(loop.fine, ff), bg, torch.tensor([-1.0] * 3, dtype=F64),
                              torch.tensor([1.0] * 3, dtype=F64), rays.double(), tc, tf, uc, uf)
    for lvl in ("coarse", "fine"):
        err = (out[lvl]["outputs"].cpu().double() - ref[lvl]["outputs"]).abs().max().item()
        aerr = (out[lvl]["alphas"].cpu().double() - ref[lvl]["alphas"]).abs().max().item()
        print(f"ngp renderer {lvl}: rgb max|d| vs exact oracle {err:.2e}, alpha {aerr:.2e}")
        assert err < 1e-3 and aerr < 1e-3
    alpha = ref["fine"]["alphas"]
    assert alpha.max() > 0.5, "the test scene must have opaque rays"


@pytest.mark.parametrize("levels,m", [(6, 2500), (16, 4133), (3, 31)])
def test_ngp_mlp_bwd_level_absmax(levels, m):
    """
    lnrf_ngp_mlp_bwd's level_absmax output is exactly max |g_enc_t| over each level's two rows (a max has no rounding),
    and the scatter fed with it equals the scatter that finds the bound itself (same fixed-point scale).
    """
    import ctypes

    from learn_nerf import _lib as L
    from learn_nerf import ops

    model, params, flat = make_model(levels, 2 ** 12, precision="bf16")
    x, d, gen = points(m, seed=levels)
    _, _, _, ctx = model.forward_points(flat, x.cuda(), d.cuda(), save=True)
    desc = model._mlp_desc()
    lf = desc.enc_dim
    scratch = torch.empty(L.lib().lnrf_ngp_mlp_scratch_bytes(ctypes.byref(desc), m), dtype=torch.uint8, device="cuda")
    g_enc_t = torch.empty((lf, m), device="cuda")
    lmax = torch.zeros(lf // 2, device="cuda")
    grad = torch.zeros_like(flat)
    g_d = torch.randn(m, generator=gen).float().cuda()
    g_c = torch.randn(m, 3, generator=gen).float().cuda()
    L.check(L.lib().lnrf_ngp_mlp_bwd(ctypes.byref(desc), L.ptr(ctx["packed"], torch.uint8), L.ptr(ctx["enc_t"]),
                                     L.ptr(ctx["d"]), L.ptr(g_d), L.ptr(g_c), m, L.ptr(scratch, torch.uint8),
                                     L.ptr(g_enc_t), L.ptr(lmax), L.ptr(grad), L.stream()), "ngp_mlp_bwd")
    want = g_enc_t.abs().reshape(lf // 2, 2 * m).max(dim=1).values
    assert torch.equal(lmax, want), (lmax, want)
    enc = model.encoding()
    a = torch.zeros(enc.num_table_floats(), device="cuda")
    b = torch.zeros_like(a)
    ops.hashgrid_bwd(enc.desc(), ctx["x"], g_enc_t, a, level_absmax=lmax)
    ops.hashgrid_bwd(enc.desc(), ctx["x"], g_enc_t, b)
    assert torch.equal(a, b) or (a - b).abs().max().item() <= 1e-6 * b.abs().max().item()


def test_ngp_fused_unsupported_shape_uses_dense_path():
    """hidden_dim 32 has no fused kernel: precision="bf16" then means bf16 operands on the generic dense path."""
    from oracle.model import bf16_round

    model, params, flat = make_model(4, 2 ** 10, hidden=32, precision="bf16")
    assert not model._use_fused()
    x, d, _ = points(100)
    dens, rgb, _ = model.apply(dict(params=params), x.cuda(), d.cuda())
    rd, rr, _ = ON.ngp_model(flat.cpu().float(), x, d, model.table_sizes, model.grid_sizes, BMIN, BMAX, hidden_dim=32,
                             operand_round=bf16_round)
    assert (rgb.cpu() - rr).abs().max().item() < 2e-3
    model32, params32, flat32 = make_model(4, 2 ** 10, hidden=32, precision="fp32")
    _, rgb32, _ = model32.apply(dict(params=params32), x.cuda(), d.cuda())
    ed, er, _ = ON.ngp_model(flat32.cpu().double(), x.double(), d.double(), model.table_sizes, model.grid_sizes, BMIN,
                             BMAX, hidden_dim=32)
    assert (rgb32.cpu().double() - er).abs().max().item() < 1e-4


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_ngp_train_step_matches_oracle(precision):
    from learn_nerf.instant_ngp import InstantNGPModel
    from learn_nerf.rng import Key, split
    from learn_nerf.train import TrainLoop
    from oracle.model import bf16_round

    rnd = bf16_round if precision == "bf16" else None

    def mk(levels):
        return InstantNGPModel(table_sizes=[2 ** 12] * levels, grid_sizes=[2 ** (4 + i // 2) for i in range(levels)],
                               bbox_min=(-1.0,) * 3, bbox_max=(1.0,) * 3, precision=precision)

    n, tc, tf, lr = 128, 16, 32, 1e-2
    loop = TrainLoop(mk(3), mk(5), init_rng=4, lr=lr, coarse_ts=tc, fine_ts=tf, adam_eps=1e-15, adam_b1=0.9,
                     adam_b2=0.99)  # scripts/train_nerf.py:161
    gen = torch.Generator().manual_seed(0)
    for mdl, sl in ((loop.coarse, loop._slices(loop.flat)[0]), (loop.fine, loop._slices(loop.flat)[1])):
        nt = mdl.encoding().num_table_floats()
        sl[:nt] = ((torch.rand(nt, generator=gen) * 2 - 1) * 0.5).cuda()
    o = torch.randn(n, 3, generator=gen)
    o = 4 * o / o.norm(dim=-1, keepdim=True)
    dd = -o + (torch.rand(n, 3, generator=gen) - 0.5) * 0.6
    dd = dd / dd.norm(dim=-1, keepdim=True)
    batch = torch.stack([o, dd, torch.rand(n, 3, generator=gen) * 2 - 1], 1).float().contiguous()
    cf, ff, bg = [t.cpu().double() for t in loop._slices(loop.flat)]
    log = loop.step_fn((-1.0,) * 3, (1.0,) * 3)(77, batch.cuda())
    rk, _ = split(Key(77), 2)
    ck, fk = split(rk, 2)
    uc = torch.from_numpy(philox.ray_uniforms(ck.seed, 0, 0, n, tc)).double()
    uf = torch.from_numpy(philox.ray_uniforms(fk.seed, 1, 0, n, tf)).double()

    def make_fn(model, fl):
        return lambda x, d: ON.ngp_model(fl, x, d, model.table_sizes, model.grid_sizes, (-1.0,) * 3, (1.0,) * 3,
                                         operand_round=rnd)

    params = [cf.clone().requires_grad_(True), ff.clone().requires_grad_(True), bg.clone().requires_grad_(True)]
    total, ld, _ = OT.losses(make_fn(loop.coarse, params[0]), make_fn(loop.fine, params[1]), params[2],
                             torch.tensor([-1.0] * 3, dtype=F64), torch.tensor([1.0] * 3, dtype=F64), batch.double(),
                             tc, tf, uc, uf)
    grads = torch.autograd.grad(total, params)
    ref_grad = torch.cat([g.reshape(-1) for g in grads])
    rel = ((loop.grad.cpu().double() - ref_grad).norm() / ref_grad.norm()).item()
    print(f"ngp step: coarse {float(log['coarse']):.6f}/{float(ld['coarse']):.6f} fine {float(log['fine']):.6f}/"
          f"{float(ld['fine']):.6f} grad rel err {rel:.2e}")
    loss_tol, grad_tol = (1e-5, 5e-3) if precision == "fp32" else (2e-4, 3e-2)
    assert abs(float(log["coarse"]) - float(ld["coarse"])) < loss_tol
    assert abs(float(log["fine"]) - float(ld["fine"])) < loss_tol
    assert rel < grad_tol
    if precision == "bf16":
        return  # the Adam update itself is covered by the fp32 case
    new = [OT.adam_update(p.detach(), g, torch.zeros_like(g), torch.zeros_like(g), 1, lr, 0.9, 0.99, 1e-15)[0]
           for p, g in zip(params, grads)]
    upd = (loop.flat.cpu().double() - torch.cat([t.reshape(-1) for t in new])).abs()
    # entries with (numerically) zero gradient move by +-lr under eps = 1e-15 depending on rounding noise
    big = ref_grad.abs() > 1e-9
    assert upd[big].max().item() < 1e-4


@pytest.mark.parametrize("levels,table", [(4, 2 ** 10), (8, 2 ** 12)])
def test_ngp_ref_nerf_forward_backward(levels, table):
    """InstantNGPRefNERFModel (instant_ngp.py:57-89) incl. the second-order term, vs float64 autograd."""
    from learn_nerf.instant_ngp import InstantNGPRefNERFModel

    grids = [2 ** (3 + i // 2) for i in range(levels)]
    model = InstantNGPRefNERFModel(sh_degree=4, table_sizes=[table] * levels, grid_sizes=grids, bbox_min=BMIN,
                                   bbox_max=BMAX, precision="fp32")
    params = model.init(dict(params=5))["params"]
    flat = model.flat(params)
    nt = model.encoding().num_table_floats()
    gen = torch.Generator().manual_seed(6)
    flat[:nt] = ((torch.rand(nt, generator=gen) * 2 - 1) * 0.5).cuda()
    m = 1200
    lo, hi = torch.tensor(BMIN), torch.tensor(BMAX)
    x = (torch.rand(m, 3, generator=gen) * (hi - lo) * 0.98 + lo + 0.01 * (hi - lo)).float().contiguous()
    d = torch.randn(m, 3, generator=gen)
    d = (d / d.norm(dim=-1, keepdim=True)).float().contiguous()
    assert model.num_params() == nt + sum(i * o + o for i, o in ON.ngp_ref_spec(model.table_sizes, grids)[1])
    dens, rgb, aux = model.apply(dict(params=params), x.cuda(), d.cuda())
    f64 = flat.cpu().double().requires_grad_(True)
    rd, rr, raux = ON.ngp_ref_nerf_model(f64, x.double(), d.double(), model.table_sizes, grids, BMIN, BMAX)
    assert (rgb.cpu().double() - rr).abs().max().item() < 1e-4
    assert ((dens.cpu().double() - rd).abs() / (1 + rd.abs())).max().item() < 1e-4
    for k in aux:
        assert (aux[k].cpu().double() - raux[k]).abs().max().item() < 5e-3, k
    g_d = torch.randn(m, generator=gen).float()
    g_c = torch.randn(m, 3, generator=gen).float()
    g_a = {"normal_mse": torch.rand(m, generator=gen).float(), "neg_normal": torch.rand(m, generator=gen).float()}
    loss = (rd[:, 0] * g_d.double()).sum() + (rr * g_c.double()).sum() + sum((raux[k] * g_a[k].double()).sum() for k in g_a)
    (g_ref,) = torch.autograd.grad(loss, f64)
    _, _, _, ctx = model.forward_points(flat, x.cuda(), d.cuda(), save=True)
    grad = torch.zeros_like(flat)
    model.backward(ctx, g_d.cuda(), g_c.cuda(), {k: v.cuda() for k, v in g_a.items()}, grad)
    got = grad.cpu().double()
    for name, a, b in (("tables", got[:nt], g_ref[:nt]), ("mlp", got[nt:], g_ref[nt:])):
        rel = ((a - b).norm() / b.norm()).item()
        print(f"ngp-ref L={levels} {name}: rel L2 err {rel:.2e}")
        assert rel < 5e-3, (name, rel)


@pytest.mark.parametrize("clustered", [False, True])
def test_bucketed_scatter_matches_direct_path(clustered):
    """Levels with 4..128 table slices (dense 32^3 / 48^3 and hashed 128^3 / 512^3 here) use the bucketed
    reduction; same result as the LDS-sliced / atomic kernels, for value weights and for derivative weights.
    clustered: all samples in 2 % of the box, so the dense levels overflow their bucket capacity and exercise
    the atomic fallback of the bin kernel."""
    import ctypes

    from learn_nerf import _lib as L
    from learn_nerf import ops
    from learn_nerf.instant_ngp import MultiresHashTableEncoding

    grids = [16, 32, 48, 128, 512]
    enc = MultiresHashTableEncoding([2 ** 17] * len(grids), grids, BMIN, BMAX, 2, True)
    desc = enc.desc()
    m = 20000
    x, _, gen = points(m, seed=11)
    if clustered:
        lo, hi = torch.tensor(BMIN), torch.tensor(BMAX)
        x = (lo + (hi - lo) * (0.4 + 0.02 * torch.rand(m, 3, generator=gen))).float().contiguous()
    g = torch.randn(2 * len(grids), m, generator=gen).float().cuda()
    u = torch.randn(m, 3, generator=gen).float().cuda()
    assert L.lib().lnrf_hashgrid_bwd_scratch_bytes(ctypes.byref(desc), m) > 4 * 20000 * 8 * 12
    for uu in (None, u):
        a = torch.zeros(enc.num_table_floats(), device="cuda")
        b = torch.zeros_like(a)
        ops.hashgrid_bwd(desc, x.cuda(), g, a, u=uu)  # bucketed for all but the 16^3 level
        L.check(L.lib().lnrf_hashgrid_bwd_bucketed(ctypes.byref(desc), L.ptr(x.cuda()), L.ptr(uu), m, L.ptr(g),
                                                   None, L.ptr(b), None, 0, L.stream()))  # no scratch: direct kernels
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() < 1e-5 * scale
        assert a.abs().sum().item() > 0
        off = 0
        for r in enc.rows():  # every level contributes
            assert a[off:off + 2 * r].abs().sum().item() > 0
            off += 2 * r


def test_ngp_ref_nerf_bf16_dense_path():
    """InstantNGPRefNERFModel with precision="bf16" (LNRF_DENSE_BF16 operands) vs the operand-rounding oracle."""
    from learn_nerf.instant_ngp import InstantNGPRefNERFModel
    from oracle.model import bf16_round

    levels, table = 4, 2 ** 10
    grids = [2 ** (3 + i // 2) for i in range(levels)]
    model = InstantNGPRefNERFModel(sh_degree=4, table_sizes=[table] * levels, grid_sizes=grids, bbox_min=BMIN,
                                   bbox_max=BMAX, precision="bf16")
    params = model.init(dict(params=5))["params"]
    flat = model.flat(params)
    nt = model.encoding().num_table_floats()
    gen = torch.Generator().manual_seed(6)
    flat[:nt] = ((torch.rand(nt, generator=gen) * 2 - 1) * 0.5).cuda()
    m = 800
    lo, hi = torch.tensor(BMIN), torch.tensor(BMAX)
    x = (torch.rand(m, 3, generator=gen) * (hi - lo) * 0.98 + lo + 0.01 * (hi - lo)).float().contiguous()
    d = torch.randn(m, 3, generator=gen)
    d = (d / d.norm(dim=-1, keepdim=True)).float().contiguous()
    dens, rgb, aux = model.apply(dict(params=params), x.cuda(), d.cuda())
    rd, rr, raux = ON.ngp_ref_nerf_model(flat.cpu().float(), x, d, model.table_sizes, grids, BMIN, BMAX, sh_degree=4,
                                         operand_round=bf16_round)
    ed, er, _ = ON.ngp_ref_nerf_model(flat.cpu().double(), x.double(), d.double(), model.table_sizes, grids, BMIN, BMAX,
                                      sh_degree=4)
    e_rgb = (rgb.cpu() - rr.detach()).abs().max().item()
    x_rgb = (rgb.cpu().double() - er.detach()).abs().max().item()
    print(f"ngp-ref bf16: rgb {e_rgb:.2e} vs bf16-operand oracle, {x_rgb:.2e} vs exact")
    assert e_rgb < 4e-3 and x_rgb < 5e-2
    assert ((dens.reshape(-1).cpu() - rd.detach()[:, 0]).abs() / (1 + rd.detach()[:, 0].abs())).max().item() < 4e-3


def _scatter_case(m=3000, seed=3, decades=12):
    """Points on the lattice k / 1024 of the unit box: cell coordinates (G - 1) * x are exact in fp32, so the device's
    trilinear weights equal the float64 oracle's to a few 1e-8 relative and the comparison is about the SUMMATION."""
    from learn_nerf.instant_ngp import MultiresHashTableEncoding

    ts, gs = [2 ** 14] * 4, [16, 32, 64, 64]  # one dense level, three hashed; about as many contributions as entries
    lo, hi = (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)
    enc = MultiresHashTableEncoding(ts, gs, lo, hi, 2, False)
    gen = torch.Generator().manual_seed(seed)
    x = (torch.randint(0, 1025, (m, 3), generator=gen).float() / 1024.0).contiguous()
    # per-sample magnitudes spread over `decades` decades: the entries of a table then see sums from 1e-12 to 1
    mag = 10.0 ** (-decades * torch.rand(m, 1, generator=gen))
    g_enc = (torch.randn(m, 2 * len(gs), generator=gen) * mag).float()

    def scatter64(g):
        t64 = torch.zeros(enc.num_table_floats(), dtype=F64, requires_grad=True)
        off, feats = 0, []
        for r, t, gsz in zip(enc.rows(), ts, gs):
            feats.append(ON.hash_table_encoding(x.double(), t64[off:off + 2 * r].reshape(r, 2), gsz, t,
                                                torch.tensor(lo, dtype=F64), torch.tensor(hi, dtype=F64), False))
            off += 2 * r
        (out,) = torch.autograd.grad((torch.cat(feats, 1) * g).sum(), t64)
        return out

    return enc, x, g_enc, scatter64(g_enc.double()), scatter64(g_enc.double().abs())


def test_scatter_without_bound_keeps_small_entries():
    """the plain scatter of the exact-fp32 path (no level bound handed in) must not flush small contributions.
    Gradients spanning 12 decades, checked PER ENTRY against the float64 scatter: every entry within 1e-5 relative + an
    absolute floor of 2^-40 of the level's largest contribution (the 64-bit fixed point of the reduce pass)."""
    from learn_nerf import ops

    enc, x, g_enc, g_ref, g_abs = _scatter_case()
    g_tab = torch.zeros(enc.num_table_floats(), device="cuda")
    ops.hashgrid_bwd(enc.desc(), x.cuda(), g_enc.t().contiguous().cuda(), g_tab)
    got = g_tab.cpu().double()
    floor = g_enc.abs().max().item() * 2.0 ** -40
    err = (got - g_ref).abs()
    # 1e-5 of the entry itself, 1e-6 of the sum of the magnitudes that went into it (cancellation), the fixed-point floor
    bad = err > 1e-5 * g_ref.abs() + 1e-6 * g_abs + floor
    small = (g_ref.abs() > 0) & (g_ref.abs() < 1e-8)
    print(f"entries {int((g_ref != 0).sum())}, of them below 1e-8: {int(small.sum())}; worst relative error "
          f"{(err / g_ref.abs().clamp_min(1e-30))[g_ref.abs() > floor * 1e3].max().item():.2e}")
    assert int(small.sum()) > 100, "the case must contain small entries"
    assert not bad.any(), int(bad.sum())
    # the small entries are really there (not flushed): at least 99 % of them non-zero on the device
    assert (got[small] != 0).float().mean().item() > 0.99


def test_scatter_with_bound_is_quantised_as_documented():
    """With a per-level bound (what lnrf_ngp_mlp_bwd hands over on the fused bf16 path) the contributions are 26-bit fixed
    point of that bound: per entry the error is at most (number of contributions) x 2^-26 x bound, small entries may be
    flushed — the documented quantisation (parity of InstantNGP training on the fused path is unpinned beyond it)."""
    from learn_nerf import ops

    enc, x, g_enc, g_ref, _ = _scatter_case()
    g_t = g_enc.t().contiguous().cuda()
    bound = g_t.abs().reshape(len(enc.grid_sizes), -1).max(dim=1).values.contiguous()
    g_tab = torch.zeros(enc.num_table_floats(), device="cuda")
    ops.hashgrid_bwd(enc.desc(), x.cuda(), g_t, g_tab, level_absmax=bound)
    err = (g_tab.cpu().double() - g_ref).abs().max().item()
    # <= 8 m contributions in total; per entry far fewer: a generous count of 4096 per entry
    assert err < 4096 * 2.0 ** -26 * bound.max().item() + 1e-6 * g_ref.abs().max().item(), err


@pytest.mark.parametrize("with_bound", [False, True])
def test_scatter_propagates_nan_and_inf(with_bound):
    """A NaN or Inf in d loss / d enc must reach the table entries its sample touches (as jax's scatter-add would), not be
    flushed to zero by the fixed-point path; everything else stays finite and correct."""
    from learn_nerf import ops

    enc, x, g_enc, g_ref, _ = _scatter_case(m=2000, decades=2)
    g_enc = g_enc.clone()
    g_enc[17, 0] = float("nan")   # level 0, feature 0 of sample 17
    g_enc[99, 3] = float("inf")   # level 1, feature 1 of sample 99
    g_t = g_enc.t().contiguous().cuda()
    bound = None
    if with_bound:
        bound = torch.nan_to_num(g_t, nan=0.0, posinf=0.0, neginf=0.0).abs().reshape(len(enc.grid_sizes), -1).max(dim=1).values.contiguous()
    g_tab = torch.zeros(enc.num_table_floats(), device="cuda")
    ops.hashgrid_bwd(enc.desc(), x.cuda(), g_t, g_tab, level_absmax=bound)
    got = g_tab.cpu()
    rows = enc.rows()
    lvl0 = got[:2 * rows[0]].reshape(rows[0], 2)
    lvl1 = got[2 * rows[0]:2 * (rows[0] + rows[1])].reshape(rows[1], 2)
    assert torch.isnan(lvl0[:, 0]).sum().item() >= 1 and not torch.isnan(lvl0[:, 1]).any()
    assert (torch.isinf(lvl1[:, 1]) | torch.isnan(lvl1[:, 1])).sum().item() >= 1
    rest = got[2 * (rows[0] + rows[1]):]
    assert torch.isfinite(rest).all()


@pytest.mark.parametrize("levels,m", [(6, 2500), (16, 9000)])
def test_ngp_fused_backward_dense_gradients_are_bit_reproducible(levels, m):
    """The persistent InstantNGP backward stores every workgroup's share of the Dense gradients as a row and
    ngp_wparts_reduce_kernel folds the rows in a fixed order: the Dense part of the gradient is bit-reproducible.  (The
    table part is summed with 64-bit integer LDS atomics per bucket — order-independent — except for buckets that are split
    among workgroups or overflow, which meet in fp32 atomics: reproducible to rounding, not bit for bit.)"""
    model, params, flat = make_model(levels, 2 ** 12, precision="bf16")
    x, d, gen = points(m, seed=levels)
    g_d = torch.randn(m, generator=gen).float().cuda()
    g_c = torch.randn(m, 3, generator=gen).float().cuda()
    nt = model.encoding().num_table_floats()
    grads = []
    for _ in range(3):
        _, _, _, ctx = model.forward_points(flat, x.cuda(), d.cuda(), save=True)
        g = torch.zeros_like(flat)
        model.backward(ctx, g_d, g_c, None, g)
        grads.append(g)
    assert grads[0][nt:].abs().max().item() > 0
    assert torch.equal(grads[0][nt:], grads[1][nt:]) and torch.equal(grads[0][nt:], grads[2][nt:])
    rel = ((grads[0][:nt] - grads[1][:nt]).norm() / grads[0][:nt].norm()).item()
    assert rel < 1e-5
