"""
BASELINE.json full size (4096 rays x 64 coarse + 128 fine samples, fused bf16 kernels): size-independent
properties of the whole step plus an oracle check on a random subset of the rays (same ts, same weights).
"""
import pytest
import torch

from oracle import model as OM
from oracle import render as OR

pytestmark = pytest.mark.gpu
F64 = torch.float64
N, TC, TF = 4096, 64, 128
BMIN, BMAX = (-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)


def test_full_size_step_properties_and_subset_parity():
    from learn_nerf import ops
    from learn_nerf.model import NeRFModel
    from learn_nerf.render import NeRFRenderer
    from learn_nerf.rng import Key
    from learn_nerf.train import TrainLoop

    loop = TrainLoop(NeRFModel(), NeRFModel(), init_rng=0, lr=1e-4, coarse_ts=TC, fine_ts=TF)
    for name in ("coarse", "fine"):
        loop.state.params[name]["Dense_9"]["kernel"].mul_(6.0)
        loop.state.params[name]["Dense_9"]["bias"].add_(1.5)
    gen = torch.Generator().manual_seed(0)
    o = torch.randn(N, 3, generator=gen)
    o = 4 * o / o.norm(dim=-1, keepdim=True)
    d = -o + (torch.rand(N, 3, generator=gen) - 0.5) * 0.6
    d[:200] = torch.randn(200, 3, generator=gen)
    d = d / d.norm(dim=-1, keepdim=True)
    batch = torch.stack([o, d, torch.rand(N, 3, generator=gen) * 2 - 1], 1).float().contiguous().cuda()
    p = loop.state.params
    renderer = NeRFRenderer(coarse=loop.coarse, fine=loop.fine, coarse_params=p["coarse"], fine_params=p["fine"],
                            background=p["background"], bbox_min=BMIN, bbox_max=BMAX, coarse_ts=TC, fine_ts=TF)
    out = renderer.render_rays(Key(5), batch)
    t_min, t_max, mask = renderer.t_range(batch)
    fine, coarse = out["fine"], out["coarse"]
    assert fine["rgbs"].shape == (N, TC + TF, 3) and fine["densities"].shape == (N, TC + TF)
    for lvl in (coarse, fine):
        assert torch.isfinite(lvl["outputs"]).all() and torch.isfinite(lvl["densities"]).all()
        assert (lvl["densities"] >= 0).all() and (lvl["rgbs"].abs() <= 1).all()
        a = lvl["alphas"][:, 0]
        assert (a >= 0).all() and (a <= 1 + 1e-6).all()
        assert (a[~mask] == 0).all()
        assert torch.equal(lvl["outputs"][~mask], p["background"].expand(int((~mask).sum()), 3))
    assert 0 < int((~mask).sum()) < N
    # identical call -> identical result (the forward path has no atomics)
    out2 = renderer.render_rays(Key(5), batch)
    assert torch.equal(out2["fine"]["outputs"], fine["outputs"])
    # termination probabilities of the fine pass sum to 1 for every ray (render.py:270-287)
    ts_c = ops.ray_aabb_stratified(batch, BMIN, BMAX, TC, seed=Key(5).split(2)[0].seed, stream_id=0)[3]
    ts_f = ops.fine_sample(ts_c, t_min, t_max, coarse["densities"], TF, seed=Key(5).split(2)[1].seed, stream_id=1)
    assert (ts_f[:, 1:] >= ts_f[:, :-1]).all()
    probs = ops.termination_probs(ts_f, t_min, t_max, fine["densities"])
    assert (probs.sum(1) - 1).abs().max().item() < 1e-4
    # oracle on a random subset of rays: same fine ts, same weights (bf16-operand oracle)
    sel = torch.randperm(N, generator=gen)[:48]
    cf, ff, bg = [t.cpu().double() for t in loop._slices(loop.flat)]
    rays_s = batch[sel.cuda(), :2].cpu().double()
    bbox = torch.tensor([BMIN, BMAX], dtype=F64)
    tmin_s, tmax_s, mask_s = OR.ray_t_range(bbox, rays_s)
    samples = OR.RaySamples(tmin_s, tmax_s, mask_s, ts_f[sel.cuda()].cpu().double())
    ref, _ = OR.render_rays(OM.make_nerf_fn(ff, OM.bf16_round), bg, rays_s, samples)
    err = (fine["outputs"][sel.cuda()].cpu().double() - ref["outputs"]).abs().max().item()
    print(f"full size: subset max |rgb - oracle| = {err:.2e}")
    assert err < 4e-3
    # one training step at full size: finite losses, parameters move, loss decreases over a few steps
    step = loop.step_fn(BMIN, BMAX)
    first = step(Key(1), batch)
    for i in range(2, 12):
        last = step(Key(i), batch)
    assert all(torch.isfinite(v) for v in last.values())
    assert float(last["fine"]) < float(first["fine"])
