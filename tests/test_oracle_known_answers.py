"""
Known-answer identities that pin the oracle.  The reference ships no golden
vectors for this path ("parity unpinned"), so these analytic identities, which follow from the
reference source, are what any correct restatement must satisfy.
"""
import math

import numpy as np
import torch

from oracle import model as OM
from oracle import philox
from oracle import render as OR

torch.manual_seed(0)
F64 = torch.float64


def _rays(n, gen):
    o = torch.randn(n, 3, generator=gen, dtype=F64)
    o = 4 * o / o.norm(dim=-1, keepdim=True)
    d = -o + (torch.rand(n, 3, generator=gen, dtype=F64) - 0.5) * 0.6
    d = d / d.norm(dim=-1, keepdim=True)
    return torch.stack([o, d], dim=1)


def _samples(n=32, t=64, seed=1):
    gen = torch.Generator().manual_seed(seed)
    rays = _rays(n, gen)
    bbox = torch.tensor([[-1.0, -1, -1], [1, 1, 1]], dtype=F64)
    t_min, t_max, mask = OR.ray_t_range(bbox, rays)
    u = torch.rand(n, t, generator=gen, dtype=F64)
    return rays, OR.RaySamples(t_min, t_max, mask, OR.stratified_ts(t_min, t_max, t, u)), gen


def test_deltas_sum_to_range():  # identity (1), render.py:259-268
    _, s, _ = _samples()
    assert torch.allclose(s.deltas().sum(1), s.t_max - s.t_min, atol=1e-12)


def test_termination_probs_sum_to_one():  # identity (2), render.py:270-287
    _, s, gen = _samples()
    dens = torch.rand(s.ts.shape, generator=gen, dtype=F64) * 5
    p = s.termination_probs(dens)
    assert p.shape[1] == s.ts.shape[1] + 1
    assert torch.allclose(p.sum(1), torch.ones(p.shape[0], dtype=F64), atol=1e-12)


def test_constant_density_closed_form():  # identity (3)
    _, s, _ = _samples()
    sigma0 = 0.7
    dens = torch.full(s.ts.shape, sigma0, dtype=F64)
    c = torch.tensor([0.3, -0.2, 0.9], dtype=F64)
    bg = torch.tensor([-1.0, -1, -1], dtype=F64)
    rgbs = c[None, None].expand(*s.ts.shape, 3)
    alpha = 1 - torch.exp(-sigma0 * (s.t_max - s.t_min))
    out = s.render_rays(dens, rgbs, bg)
    expect = alpha[:, None] * c + (1 - alpha[:, None]) * bg
    m = s.mask
    assert m.any()
    assert torch.allclose(out[m], expect[m], atol=1e-12)
    assert torch.allclose(s.render_alpha(dens)[m, 0], alpha[m], atol=1e-12)


def test_masked_rays_give_background():  # identity (4), render.py:174-176,190,387-389
    bbox = torch.tensor([[-1.0, -1, -1], [1, 1, 1]], dtype=F64)
    rays = torch.tensor([[[5.0, 5, 5], [0, 0, 1.0]], [[0.0, 0, -3], [0, 0, 1.0]]], dtype=F64)
    t_min, t_max, mask = OR.ray_t_range(bbox, rays)
    assert mask.tolist() == [False, True]
    assert t_min[0] == 0 and abs(t_max[0] - 1e-3) < 1e-15
    assert abs(t_min[1] - 2) < 1e-6 and abs(t_max[1] - 4) < 1e-6
    u = torch.full((2, 8), 0.5, dtype=F64)
    s = OR.RaySamples(t_min, t_max, mask, OR.stratified_ts(t_min, t_max, 8, u))
    dens = torch.ones(2, 8, dtype=F64)
    bg = torch.tensor([0.1, 0.2, 0.3], dtype=F64)
    out = s.render_rays(dens, torch.zeros(2, 8, 3, dtype=F64), bg)
    assert torch.equal(out[0], bg)
    assert s.render_alpha(dens)[0, 0] == 0


def test_sinusoidal_emb_known_values():  # identity (5), model.py:72-77
    z = OM.sinusoidal_emb(torch.zeros(2, 3, dtype=F64), 4)
    assert z.shape == (2, 24)
    assert torch.equal(z[0].reshape(3, 8), torch.tensor([0.0] * 4 + [1.0] * 4, dtype=F64).expand(3, 8))
    x = torch.tensor([[0.3, -1.2, 2.0]], dtype=F64)
    e = OM.sinusoidal_emb(x, 1)
    assert torch.allclose(e, torch.stack([x.sin(), x.cos()], -1).reshape(1, 6))
    e10 = OM.sinusoidal_emb(x, 10)
    # coordinate outermost, then sin block, then cos block; no pi factor, raw coordinate absent
    assert abs(e10[0, 20 + 3] - math.sin(8 * -1.2)) < 1e-12
    assert abs(e10[0, 40 + 10 + 9] - math.cos(512 * 2.0)) < 1e-9


def test_fine_sampling_uniform_weights_is_affine():  # identity (8), render.py:253-255
    _, s, gen = _samples(n=16, t=64)
    # equal-width coarse bins with equal weights -> CDF is linear in t -> u maps affinely
    n, t = s.ts.shape
    u0 = torch.full((n, t), 0.5, dtype=F64)
    s = OR.RaySamples(s.t_min, s.t_max, s.mask, OR.stratified_ts(s.t_min, s.t_max, t, u0))
    dens = torch.zeros(n, t, dtype=F64)  # probs 0 -> w = eps everywhere (uniform)
    uf = torch.rand(n, 128, generator=gen, dtype=F64)
    fine = s.fine_sampling(128, uf, dens, combine=False)
    up = (torch.arange(128, dtype=F64)[None] + uf) / 128
    expect = s.t_min[:, None] + up * (s.t_max - s.t_min)[:, None]
    assert torch.allclose(fine.ts, expect, atol=1e-9)
    comb = s.fine_sampling(128, uf, dens, combine=True).ts
    assert comb.shape == (n, 192)
    assert (comb[:, 1:] >= comb[:, :-1]).all()
    # coarse ts are a subset of the combined samples
    for i in range(n):
        assert np.isin(s.ts[i].numpy(), comb[i].numpy()).all()


def test_fine_samples_zero_keeps_coarse():  # identity (9)
    _, s, _ = _samples(n=4, t=16)
    dens = torch.rand(4, 16, dtype=F64)
    fine = s.fine_sampling(0, torch.zeros(4, 0, dtype=F64), dens)
    assert torch.equal(fine.ts, s.ts)


def test_interp_matches_numpy():
    gen = torch.Generator().manual_seed(3)
    w = torch.rand(5, 20, generator=gen, dtype=F64)
    w[:, 5:9] = 0  # flat segments
    xs = torch.cat([torch.zeros(5, 1, dtype=F64), torch.cumsum(w, 1)], 1)
    xs = xs / xs[:, -1:]
    ys = torch.cumsum(torch.rand(5, 21, generator=gen, dtype=F64), 1)
    x = torch.rand(5, 50, generator=gen, dtype=F64)
    got = OR.interp_rows(x, xs, ys)
    for i in range(5):
        exp = np.interp(x[i].numpy(), xs[i].numpy(), ys[i].numpy())
        assert np.allclose(got[i].numpy(), exp, atol=1e-12)


def test_nerf_param_count_and_shapes():
    dims = OM.nerf_layer_dims()
    assert OM.param_count(dims) == 593_924
    assert dims[0] == (60, 256) and dims[5] == (316, 256) and dims[9] == (256, 1)
    assert dims[10] == (280, 128) and dims[11] == (128, 3)
    assert sum(i * o for i, o in dims) == 591_488  # MACs per evaluation


def test_nerf_mlp_output_ranges_and_grad():
    gen = torch.Generator().manual_seed(0)
    dims = OM.nerf_layer_dims(hidden_dim=32, color_layer_dim=16)
    flat = OM.lecun_normal_init(dims, gen, dtype=F64)
    x = torch.rand(7, 3, generator=gen, dtype=F64) * 2 - 1
    d = torch.randn(7, 3, generator=gen, dtype=F64)
    dens, rgb, aux = OM.nerf_mlp(flat, x, d, hidden_dim=32, color_layer_dim=16)
    assert dens.shape == (7, 1) and rgb.shape == (7, 3) and aux == {}
    assert (dens >= 0).all() and (rgb.abs() < 1).all()

    def f(p):
        a, b, _ = OM.nerf_mlp(p, x, d, hidden_dim=32, color_layer_dim=16)
        return (a.sum() + (b * b).sum())

    p = flat.clone().requires_grad_(True)
    assert torch.autograd.gradcheck(f, (p,), eps=1e-6, atol=1e-5, nondet_tol=0)


def test_philox_known_answer():
    # Random123 known-answer vectors for philox4x32-10
    out = philox.philox4x32_10([0], [0], [0], [0], 0, 0)
    assert [int(o[0]) for o in out] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    out = philox.philox4x32_10([0xFFFFFFFF], [0xFFFFFFFF], [0xFFFFFFFF], [0xFFFFFFFF], 0xFFFFFFFF, 0xFFFFFFFF)
    assert [int(o[0]) for o in out] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    out = philox.philox4x32_10([0x243F6A88], [0x85A308D3], [0x13198A2E], [0x03707344], 0xA4093822, 0x299F31D0)
    assert [int(o[0]) for o in out] == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    u = philox.ray_uniforms(1234, 0, 0, 8, 64)
    assert u.shape == (8, 64) and u.dtype == np.float32
    assert (u >= 0).all() and (u < 1).all() and abs(u.mean() - 0.5) < 0.05
    # ray_offset consistency (data-parallel shards see the same numbers as one big batch)
    assert np.array_equal(philox.ray_uniforms(1234, 0, 3, 5, 64), u[3:8])
