"""Known answers for the Ref-NeRF restatement — CPU."""
import math

import torch

from oracle import ref_nerf as ORF

F64 = torch.float64


def unit(n, seed=0):
    gen = torch.Generator().manual_seed(seed)
    v = torch.randn(n, 3, generator=gen, dtype=F64)
    return v / v.norm(dim=-1, keepdim=True)


def test_sh_constants_on_z_axis():  # ref_nerf.py:174-186
    z = torch.tensor([[0.0, 0.0, 1.0]], dtype=F64)
    out = ORF.spherical_harmonic(8, z)[0]
    assert out.shape == (64,)
    assert abs(out[0] - 0.28209479177387814) < 1e-14
    assert abs(out[2] - 0.48860251190291987) < 1e-14
    assert abs(out[6] - (0.94617469575755997 - 0.31539156525251999)) < 1e-14
    assert abs(out[1]) < 1e-15 and abs(out[3]) < 1e-15


def test_sh_matches_reference_polynomial_table():
    """The one numeric table the reference holds on this path: all 64 polynomials of ref_nerf.py:174-311, evaluated
    from the reference's coefficients (tests/golden/make_sh_table_golden.py) at 14 fixed unit vectors.  This PINS
    the oracle's spherical harmonics (Legendre-recurrence restatement) for every degree, incl. the +-m pairing."""
    import os

    import numpy as np

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sh_table_v1.npz"))
    dirs = torch.from_numpy(g["dirs"])
    assert g["values"].shape == (14, 64) and np.allclose(np.linalg.norm(g["dirs"], axis=1), 1.0)
    for degree in range(1, 9):
        got = ORF.spherical_harmonic(degree, dirs).numpy()
        assert got.shape == (14, degree * degree)
        err = np.abs(got - g["values"][:, :degree * degree]).max()
        assert err < 1e-13, (degree, err)
    # integrated directional encoding: the same values attenuated per level l by exp(-rho l (l + 1) / 2)
    rho = torch.linspace(0.0, 1.5, 14, dtype=F64)[:, None]
    ide = ORF.integrated_directional_encoding(8, dirs, rho).numpy()
    levels = np.concatenate([[l] * (2 * l + 1) for l in range(8)])
    assert np.abs(ide - g["values"] * np.exp(-rho.numpy() * levels * (levels + 1) / 2)).max() < 1e-13


def test_sh_low_degree_closed_forms():
    v = unit(20)
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    out = ORF.spherical_harmonic(4, v)
    a = 0.48860251190291987
    assert torch.allclose(out[:, 1], -a * y) and torch.allclose(out[:, 2], a * z) and torch.allclose(out[:, 3], -a * x)
    b = 1.0925484305920792
    assert torch.allclose(out[:, 4], b * x * y) and torch.allclose(out[:, 5], -b * y * z)
    assert torch.allclose(out[:, 7], -b * x * z) and torch.allclose(out[:, 8], 0.5 * b * (x * x - y * y))
    assert torch.allclose(out[:, 9], 0.59004358992664352 * y * (-3 * x * x + y * y))
    assert torch.allclose(out[:, 10], 2.8906114426405538 * x * y * z)
    assert torch.allclose(out[:, 12], 0.3731763325901154 * z * (5 * z * z - 3))
    assert torch.allclose(out[:, 15], 0.59004358992664352 * x * (-x * x + 3 * y * y))


def test_sh_orthonormal_by_quadrature():
    # Gauss-Legendre in cos(theta) x uniform in phi integrates degree <= 15 products exactly enough
    import numpy as np

    nodes, weights = np.polynomial.legendre.leggauss(24)
    phi = np.arange(48) * (2 * np.pi / 48)
    ct, ph = np.meshgrid(nodes, phi, indexing="ij")
    st = np.sqrt(1 - ct ** 2)
    pts = torch.tensor(np.stack([st * np.cos(ph), st * np.sin(ph), ct], -1).reshape(-1, 3), dtype=F64)
    w = torch.tensor(np.repeat(weights, 48) * (2 * np.pi / 48), dtype=F64)
    Y = ORF.spherical_harmonic(8, pts)
    gram = Y.T @ (Y * w[:, None])
    assert torch.allclose(gram, torch.eye(64, dtype=F64), atol=1e-10)


def test_ide_roughness_zero_is_sh_and_attenuation():
    v = unit(7, seed=1)
    z = torch.zeros(7, 1, dtype=F64)
    assert torch.allclose(ORF.integrated_directional_encoding(4, v, z), ORF.spherical_harmonic(4, v))
    r = torch.full((7, 1), 0.3, dtype=F64)
    ide = ORF.integrated_directional_encoding(3, v, r)
    sh = ORF.spherical_harmonic(3, v)
    assert torch.allclose(ide[:, 0], sh[:, 0]) and torch.allclose(ide[:, 1:4], sh[:, 1:4] * math.exp(-0.3))
    assert torch.allclose(ide[:, 4:9], sh[:, 4:9] * math.exp(-0.9))


def test_srgb_and_leaky_clip():
    c = torch.tensor([-0.5, 0.0, 0.002, 0.0031308, 0.5, 1.0, 1.7], dtype=F64, requires_grad=True)
    s = ORF.linear_rgb_to_srgb(c)
    assert abs(s[2] - 12.92 * 0.002) < 1e-15 and abs(s[5] - 1.0) < 1e-12
    lc = ORF._leaky_clip(c)
    assert torch.equal(lc.detach(), torch.clamp(c.detach(), 0, 1))
    (g,) = torch.autograd.grad(lc.sum(), c)
    assert torch.equal(g, torch.ones_like(g))  # identity gradient outside the bounds too


def test_ref_nerf_model_shapes_and_second_order_gradcheck():
    gen = torch.Generator().manual_seed(0)
    kw = dict(hidden_dim=16, color_layer_dim=8, x_freqs=3, sh_degree=3, input_layers=2, mid_layers=2)
    dims = ORF.ref_nerf_layer_dims(**kw)
    n = sum(i * o + o for i, o in dims)
    flat = torch.randn(n, generator=gen, dtype=F64) * 0.3
    x = torch.rand(4, 3, generator=gen, dtype=F64) * 2 - 1
    d = unit(4, seed=2)
    dens, rgb, aux = ORF.ref_nerf_model(flat, x, d, **kw)
    assert dens.shape == (4, 1) and rgb.shape == (4, 3) and set(aux) == {"normal_mse", "neg_normal"}
    assert (dens > 0).all() and (rgb >= -1).all() and (rgb <= 1).all() and (aux["normal_mse"] >= 0).all()
    assert ORF.ref_nerf_layer_dims() [9] == (273, 128)

    def f(p):
        a, b, c = ORF.ref_nerf_model(p, x, d, **kw)
        return a.sum() + (b * b).sum() + c["normal_mse"].sum() + c["neg_normal"].sum()

    p = flat.clone().requires_grad_(True)
    assert torch.autograd.gradcheck(f, (p,), eps=1e-6, atol=1e-4, nondet_tol=0)
