"""Known-answer identities for the hash-grid restatement — CPU."""
import torch

from oracle import instant_ngp as ON

F64 = torch.float64
BMIN, BMAX = (-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)


def test_hash_known_answers():
    T = 2 ** 18
    c = torch.tensor([[1, 0, 0], [0, 1, 0], [0, 0, 1], [2047, 2047, 2047]])
    idx = ON.hash_table_lookup_index(c, T)
    assert idx[0] == 1 and idx[1] == 19349663 % T and idx[2] == 83492791 % T
    expect = (2047 ^ ((19349663 * 2047) & 0xFFFFFFFF) ^ ((83492791 * 2047) & 0xFFFFFFFF)) % T  # uint32 wrap
    assert idx[3] == expect


def test_dense_vs_hashed_switch_is_strict():
    assert ON.level_rows(64, 2 ** 18) == (64 ** 3, False)  # 64^3 == 2^18 is NOT > T (instant_ngp.py:178)
    assert ON.level_rows(128, 2 ** 18) == (2 ** 18, True)
    assert ON.level_rows(16, 2 ** 18) == (4096, False)


def test_reference_cli_table_sizes():
    # scripts/train_nerf.py:150-160: coarse 6 levels, fine 16 levels, T = 2^18, grids 2^(4 + i//2)
    for levels, floats in ((6, 1_196_032), (16, 6_438_912)):
        rows, dims = ON.ngp_spec([2 ** 18] * levels, [2 ** (4 + i // 2) for i in range(levels)])
        assert sum(r * 2 for r in rows) == floats
        assert dims[0] == (2 * levels, 64) and dims[1] == (64, 16) and dims[2] == (40, 64) and dims[-1] == (64, 3)
    assert sum(i * o + o for i, o in dims) == 10_131  # fine MLP parameters


def test_partition_of_unity_and_vertex_values():
    gen = torch.Generator().manual_seed(0)
    G, T = 8, 2 ** 12
    rows, _ = ON.level_rows(G, T)
    x = torch.rand(50, 3, generator=gen, dtype=F64) * 2 - 1
    ones = torch.ones(rows, 2, dtype=F64)
    bmin, bmax = torch.tensor(BMIN, dtype=F64), torch.tensor(BMAX, dtype=F64)
    for smooth in (False, True):
        e = ON.hash_table_encoding(x, ones, G, T, bmin, bmax, smooth)
        assert torch.allclose(e, torch.ones(50, 2, dtype=F64), atol=1e-12)  # trilinear weights sum to 1
    table = torch.rand(rows, 2, generator=gen, dtype=F64)
    v = torch.tensor([[3, 5, 2]])
    xv = v.double() / (G - 1) * 2 - 1  # exactly on a grid vertex
    e = ON.hash_table_encoding(xv, table, G, T, bmin, bmax, False)
    assert torch.allclose(e[0], table[3 + G * (5 + G * 2)], atol=1e-12)
    # outside the box: clipped to the boundary
    far = torch.tensor([[5.0, -7.0, 0.3]], dtype=F64)
    near = torch.tensor([[1.0, -1.0, 0.3]], dtype=F64)
    assert torch.allclose(ON.hash_table_encoding(far, table, G, T, bmin, bmax), ON.hash_table_encoding(near, table, G, T, bmin, bmax))


def test_ngp_model_shapes_and_gradcheck():
    gen = torch.Generator().manual_seed(1)
    ts, gs = [64, 64, 64], [3, 4, 6]  # tiny: level 2 hashed (216 > 64)
    n = ON.ngp_param_count(ts, gs, hidden_dim=8, density_dim=4)
    flat = (torch.rand(n, generator=gen, dtype=F64) - 0.5) * 0.5
    x = torch.rand(5, 3, generator=gen, dtype=F64) * 1.8 - 0.9
    d = torch.randn(5, 3, generator=gen, dtype=F64)
    dens, rgb, aux = ON.ngp_model(flat, x, d, ts, gs, BMIN, BMAX, hidden_dim=8, density_dim=4)
    assert dens.shape == (5, 1) and rgb.shape == (5, 3) and (dens > 0).all() and (rgb.abs() < 1).all()
    p = flat.clone().requires_grad_(True)
    f = lambda q: sum(t.sum() for t in ON.ngp_model(q, x, d, ts, gs, BMIN, BMAX, hidden_dim=8, density_dim=4)[:2])
    assert torch.autograd.gradcheck(f, (p,), eps=1e-6, atol=1e-5)
