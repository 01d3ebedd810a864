"""
Oracle restatement of learn_nerf/instant_ngp.py (hash-grid encoding + InstantNGPModel).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  uint32 wrap-around arithmetic of
hash_table_lookup (instant_ngp.py:219-223) is restated with int64 and explicit masking.
Parameter vector order (Flax creation order, instant_ngp.py:37-54): tables of level 0..L-1, then
Dense_0 (L*F -> hidden), ..., Dense_{density_layers} (hidden -> density_dim), colour Dense layers, Dense (-> 3).
"""
from typing import List, Sequence, Tuple

import torch

from .model import sinusoidal_emb

_MASK = 0xFFFFFFFF


def level_rows(grid_size: int, table_size: int) -> Tuple[int, bool]:
    """(rows of the level's table, hashed?) — dense below the table size, hashed above (instant_ngp.py:178)."""
    hashed = grid_size ** 3 > table_size
    return (table_size if hashed else grid_size ** 3), hashed


def hash_table_lookup_index(coords: torch.Tensor, table_rows: int) -> torch.Tensor:
    """instant_ngp.py:211-224: (x ^ 19349663*y ^ 83492791*z) mod T in uint32 arithmetic."""
    c = coords.to(torch.int64) & _MASK
    idx = (c[:, 0] ^ ((19_349_663 * c[:, 1]) & _MASK) ^ ((83_492_791 * c[:, 2]) & _MASK)) & _MASK
    return idx % table_rows


def hash_table_encoding(x: torch.Tensor, table: torch.Tensor, grid_size: int, table_size: int,
                        bbox_min: torch.Tensor, bbox_max: torch.Tensor, smooth: bool = False) -> torch.Tensor:
    """HashTableEncoding.__call__ (instant_ngp.py:134-208) -> [N, F]."""
    frac = torch.clamp((x - bbox_min) / (bbox_max - bbox_min), 0, 1)  # :138-140
    if smooth:
        fi = 0.5 + (grid_size - 2) * frac  # :144
    else:
        fi = (grid_size - 1) * frac  # :146
    floored = torch.clamp(torch.floor(fi), max=grid_size - 2)  # :147-150
    c = fi - floored  # :152
    if smooth:
        c = (c ** 2) * (3 - 2 * c)  # :154
    base = floored.to(torch.int64)
    rows, hashed = level_rows(grid_size, table_size)
    assert table.shape[0] == rows
    out = torch.zeros(x.shape[0], table.shape[1], dtype=table.dtype)
    for xo in (0, 1):  # :160-175
        for yo in (0, 1):
            for zo in (0, 1):
                off = torch.tensor([xo, yo, zo], dtype=torch.int64)
                offf = off.to(c.dtype)
                w = torch.prod(1 + (2 * c - 1) * offf - c, dim=-1, keepdim=True)
                coords = base + off
                if hashed:
                    idx = hash_table_lookup_index(coords, rows)  # :186-188
                else:
                    idx = coords[:, 0] + grid_size * (coords[:, 1] + grid_size * coords[:, 2])  # :199-201
                out = out + w * table[idx]
    return out


def ngp_spec(table_sizes: Sequence[int], grid_sizes: Sequence[int], feature_dim=2, d_freqs=4, hidden_dim=64,
             density_dim=16, density_layers=1, color_layers=2):
    """-> (table_rows list, dense dims list)"""
    rows = [level_rows(g, t)[0] for t, g in zip(table_sizes, grid_sizes)]
    dims = []
    fan = len(rows) * feature_dim
    for _ in range(density_layers):
        dims.append((fan, hidden_dim))
        fan = hidden_dim
    dims.append((fan, density_dim))
    fan = 6 * d_freqs + density_dim
    for _ in range(color_layers):
        dims.append((fan, hidden_dim))
        fan = hidden_dim
    dims.append((fan, 3))
    return rows, dims


def ngp_param_count(table_sizes, grid_sizes, feature_dim=2, **kw) -> int:
    rows, dims = ngp_spec(table_sizes, grid_sizes, feature_dim, **kw)
    return sum(r * feature_dim for r in rows) + sum(i * o + o for i, o in dims)


def ngp_model(flat: torch.Tensor, x: torch.Tensor, d: torch.Tensor, table_sizes, grid_sizes, bbox_min, bbox_max,
              feature_dim=2, smooth=False, d_freqs=4, hidden_dim=64, density_dim=16, density_layers=1,
              color_layers=2, operand_round=None):
    """InstantNGPModel.__call__ (instant_ngp.py:34-54) -> (density[N,1], rgb[N,3], {}).
    ``operand_round`` (e.g. oracle.model.bf16_round) is applied to both operands of every Dense matmul; it
    models the bf16-operand / fp32-accumulate arithmetic of the fused MFMA kernel, nothing else changes."""
    rnd = operand_round if operand_round is not None else (lambda t: t)
    rows, dims = ngp_spec(table_sizes, grid_sizes, feature_dim, d_freqs, hidden_dim, density_dim, density_layers,
                          color_layers)
    off = 0
    feats = []
    bmin = torch.as_tensor(bbox_min, dtype=x.dtype)
    bmax = torch.as_tensor(bbox_max, dtype=x.dtype)
    for r, t, g in zip(rows, table_sizes, grid_sizes):  # instant_ngp.py:104-118
        table = flat[off:off + r * feature_dim].reshape(r, feature_dim)
        off += r * feature_dim
        feats.append(hash_table_encoding(x, table, g, t, bmin, bmax, smooth))
    out = torch.cat(feats, dim=1)
    layers = []
    for i, o in dims:
        k = flat[off:off + i * o].reshape(i, o)
        off += i * o
        b = flat[off:off + o]
        off += o
        layers.append((k, b))
    assert off == flat.numel()
    d_emb = sinusoidal_emb(d, d_freqs)  # :37
    li = 0
    for _ in range(density_layers):  # :46-47
        out = torch.relu(rnd(out) @ rnd(layers[li][0]) + layers[li][1])
        li += 1
    out = rnd(out) @ rnd(layers[li][0]) + layers[li][1]  # :48
    li += 1
    density = torch.exp(out[:, :1])  # :49 (unclamped)
    out = torch.cat([d_emb, out], dim=1)  # :50 (d_emb first)
    for _ in range(color_layers):  # :51-52
        out = torch.relu(rnd(out) @ rnd(layers[li][0]) + layers[li][1])
        li += 1
    color = torch.tanh(rnd(out) @ rnd(layers[li][0]) + layers[li][1])  # :53
    return density, color, {}


def ngp_ref_spec(table_sizes, grid_sizes, feature_dim=2, hidden_dim=64, density_dim=16, density_layers=1,
                 color_layers=2, sh_degree=4):
    rows = [level_rows(g, t)[0] for t, g in zip(table_sizes, grid_sizes)]
    dims, fan = [], len(rows) * feature_dim
    for _ in range(density_layers):
        dims.append((fan, hidden_dim))
        fan = hidden_dim
    dims.append((fan, density_dim))
    fan = density_dim + sh_degree * sh_degree + 1
    for _ in range(color_layers):
        dims.append((fan, hidden_dim))
        fan = hidden_dim
    dims.append((fan, 3))
    return rows, dims


def ngp_ref_nerf_model(flat, x, d, table_sizes, grid_sizes, bbox_min, bbox_max, sh_degree=4, feature_dim=2,
                       hidden_dim=64, density_dim=16, density_layers=1, color_layers=2, operand_round=None):
    """InstantNGPRefNERFModel (instant_ngp.py:57-89): Ref-NeRF head on a smooth hash grid.  ``operand_round`` is
    applied to both operands of every Dense matmul (models LNRF_DENSE_BF16), nothing else changes."""
    from .ref_nerf import ref_nerf_base

    rnd = operand_round if operand_round is not None else (lambda t: t)

    rows, dims = ngp_ref_spec(table_sizes, grid_sizes, feature_dim, hidden_dim, density_dim, density_layers,
                              color_layers, sh_degree)
    off = 0
    tables = []
    for r in rows:
        tables.append(flat[off:off + r * feature_dim].reshape(r, feature_dim))
        off += r * feature_dim
    layers = []
    for i, o in dims:
        k = flat[off:off + i * o].reshape(i, o)
        off += i * o
        layers.append((k, flat[off:off + o]))
        off += o
    assert off == flat.numel()
    bmin = torch.as_tensor(bbox_min, dtype=x.dtype)
    bmax = torch.as_tensor(bbox_max, dtype=x.dtype)

    def spatial_block(xx):  # instant_ngp.py:72-83 (smooth=True)
        h = torch.cat([hash_table_encoding(xx, tb, g, t, bmin, bmax, True)
                       for tb, t, g in zip(tables, table_sizes, grid_sizes)], dim=1)
        li = 0
        for _ in range(density_layers):
            h = torch.relu(rnd(h) @ rnd(layers[li][0]) + layers[li][1])
            li += 1
        return rnd(h) @ rnd(layers[li][0]) + layers[li][1]

    def directional_block(inp):  # instant_ngp.py:85-89
        li = density_layers + 1
        h = inp
        for _ in range(color_layers):
            h = torch.relu(rnd(h) @ rnd(layers[li][0]) + layers[li][1])
            li += 1
        return rnd(h) @ rnd(layers[li][0]) + layers[li][1]

    return ref_nerf_base(spatial_block, directional_block, x, d, sh_degree)
