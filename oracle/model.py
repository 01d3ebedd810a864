"""
Oracle restatement of learn_nerf/model.py (NeRFModel, sinusoidal_emb).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Parameter container: one flat vector per model in Flax creation order
(model.py:49-60): Dense_0.kernel[in,out] row-major, Dense_0.bias,
Dense_1.kernel, ... Dense_11.bias  (593,924 floats for the default model).
"""

import math
from typing import Dict, List, Optional, Tuple

import torch


def sinusoidal_emb(coords: torch.Tensor, freqs: int) -> torch.Tensor:
    """model.py:65-77: per coordinate [sin(2^0 c..2^(F-1) c), cos(2^0 c..2^(F-1) c)]."""
    coeffs = 2.0 ** torch.arange(freqs, dtype=coords.dtype)  # model.py:72
    inputs = coords[..., None] * coeffs  # model.py:73
    combined = torch.cat([torch.sin(inputs), torch.cos(inputs)], dim=-1)  # model.py:74-76
    return combined.reshape(combined.shape[:-2] + (-1,))  # model.py:77


def nerf_layer_dims(input_layers=5, mid_layers=4, hidden_dim=256, color_layer_dim=128,
                    x_freqs=10, d_freqs=4) -> List[Tuple[int, int]]:
    """(fan_in, fan_out) of Dense_0..Dense_{n-1} in creation order (model.py:49-60)."""
    xe, de = 6 * x_freqs, 6 * d_freqs
    dims = []
    fan_in = xe
    for _ in range(input_layers):  # model.py:50-51
        dims.append((fan_in, hidden_dim))
        fan_in = hidden_dim
    fan_in = hidden_dim + xe  # model.py:52
    for _ in range(mid_layers):  # model.py:53-56
        dims.append((fan_in, hidden_dim))
        fan_in = hidden_dim
    dims.append((hidden_dim, 1))  # model.py:57
    dims.append((hidden_dim + de, color_layer_dim))  # model.py:58-59
    dims.append((color_layer_dim, 3))  # model.py:60
    return dims


def param_count(dims) -> int:
    return sum(i * o + o for i, o in dims)


def unflatten(flat: torch.Tensor, dims) -> List[Tuple[torch.Tensor, torch.Tensor]]:
    out, off = [], 0
    for i, o in dims:
        k = flat[off:off + i * o].reshape(i, o)
        off += i * o
        b = flat[off:off + o]
        off += o
        out.append((k, b))
    assert off == flat.numel()
    return out


def lecun_normal_init(dims, generator: torch.Generator, dtype=torch.float32) -> torch.Tensor:
    """
    Flax nn.Dense defaults (train.py:49-50 -> model.init): kernel = lecun_normal =
    truncated normal on [-2, 2] scaled by sqrt(1/fan_in)/0.87962566103423978, bias = 0.
    """
    parts = []
    for i, o in dims:
        w = torch.empty(i, o, dtype=torch.float64)
        torch.nn.init.trunc_normal_(w, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=generator)
        w = w * (math.sqrt(1.0 / i) / 0.87962566103423978)
        parts += [w.reshape(-1), torch.zeros(o, dtype=torch.float64)]
    return torch.cat(parts).to(dtype)


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    """Round-to-nearest-even to bfloat16 and back (what v_cvt_pk_bf16_f32 does)."""
    return x.to(torch.float32).to(torch.bfloat16).to(x.dtype)


def nerf_mlp(flat: torch.Tensor, x: torch.Tensor, d: torch.Tensor, input_layers=5, mid_layers=4,
             hidden_dim=256, color_layer_dim=128, x_freqs=10, d_freqs=4,
             operand_round=None, return_hidden: bool = False):
    """
    NeRFModel.__call__ (model.py:43-62).  ``operand_round`` (e.g. bf16_round) is applied to
    every matmul operand (layer inputs and kernels) to model the bf16-MFMA kernel, which keeps
    biases, accumulation and activations in fp32.
    Returns (density[M,1], rgb[M,3], {}).
    """
    rnd = operand_round if operand_round is not None else (lambda t: t)
    dims = nerf_layer_dims(input_layers, mid_layers, hidden_dim, color_layer_dim, x_freqs, d_freqs)
    layers = unflatten(flat, dims)
    x_emb = sinusoidal_emb(x, x_freqs)  # model.py:46
    d_emb = sinusoidal_emb(d, d_freqs)  # model.py:47

    def dense(idx, inp):
        k, b = layers[idx]
        return rnd(inp) @ rnd(k) + b

    hidden = []
    li = 0
    z = x_emb
    for _ in range(input_layers):  # model.py:50-51
        z = torch.relu(dense(li, z))
        hidden.append(z)
        li += 1
    z = torch.cat([z, x_emb], dim=-1)  # model.py:52
    for i in range(mid_layers):  # model.py:53-56
        if i > 0:
            z = torch.relu(z)
        z = dense(li, z)
        hidden.append(z)
        li += 1
    density = torch.nn.functional.softplus(dense(li, z))  # model.py:57
    li += 1
    z = torch.cat([z, d_emb], dim=-1)  # model.py:58
    z = torch.relu(dense(li, z))  # model.py:59
    hidden.append(z)
    li += 1
    rgb = torch.tanh(dense(li, z))  # model.py:60
    if return_hidden:
        return density, rgb, {}, hidden
    return density, rgb, {}


def make_nerf_fn(flat: torch.Tensor, operand_round=None, **kw):
    return lambda x, d: nerf_mlp(flat, x, d, operand_round=operand_round, **kw)
