"""
Oracle restatement of learn_nerf/ref_nerf.py (Ref-NeRF head on the NeRF spatial MLP).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The spherical-harmonic basis is generated from the
associated-Legendre recurrence (real SH with Condon-Shortley phase, index l*l + l + m), which is the
convention of the closed-form table at ref_nerf.py:146-311 (tests pin it against closed forms and
orthonormality).  Analytic normals use torch.autograd.grad(create_graph=True), as the reference uses
jax.grad inside the model (ref_nerf.py:38-43), so second-order terms are exact.
"""
import math
from typing import Dict, List, Tuple

import torch

from .model import sinusoidal_emb, unflatten

HARMONIC_COUNTS = [1, 3, 5, 7, 9, 11, 13, 15]  # ref_nerf.py:15


def spherical_harmonic(sh_degree: int, coords: torch.Tensor) -> torch.Tensor:
    """[N,3] unit vectors -> [N, sh_degree^2] real spherical harmonics (ref_nerf.py:146-311)."""
    assert 1 <= sh_degree <= 8
    x, y, z = coords[:, 0], coords[:, 1], coords[:, 2]
    out = [None] * (sh_degree * sh_degree)
    # A_m + i B_m = (x + i y)^m
    A = [torch.ones_like(x)]
    B = [torch.zeros_like(x)]
    for m in range(1, sh_degree):
        A.append(A[m - 1] * x - B[m - 1] * y)
        B.append(A[m - 1] * y + B[m - 1] * x)
    for m in range(sh_degree):
        # Pi_l^m(z) = d^m P_l / dz^m : Pi_m^m = (2m-1)!!, Pi_{m+1}^m = (2m+1) z Pi_m^m, three-term recurrence
        pmm = float(math.prod(range(2 * m - 1, 0, -2))) if m > 0 else 1.0
        prev2 = None
        prev1 = torch.full_like(z, pmm)
        for l in range(m, sh_degree):
            if l == m:
                pi = prev1
            elif l == m + 1:
                pi = (2 * m + 1) * z * prev1
                prev2, prev1 = prev1, pi
            else:
                pi = ((2 * l - 1) * z * prev1 - (l + m - 1) * prev2) / (l - m)
                prev2, prev1 = prev1, pi
            k = math.sqrt((2 * l + 1) / (4 * math.pi) * math.factorial(l - m) / math.factorial(l + m))
            if m == 0:
                out[l * l + l] = k * pi
            else:
                c = math.sqrt(2.0) * k * (-1.0) ** m
                out[l * l + l + m] = c * pi * A[m]
                out[l * l + l - m] = c * pi * B[m]
    return torch.stack(out, dim=1)


def integrated_directional_encoding(sh_degree: int, coords: torch.Tensor, roughness: torch.Tensor) -> torch.Tensor:
    """ref_nerf.py:121-143: harmonics attenuated by exp(-roughness * l(l+1)/2), l repeated 2l+1 times."""
    levels = torch.tensor([l for l, c in enumerate(HARMONIC_COUNTS[:sh_degree]) for _ in range(c)],
                          dtype=roughness.dtype)
    attenuation = torch.exp(-roughness * (levels * (levels + 1)) / 2)
    return spherical_harmonic(sh_degree, coords) * attenuation


def linear_rgb_to_srgb(colors: torch.Tensor) -> torch.Tensor:
    """ref_nerf.py:110-118."""
    safe = torch.clamp(colors, min=1e-5)
    return torch.where(colors <= 0.0031308, 12.92 * colors, 1.055 * safe ** (1 / 2.4) - 0.055)


def _safe_normalize(v: torch.Tensor, eps: float = 1e-10) -> torch.Tensor:
    return v / torch.sqrt((v ** 2).sum(dim=-1, keepdim=True) + eps)  # ref_nerf.py:314-317


def _leaky_clip(x: torch.Tensor) -> torch.Tensor:
    return x + (torch.clamp(x, 0, 1) - x).detach()  # ref_nerf.py:320-326


def ref_nerf_layer_dims(input_layers=5, mid_layers=4, hidden_dim=256, color_layer_dim=128, x_freqs=10,
                        sh_degree=4) -> List[Tuple[int, int]]:
    """Dense_0..8 spatial block (ref_nerf.py:92-103), Dense_9 (hidden + sh^2 + 1 -> 128), Dense_10 (128 -> 3)."""
    xe = 6 * x_freqs
    dims, fan = [], xe
    for _ in range(input_layers):
        dims.append((fan, hidden_dim))
        fan = hidden_dim
    fan = hidden_dim + xe
    for _ in range(mid_layers):
        dims.append((fan, hidden_dim))
        fan = hidden_dim
    dims.append((hidden_dim + sh_degree * sh_degree + 1, color_layer_dim))
    dims.append((color_layer_dim, 3))
    return dims


def ref_nerf_base(spatial_block, directional_block, x: torch.Tensor, d: torch.Tensor, sh_degree: int):
    """RefNERFBase.__call__ (ref_nerf.py:35-77) for arbitrary spatial / directional blocks."""
    xr = x if x.requires_grad else x.clone().requires_grad_(True)
    out = spatial_block(xr)
    (real_normal,) = torch.autograd.grad(-out[:, 0].sum(), xr, create_graph=True)  # ref_nerf.py:38-42
    real_normal = _safe_normalize(real_normal)
    density = torch.exp(out[:, 0:1])  # :45-48
    diffuse = torch.sigmoid(out[:, 1:4] - math.log(3))  # :52
    spectral = torch.sigmoid(out[:, 4:5])
    roughness = torch.nn.functional.softplus(out[:, 5:6])
    normal = _safe_normalize(out[:, 6:9])
    reflection = d - 2 * normal * (d * normal).sum(dim=-1, keepdim=True)  # :59
    enc = integrated_directional_encoding(sh_degree, reflection, roughness)
    normal_dot = (-d * normal).sum(dim=-1, keepdim=True)
    dir_input = torch.cat([out, enc, normal_dot], dim=1)  # :63 (the WHOLE spatial_out, not just the bottleneck)
    dir_output = directional_block(dir_input)
    spectral_color = torch.sigmoid(dir_output)
    full_color = linear_rgb_to_srgb(_leaky_clip(spectral_color * spectral + diffuse)) * 2 - 1  # :67-71
    aux = dict(normal_mse=((normal - real_normal) ** 2).sum(dim=-1),  # :72-75
               neg_normal=torch.clamp((normal * d).sum(dim=-1), min=0.0) ** 2)
    return density, full_color, aux


def ref_nerf_model(flat: torch.Tensor, x: torch.Tensor, d: torch.Tensor, sh_degree=4, input_layers=5, mid_layers=4,
                   hidden_dim=256, color_layer_dim=128, x_freqs=10, operand_round=None):
    """RefNERFModel (ref_nerf.py:80-107) -> (density, rgb, aux dict).  ``operand_round`` (e.g.
    oracle.model.bf16_round) is applied to both operands of every Dense matmul (models LNRF_DENSE_BF16)."""
    rnd = operand_round if operand_round is not None else (lambda t: t)
    dims = ref_nerf_layer_dims(input_layers, mid_layers, hidden_dim, color_layer_dim, x_freqs, sh_degree)
    layers = unflatten(flat, dims)

    def spatial_block(xx):  # ref_nerf.py:92-103
        x_emb = sinusoidal_emb(xx, x_freqs)
        z = x_emb
        li = 0
        for _ in range(input_layers):
            z = torch.relu(rnd(z) @ rnd(layers[li][0]) + layers[li][1])
            li += 1
        z = torch.cat([z, x_emb], dim=-1)
        for i in range(mid_layers):
            if i > 0:
                z = torch.relu(z)
            z = rnd(z) @ rnd(layers[li][0]) + layers[li][1]
            li += 1
        return z

    def directional_block(inp):  # ref_nerf.py:105-107
        li = input_layers + mid_layers
        h = torch.relu(rnd(inp) @ rnd(layers[li][0]) + layers[li][1])
        return rnd(h) @ rnd(layers[li + 1][0]) + layers[li + 1][1]

    return ref_nerf_base(spatial_block, directional_block, x, d, sh_degree)
