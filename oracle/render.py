"""
Oracle restatement of the reference renderer (learn_nerf/render.py).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  torch-CPU, dtype follows the
inputs (float64 for golden vectors, float32 for the timed CPU baseline).
Sampling noise enters as explicit uniforms ``u`` instead of a jax PRNG key.
"""

from dataclasses import dataclass
from typing import Callable, Dict, Tuple

import torch


def ray_t_range(bbox: torch.Tensor, rays: torch.Tensor, min_t_range: float = 1e-3,
                epsilon: float = 1e-8) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """
    Slab test of rays[N,2,3] against bbox[2,3]; returns (t_min[N], t_max[N], mask[N]).
    Follows ray_t_range (render.py:346-389) vmapped by NeRFRenderer.t_range (render.py:93-111).
    """
    origin = rays[:, 0]  # render.py:363
    direction = rays[:, 1]  # render.py:364
    offsets = bbox[None] - origin[:, None]  # render.py:370  [N,2,3]
    ts = offsets / (direction[:, None] + epsilon)  # render.py:371 (epsilon added, not sign-matched)
    lo = ts.min(dim=1).values  # render.py:374-380
    hi = ts.max(dim=1).values
    min_t = torch.clamp(lo.max(dim=-1).values, min=0)  # render.py:383
    max_t = hi.min(dim=-1).values  # render.py:384
    max_t_clipped = torch.maximum(max_t, min_t + min_t_range)  # render.py:385
    mask = min_t < max_t  # render.py:388
    zero = torch.zeros_like(min_t)
    t_min = torch.where(mask, min_t, zero)  # render.py:389 null_range = [0, min_t_range]
    t_max = torch.where(mask, max_t_clipped, zero + min_t_range)
    return t_min, t_max, mask


def stratified_ts(t_min: torch.Tensor, t_max: torch.Tensor, count: int, u: torch.Tensor) -> torch.Tensor:
    """RaySamples.stratified_sampling (render.py:121-143): ts = t_min + i*bin + u*bin."""
    n = t_min.shape[0]
    if count == 0:
        return torch.zeros((n, 0), dtype=t_min.dtype)
    bin_size = ((t_max - t_min) / count)[:, None]  # render.py:138
    idx = torch.arange(count, dtype=t_min.dtype)[None]
    bin_starts = idx * bin_size + t_min[:, None]  # render.py:139-141
    return bin_starts + u.to(t_min.dtype) * bin_size  # render.py:142-143


@dataclass
class RaySamples:
    """Mirror of RaySamples (render.py:114-290)."""

    t_min: torch.Tensor
    t_max: torch.Tensor
    mask: torch.Tensor
    ts: torch.Tensor

    def points(self, rays: torch.Tensor) -> torch.Tensor:
        return rays[:, :1] + rays[:, 1:2] * self.ts[:, :, None]  # render.py:153

    def starts(self) -> torch.Tensor:  # render.py:259-261
        t_mid = (self.ts[:, 1:] + self.ts[:, :-1]) / 2
        return torch.cat([self.t_min[:, None], t_mid], dim=1)

    def ends(self) -> torch.Tensor:  # render.py:263-265
        t_mid = (self.ts[:, 1:] + self.ts[:, :-1]) / 2
        return torch.cat([t_mid, self.t_max[:, None]], dim=1)

    def deltas(self) -> torch.Tensor:  # render.py:267-268
        return self.ends() - self.starts()

    def termination_probs(self, densities: torch.Tensor) -> torch.Tensor:
        """render.py:270-287 -> [N, T+1]; slot T is 'reached the background'."""
        density_dt = densities * self.deltas()
        acc_cur = torch.cumsum(density_dt, dim=1)  # render.py:275
        acc_prev = torch.cat([torch.zeros_like(acc_cur[:, :1]), acc_cur], dim=1)
        prob_survive = torch.exp(-acc_prev)  # render.py:279
        prob_terminate = torch.cat(
            [1 - torch.exp(-density_dt), torch.ones_like(acc_cur[:, :1])], dim=1
        )  # render.py:283-285
        return prob_survive * prob_terminate

    def render_rays(self, densities, channels, background) -> torch.Tensor:
        """render.py:155-176."""
        probs = self.termination_probs(densities)
        bg = background.to(channels.dtype)[None, None].expand(channels.shape[0], 1, -1)
        colors = torch.cat([channels, bg], dim=1)
        summed = (probs[..., None] * colors).sum(dim=1)
        return torch.where(self.mask[:, None], summed, background.to(channels.dtype)[None])

    def render_alpha(self, densities) -> torch.Tensor:
        """render.py:178-190."""
        probs = self.termination_probs(densities)
        return torch.where(self.mask[:, None], 1 - probs[:, -1:], torch.zeros_like(probs[:, -1:]))

    def average_aux_losses(self, densities, aux: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """render.py:192-209."""
        probs = self.termination_probs(densities)[:, :-1]
        out = {}
        for k, v in aux.items():
            per_ray = (v * probs).sum(dim=-1)
            out[k] = torch.where(self.mask, per_ray, torch.zeros_like(per_ray)).mean()
        return out

    def fine_sampling(self, count: int, u: torch.Tensor, densities: torch.Tensor,
                      combine: bool = True, eps: float = 1e-8) -> "RaySamples":
        """
        render.py:211-257. ``u`` [N,count] replaces the jax key. jnp.interp is restated
        as SURVEY.md Appendix A.6 (searchsorted right, clip, flat-segment rule).
        """
        w = self.termination_probs(densities)[:, :-1] + eps  # render.py:232
        xs = torch.cumsum(w, dim=1)  # render.py:235
        xs = torch.cat([torch.zeros_like(xs[:, :1]), xs], dim=1)
        xs = xs / xs[:, -1:]  # render.py:237
        ys = torch.cat([self.t_min[:, None], self.ends()], dim=1)  # render.py:238-241
        zero = torch.zeros_like(self.t_min)
        inp = stratified_ts(zero, zero + 1, count, u)  # render.py:244-250
        new_ts = interp_rows(inp, xs, ys)  # render.py:251
        if combine:
            new_ts = torch.sort(torch.cat([self.ts, new_ts], dim=1), dim=1).values  # render.py:253-255
        return RaySamples(self.t_min, self.t_max, self.mask, new_ts)


def interp_rows(x: torch.Tensor, xs: torch.Tensor, ys: torch.Tensor) -> torch.Tensor:
    """Row-wise jnp.interp / numpy.interp (public semantics)."""
    if x.shape[1] == 0:
        return x.clone()
    L = xs.shape[1]
    i = torch.searchsorted(xs.contiguous(), x.contiguous(), right=True).clamp(1, L - 1)
    x0 = torch.gather(xs, 1, i - 1)
    x1 = torch.gather(xs, 1, i)
    y0 = torch.gather(ys, 1, i - 1)
    y1 = torch.gather(ys, 1, i)
    dx = x1 - x0
    flat = dx.abs() <= torch.finfo(x.dtype).tiny
    safe_dx = torch.where(flat, torch.ones_like(dx), dx)
    out = torch.where(flat, y0, y0 + (x - x0) / safe_dx * (y1 - y0))
    out = torch.where(x < xs[:, :1], ys[:, :1].expand_as(out), out)
    out = torch.where(x > xs[:, -1:], ys[:, -1:].expand_as(out), out)
    return out


ModelFn = Callable[[torch.Tensor, torch.Tensor], Tuple[torch.Tensor, torch.Tensor, Dict[str, torch.Tensor]]]


def render_rays(model_fn: ModelFn, background: torch.Tensor, batch: torch.Tensor, ts: RaySamples):
    """Free function render_rays (render.py:293-343). model_fn(x[M,3], d[M,3])."""
    all_points = ts.points(batch)  # render.py:318
    n, t = all_points.shape[:2]
    direction_batch = batch[:, 1:2].expand(n, t, 3)  # render.py:319
    densities, rgbs, aux = model_fn(all_points.reshape(-1, 3), direction_batch.reshape(-1, 3))
    densities = densities.reshape(n, t)
    rgbs = rgbs.reshape(n, t, 3)
    aux = {k: v.reshape(n, t) for k, v in aux.items()}
    outputs = ts.render_rays(densities, rgbs, background)  # render.py:329
    alphas = ts.render_alpha(densities)  # render.py:330
    coords = ts.render_rays(densities, all_points, torch.zeros(3, dtype=rgbs.dtype))  # render.py:331
    aux_mean = ts.average_aux_losses(densities, aux)  # render.py:332
    return dict(outputs=outputs, rgbs=rgbs, densities=densities, alphas=alphas, coords=coords), aux_mean


def render_hierarchy(coarse_fn: ModelFn, fine_fn: ModelFn, background: torch.Tensor,
                     bbox_min: torch.Tensor, bbox_max: torch.Tensor, batch: torch.Tensor,
                     coarse_ts: int, fine_ts: int, u_coarse: torch.Tensor, u_fine: torch.Tensor,
                     min_t_range: float = 1e-3):
    """NeRFRenderer.render_rays (render.py:39-91) with explicit uniforms."""
    bbox = torch.stack([bbox_min, bbox_max]).to(batch.dtype)
    t_min, t_max, mask = ray_t_range(bbox, batch, min_t_range=min_t_range)  # render.py:53
    c_ts = RaySamples(t_min, t_max, mask, stratified_ts(t_min, t_max, coarse_ts, u_coarse))  # :57-63
    coarse_out, coarse_aux = render_rays(coarse_fn, background, batch, c_ts)  # :64-70
    f_ts = c_ts.fine_sampling(fine_ts, u_fine, coarse_out["densities"].detach())  # :73-77 stop_gradient
    fine_out, fine_aux = render_rays(fine_fn, background, batch, f_ts)  # :78-84
    return dict(coarse=coarse_out, fine=fine_out, coarse_aux=coarse_aux, fine_aux=fine_aux,
                coarse_ts=c_ts, fine_ts=f_ts)
