"""
learn_nerf.render — NeRFRenderer, RaySamples, render_rays, ray_t_range
(reference: learn_nerf/render.py).  Same names, arguments and return structure; arrays are
torch tensors on the GPU and every arithmetic step is a HIP kernel behind include/lnrf.h.
"""
from dataclasses import dataclass
from typing import Any, Dict, Sequence, Tuple

import torch

from . import ops
from .model import ModelBase
from .rng import KeyLike, sampler_args, split

STREAM_COARSE = 0  # Philox stream ids (oracle/philox.py)
STREAM_FINE = 1


def _vec3(v) -> Tuple[float, float, float]:
    if isinstance(v, torch.Tensor):
        v = v.detach().cpu().tolist()
    return tuple(float(x) for x in v)


def _mask_u8(mask: torch.Tensor) -> torch.Tensor:
    return mask.view(torch.uint8) if mask.dtype == torch.bool else mask


@dataclass
class NeRFRenderer:
    """
    A NeRF hierarchy with corresponding settings for rendering rays (render.py:11-111).

    :param coarse / fine: the models.  :param coarse_params / fine_params: their parameter trees.
    :param background: the [3] RGB background tensor.  :param bbox_min / bbox_max: scene bounds.
    :param coarse_ts: samples per ray for the coarse model.  :param fine_ts: additional fine samples.
    """

    coarse: ModelBase
    fine: ModelBase
    coarse_params: Any
    fine_params: Any
    background: torch.Tensor
    bbox_min: Any
    bbox_max: Any
    coarse_ts: int
    fine_ts: int

    min_t_range: float = 1e-3

    def render_rays(self, key: KeyLike, batch: torch.Tensor) -> Dict[str, Dict[str, torch.Tensor]]:
        """
        :param key: RNG key (rng.Key / int seed) or a (coarse, fine) pair of keys / rng.Uniforms.
        :param batch: an [N x 2 x 3] (or [N x 3 x 3]) batch of (origin, direction[, colour]) rays.
        :return: dict with keys "fine", "coarse", "fine_aux", "coarse_aux" (render.py:86-91).
        """
        coarse_key, fine_key = split(key, 2)  # render.py:55
        ca = sampler_args(coarse_key, STREAM_COARSE)
        t_min, t_max, mask, ts = ops.ray_aabb_stratified(  # render.py:53, 57-63
            batch, _vec3(self.bbox_min), _vec3(self.bbox_max), self.coarse_ts, min_t_range=self.min_t_range, **ca)
        coarse_ts = RaySamples(t_min=t_min, t_max=t_max, mask=mask, ts=ts)
        coarse_out, coarse_aux = render_rays(self.coarse, self.coarse_params, self.background, batch, coarse_ts)
        fine_ts = coarse_ts.fine_sampling(count=self.fine_ts, key=fine_key,
                                          densities=coarse_out["densities"])  # render.py:73-77
        fine_out, fine_aux = render_rays(self.fine, self.fine_params, self.background, batch, fine_ts)
        return dict(coarse=coarse_out, fine=fine_out, coarse_aux=coarse_aux, fine_aux=fine_aux)

    def t_range(self, batch: torch.Tensor, epsilon: float = 1e-8):
        """(t_min, t_max, mask) of [N] tensors for the scene bounding box (render.py:93-111)."""
        t_min, t_max, mask, _ = ops.ray_aabb_stratified(batch, _vec3(self.bbox_min), _vec3(self.bbox_max), 0,
                                                        min_t_range=self.min_t_range, epsilon=epsilon)
        return t_min, t_max, mask.bool()


@dataclass
class RaySamples:
    """Samples along a batch of rays (render.py:114-290)."""

    t_min: torch.Tensor
    t_max: torch.Tensor
    mask: torch.Tensor
    ts: torch.Tensor

    @classmethod
    def stratified_sampling(cls, t_min, t_max, mask, count: int, key: KeyLike) -> "RaySamples":
        """render.py:121-143."""
        ts = ops.stratified(t_min.contiguous(), t_max.contiguous(), count, **sampler_args(key, STREAM_COARSE))
        return cls(t_min=t_min, t_max=t_max, mask=mask, ts=ts)

    def points(self, rays: torch.Tensor) -> torch.Tensor:
        """[N x T x 3] points at all ts (render.py:145-153)."""
        pts, _ = ops.ray_points(rays.contiguous(), self.ts, want_dirs=False)
        return pts

    def render_rays(self, densities, rgbs, background) -> torch.Tensor:
        """Volumetric rendering of [N x T x 3] colours -> [N x 3] (render.py:155-176)."""
        out, _, _, _ = ops.composite_fwd(None, self.ts, self.t_min, self.t_max, _mask_u8(self.mask),
                                         densities.contiguous(), rgbs.contiguous(), background.contiguous(),
                                         want_coords=False)
        return out

    def render_alpha(self, densities) -> torch.Tensor:
        """[N x 1] hit probabilities (render.py:178-190)."""
        n, t = self.ts.shape
        dummy = torch.zeros((n, t, 3), dtype=torch.float32, device=self.ts.device)
        _, alphas, _, _ = ops.composite_fwd(None, self.ts, self.t_min, self.t_max, _mask_u8(self.mask),
                                            densities.contiguous(), dummy, dummy[0, 0], want_coords=False)
        return alphas[:, None]

    def average_aux_losses(self, densities, aux: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Density-weighted means of per-sample auxiliary losses (render.py:192-209)."""
        if not aux:
            return {}
        names = list(aux.keys())
        n, t = self.ts.shape
        stacked = torch.stack([aux[k].reshape(n, t) for k in names], dim=-1).contiguous()
        dummy = torch.zeros((n, t, 3), dtype=torch.float32, device=self.ts.device)
        _, _, _, aux_sum = ops.composite_fwd(None, self.ts, self.t_min, self.t_max, _mask_u8(self.mask),
                                             densities.contiguous(), dummy, dummy[0, 0], aux=stacked,
                                             want_coords=False)
        means = aux_sum.mean(dim=0)
        return {k: means[i] for i, k in enumerate(names)}

    def fine_sampling(self, count: int, key: KeyLike, densities: torch.Tensor, combine: bool = True,
                      eps: float = 1e-8) -> "RaySamples":
        """Hierarchical (inverse-CDF) sampling from coarse densities (render.py:211-257)."""
        new_ts = ops.fine_sample(self.ts, self.t_min, self.t_max, densities.detach().contiguous(), count,
                                 combine=combine, eps=eps, **sampler_args(key, STREAM_FINE))
        return RaySamples(t_min=self.t_min, t_max=self.t_max, mask=self.mask, ts=new_ts)

    def starts(self) -> torch.Tensor:  # render.py:259-261
        return ops.bin_edges(self.ts, self.t_min, self.t_max)[0]

    def ends(self) -> torch.Tensor:  # render.py:263-265
        return ops.bin_edges(self.ts, self.t_min, self.t_max)[1]

    def deltas(self) -> torch.Tensor:  # render.py:267-268
        s, e = ops.bin_edges(self.ts, self.t_min, self.t_max)
        return e - s

    def termination_probs(self, densities: torch.Tensor) -> torch.Tensor:
        """[N x (T+1)] termination probabilities; slot T = reached the background (render.py:270-287)."""
        return ops.termination_probs(self.ts, self.t_min, self.t_max, densities.contiguous())


def render_rays(model: ModelBase, params: Any, background: torch.Tensor, batch: torch.Tensor,
                ts: RaySamples) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
    """
    Render a batch of rays using a model (render.py:293-343).

    :return: (out, aux): out has outputs[N,3], rgbs[N,T,3], densities[N,T], alphas[N,1], coords[N,3];
             aux maps loss names to scalar means over rays.
    """
    flat = model.flat(params)
    densities, rgbs, aux, _ = model.forward_rays(flat, batch, ts.ts, save=False)  # render.py:318-327
    names = list(aux.keys())
    stacked = torch.stack([aux[k] for k in names], dim=-1).contiguous() if names else None
    outputs, alphas, coords, aux_sum = ops.composite_fwd(batch, ts.ts, ts.t_min, ts.t_max, _mask_u8(ts.mask),
                                                         densities, rgbs, background.contiguous(), aux=stacked)
    aux_mean = {}
    if names:
        means = aux_sum.mean(dim=0)  # render.py:205-208
        aux_mean = {k: means[i] for i, k in enumerate(names)}
    return (dict(outputs=outputs, rgbs=rgbs, densities=densities, alphas=alphas[:, None], coords=coords),
            aux_mean)


def ray_t_range(bbox: torch.Tensor, ray: torch.Tensor, min_t_range: float = 1e-3, epsilon: float = 1e-8):
    """
    For a single ray [2 x 3] (or a batch [N x 2 x 3]) compute (t_min, t_max) against bbox [2 x 3] and the
    intersection mask (render.py:346-389).  Returns (ts[..., 2], mask[...]).
    """
    single = ray.dim() == 2
    rays = ray[None] if single else ray
    bb = bbox.detach().cpu()
    t_min, t_max, mask, _ = ops.ray_aabb_stratified(rays.contiguous(), _vec3(bb[0]), _vec3(bb[1]), 0,
                                                    min_t_range=min_t_range, epsilon=epsilon)
    ts = torch.stack([t_min, t_max], dim=-1)
    mask = mask.bool()
    return (ts[0], mask[0]) if single else (ts, mask)


def z_depth(coords: torch.Tensor, alphas: torch.Tensor, camera_origin, camera_direction,
            max_depth: float) -> torch.Tensor:
    """
    Normalised z-depth of the expected collision point (scripts/render_new_dataset.py:96-116): for rays that
    hit with probability > 0.9, ((coords - origin) . direction) / (alpha + 1e-8) clipped to [0, max_depth];
    max_depth otherwise; divided by max_depth.  coords [N,3] is the alpha-weighted mean point, alphas [N,1].
    """
    origin = torch.as_tensor(camera_origin, dtype=coords.dtype, device=coords.device)
    direction = torch.as_tensor(camera_direction, dtype=coords.dtype, device=coords.device)
    along = ((coords - origin) * direction).sum(dim=-1, keepdim=True) / (alphas + 1e-8)
    depth = torch.where(alphas > 0.9, along, torch.full_like(along, max_depth))
    return depth.clamp(0.0, max_depth) / max_depth
