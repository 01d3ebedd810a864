"""
Stand-ins for jax.random keys.  The reference threads a ``jax.random.PRNGKey`` through
NeRFRenderer.render_rays / RaySamples.*_sampling / TrainLoop (render.py:41, 127, 214; train.py:116).
Here a key is one of
  * ``Key(seed, ray_offset)`` (or a bare int): the kernels draw from the Philox4x32-10 stream
    (seed, stream_id, element index) documented in oracle/philox.py;
  * ``Uniforms(u)``: explicit uniforms in [0,1) of shape [N, count] (parity tests use these).
"""
from dataclasses import dataclass
from typing import List, Optional, Union

import torch

from .params import split_seed


@dataclass(frozen=True)
class Key:
    seed: int
    ray_offset: int = 0  # global index of the first ray of this shard (data parallel)

    def split(self, n: int = 2) -> List["Key"]:
        return [Key(s, self.ray_offset) for s in split_seed(self.seed, n)]


@dataclass
class Uniforms:
    u: torch.Tensor


KeyLike = Union[int, Key, Uniforms, torch.Generator]


def as_key(key: KeyLike) -> Union[Key, Uniforms]:
    if isinstance(key, (Key, Uniforms)):
        return key
    if isinstance(key, torch.Generator):
        return Key(int(key.initial_seed()))
    if isinstance(key, torch.Tensor) and key.numel() == 1:
        return Key(int(key.item()))
    return Key(int(key))


def split(key: KeyLike, n: int = 2):
    """jax.random.split stand-in. A tuple/list of n keys is passed through (explicit per-use keys)."""
    if isinstance(key, (tuple, list)):
        assert len(key) == n
        return [k if isinstance(k, (tuple, list)) else as_key(k) for k in key]
    k = as_key(key)
    if isinstance(k, Uniforms):
        raise TypeError("explicit Uniforms cannot be split; pass a tuple of keys")
    return k.split(n)


def sampler_args(key: KeyLike, stream_id: int):
    """-> dict(u=..., seed=..., stream_id=..., ray_offset=...) for the ops wrappers."""
    k = as_key(key)
    if isinstance(k, Uniforms):
        return dict(u=k.u.contiguous(), seed=0, stream_id=stream_id, ray_offset=0)
    return dict(u=None, seed=k.seed, stream_id=stream_id, ray_offset=k.ray_offset)
