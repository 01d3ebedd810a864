"""
Parameter trees.  The reference keeps Flax parameter pytrees, e.g.
``{"Dense_0": {"kernel": [in,out], "bias": [out]}, ...}`` (model.py:49-60, train.py:53-58).
Here every leaf is a view into ONE flat fp32 device buffer (Flax creation order), so that the
optimiser, the gradient all-reduce and the weight packer each touch a single contiguous buffer.
"""
import math
from typing import Dict, List, Sequence, Tuple

import torch


class ParamTree(dict):
    """dict subclass that remembers the flat buffer its leaves alias (``.flat``)."""

    flat: torch.Tensor = None
    spec: List[Tuple[str, str, Tuple[int, ...]]] = None


def build_tree(flat: torch.Tensor, spec: Sequence[Tuple[str, str, Tuple[int, ...]]]) -> ParamTree:
    """spec: ordered (module_name, leaf_name, shape); module_name may contain '/' for nesting."""
    tree = ParamTree()
    off = 0
    for module, leaf, shape in spec:
        n = int(math.prod(shape))
        node = tree
        for part in module.split("/"):
            node = node.setdefault(part, {})
        node[leaf] = flat[off:off + n].view(*shape)
        off += n
    assert off == flat.numel(), (off, flat.numel())
    tree.flat = flat
    tree.spec = list(spec)
    return tree


def spec_size(spec) -> int:
    return sum(int(math.prod(s)) for _, _, s in spec)


def leaves_in_order(tree: Dict, spec) -> List[torch.Tensor]:
    out = []
    for module, leaf, _ in spec:
        node = tree
        for part in module.split("/"):
            node = node[part]
        out.append(node[leaf])
    return out


def flat_of(tree: Dict, spec) -> torch.Tensor:
    """The flat buffer behind a tree (no copy when the tree is a ParamTree view, else a concat)."""
    flat = getattr(tree, "flat", None)
    if flat is not None and flat.numel() == spec_size(spec):
        return flat
    leaves = leaves_in_order(tree, spec)
    return torch.cat([torch.as_tensor(l).reshape(-1).to(torch.float32) for l in leaves])


def lecun_normal_(w: torch.Tensor, fan_in: int, generator: torch.Generator) -> None:
    """Flax nn.Dense default kernel_init (lecun_normal): truncated normal [-2,2] * sqrt(1/fan_in)/0.8796."""
    tmp = torch.empty(w.shape, dtype=torch.float32)
    torch.nn.init.trunc_normal_(tmp, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=generator)
    w.copy_(tmp * (math.sqrt(1.0 / fan_in) / 0.87962566103423978))


def as_generator(rng) -> torch.Generator:
    """The reference passes a jax PRNG key (train.py:47); here an int seed or a torch.Generator."""
    if isinstance(rng, torch.Generator):
        return rng
    g = torch.Generator()
    g.manual_seed(int(rng) & 0x7FFFFFFFFFFFFFFF)
    return g


def split_seed(seed: int, n: int = 2) -> List[int]:
    """Deterministic stand-in for jax.random.split on integer seeds (SplitMix64 steps)."""
    out = []
    x = int(seed) & 0xFFFFFFFFFFFFFFFF
    for _ in range(n):
        x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        out.append(z ^ (z >> 31))
    return out


def default_device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("learn_nerf (MI355X build) needs a ROCm GPU: torch.cuda.is_available() is False")
    return torch.device("cuda", torch.cuda.current_device())
