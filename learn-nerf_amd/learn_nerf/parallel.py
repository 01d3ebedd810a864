"""
Data-parallel plumbing for the train step (no reference counterpart: the reference is
single-device).  One process per GPU; rays are independent units, so a global
batch [N,3,3] is split into contiguous shards and the only exchange per step is ONE all-reduce (sum)
of the flat fp32 gradient buffer over RCCL (backend "nccl" on ROCm) — or gloo on CPU in the tests.
The mean over ranks is folded into the fused Adam kernel (grad_scale = 1/world).
"""
import os
from typing import Optional, Tuple

import torch


def dist_module():
    import torch.distributed as dist

    return dist if (dist.is_available() and dist.is_initialized()) else None


def world_info() -> Tuple[int, int]:
    """(rank, world_size); (0, 1) when torch.distributed is not initialised."""
    d = dist_module()
    return (d.get_rank(), d.get_world_size()) if d else (0, 1)


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, **kwargs)
    return rank, local_rank, world


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n rays for `rank`; n must divide evenly (equal-size shards keep
    the average of per-rank gradients equal to the gradient of the global mean loss, train.py:141-142)."""
    if n % world != 0:
        raise ValueError(f"global batch of {n} rays does not split evenly over {world} ranks")
    per = n // world
    return rank * per, (rank + 1) * per


def shard_rays(batch: torch.Tensor, rank: int, world: int) -> Tuple[torch.Tensor, int]:
    """-> (this rank's rays, global index of its first ray = Philox ray_offset)."""
    lo, hi = shard_bounds(batch.shape[0], rank, world)
    return batch[lo:hi].contiguous(), lo


def all_reduce_sum_(flat: torch.Tensor) -> torch.Tensor:
    """In-place sum over ranks of a flat gradient buffer (single bucket, one collective per step)."""
    d = dist_module()
    if d is not None and d.get_world_size() > 1:
        d.all_reduce(flat)
    return flat


def grad_scale() -> float:
    """1/world: applied inside the fused Adam kernel and to the logged grad_norm."""
    return 1.0 / world_info()[1]
