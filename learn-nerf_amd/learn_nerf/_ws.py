"""
Device workspaces that outlive a step.

The models need GB-sized buffers every step (the forward's activation save, the backward's gradient dump / slabs, the
hash grid's tuple buckets).  Asking torch for them per call goes through the caching allocator every time, and a miss
there (after torch.cuda.empty_cache(), or when another size has split the cached segment) is a hipMalloc of hundreds
of MB in the middle of a step: tens of milliseconds of an idle GPU.  `lease()` hands out blocks from a small pool that
is filled once and then only re-used:

    with-less use:   l = lease("nerf_save", nbytes, device); buf = l.buf; ...; l.release()  (or drop the last reference)

A block is busy from lease() to release() / garbage collection of the Lease, so two contexts that are alive at the
same time (coarse and fine forward of one step, or one model instance serving as both) never share one.  Pools are
keyed by (purpose, device, current stream): work queued on one stream is ordered, so a released block can be handed to
the next call on that stream while its last kernel is still running; another stream gets its own blocks.
"""
from typing import Dict, List, Tuple

import torch

_POOLS: Dict[Tuple[str, int, int], List[list]] = {}
_ROUND = 2 << 20


class Lease:
    """A leased block: `.buf` is a uint8 tensor of exactly the requested size."""

    __slots__ = ("buf", "_entry")

    def __init__(self, buf, entry):
        self.buf = buf
        self._entry = entry

    def release(self) -> None:
        if self._entry is not None:
            self._entry[1] = False
            self._entry = None
            self.buf = None

    def __del__(self):
        self.release()


def lease(purpose: str, nbytes: int, device) -> Lease:
    device = torch.device(device)
    if device.type != "cuda":
        raise ValueError("workspaces live in GPU memory (the HIP path has no CPU fallback)")
    idx = device.index if device.index is not None else torch.cuda.current_device()
    stream = torch.cuda.current_stream(idx).cuda_stream
    pool = _POOLS.setdefault((purpose, idx, stream), [])
    need = max(int(nbytes), 1)
    best = None
    for entry in pool:  # entry = [tensor, busy]
        if not entry[1] and entry[0].numel() >= need and (best is None or entry[0].numel() < best[0].numel()):
            best = entry
    if best is None:
        cap = (need + _ROUND - 1) // _ROUND * _ROUND
        for entry in pool:  # grow an idle block instead of keeping a too-small one around
            if not entry[1]:
                entry[0] = None
                best = entry
                break
        if best is None:
            best = [None, False]
            pool.append(best)
        best[0] = torch.empty(cap, dtype=torch.uint8, device=torch.device("cuda", idx))
    best[1] = True
    return Lease(best[0][:need], best)


def pooled_bytes() -> int:
    return sum(e[0].numel() for pool in _POOLS.values() for e in pool if e[0] is not None)


def clear() -> None:
    """Drop every idle block (tests, or before a workload of another size)."""
    for pool in _POOLS.values():
        pool[:] = [e for e in pool if e[1]]
