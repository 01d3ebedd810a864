"""
Data-parallel plumbing for the train step (no reference counterpart: the reference is
single-device).  One process per GPU; rays are independent units, so a global
batch [N,3,3] is split into contiguous shards and the only exchange per step is ONE all-reduce (sum)
of the flat fp32 gradient buffer over RCCL (backend "nccl" on ROCm) — or gloo on CPU in the tests.
The mean over ranks is folded into the fused Adam kernel (grad_scale = 1/world).
"""
import ctypes
import os
import time
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


class AbiComm:
    """
    RCCL communicator behind the C ABI (include/lnrf.h: lnrf_comm_get_unique_id / init / allreduce / destroy) —
    the exchange step as a non-torch caller of liblnrf.so would run it.  Bootstrap = hand rank 0's 128-byte unique
    id to every rank: from_process_group() uses an existing torch.distributed group for that, from_file() a shared
    path (no torch.distributed at all).  The torch.distributed all-reduce (also RCCL) stays the default backend of
    reduce_gradient_; use_abi_comm(comm) switches the gradient exchange to this handle.
    """

    def __init__(self, unique_id: bytes, rank: int, world: int):
        from . import _lib as L

        if len(unique_id) != 128:
            raise ValueError("unique id must be LNRF_COMM_UNIQUE_ID_BYTES = 128 bytes")
        self._L = L
        handle = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        L.check(L.lib().lnrf_comm_init(buf, rank, world, ctypes.byref(handle)), "comm_init")
        self._handle = handle
        self.rank, self.world = rank, world

    @staticmethod
    def new_unique_id() -> bytes:
        from . import _lib as L

        buf = ctypes.create_string_buffer(128)
        L.check(L.lib().lnrf_comm_get_unique_id(buf), "comm_get_unique_id")
        return buf.raw

    @classmethod
    def from_process_group(cls) -> "AbiComm":
        d = dist_module()
        if d is None:
            return cls(cls.new_unique_id(), 0, 1)
        box = [cls.new_unique_id() if d.get_rank() == 0 else None]
        d.broadcast_object_list(box, src=0)
        return cls(box[0], d.get_rank(), d.get_world_size())

    @classmethod
    def from_file(cls, path: str, rank: int, world: int, nonce: str = "", timeout_s: float = 120.0) -> "AbiComm":
        """Bootstrap through a shared file.  `nonce` must be the same on every rank and different for every launch (e.g.
        MASTER_PORT or the launcher's pid): the id file is `path.<nonce>`, so an id left behind by an earlier launch can
        never be read; the file also carries the nonce and is removed by rank 0 once every rank has joined (the first
        collective of the communicator is the proof)."""
        path = f"{path}.{nonce}" if nonce else path
        tag = (nonce or "-").encode()
        if rank == 0:
            if os.path.exists(path):
                os.unlink(path)  # a leftover of a crashed launch with the same nonce
            with open(path + ".tmp", "wb") as fh:
                fh.write(len(tag).to_bytes(4, "little") + tag + cls.new_unique_id())
            os.replace(path + ".tmp", path)
        deadline = time.time() + timeout_s
        uid = None
        while uid is None:
            try:
                with open(path, "rb") as fh:
                    blob = fh.read()
                n = int.from_bytes(blob[:4], "little")
                if blob[4:4 + n] == tag and len(blob) == 4 + n + 128:
                    uid = blob[4 + n:]
            except FileNotFoundError:
                pass
            if uid is None:
                if time.time() > deadline:
                    raise TimeoutError(f"rank 0 never wrote the RCCL unique id of this launch to {path}")
                time.sleep(0.05)
        comm = cls(uid, rank, world)
        probe = torch.zeros(1, device="cuda")
        comm.all_reduce_sum_(probe)  # returns on every rank only after all of them have initialised
        torch.cuda.synchronize()
        if rank == 0:
            try:
                os.unlink(path)
            except FileNotFoundError:
                pass
        return comm

    def all_reduce_sum_(self, flat: torch.Tensor) -> torch.Tensor:
        L = self._L
        L.check(L.lib().lnrf_comm_allreduce(self._handle, L.ptr(flat), flat.numel(), L.stream()), "comm_allreduce")
        return flat

    def destroy(self) -> None:
        if self._handle is not None:
            self._L.check(self._L.lib().lnrf_comm_destroy(self._handle), "comm_destroy")
            self._handle = None


_abi_comm: Optional[AbiComm] = None


def use_abi_comm(comm: Optional[AbiComm]) -> None:
    """Route reduce_gradient_ through a C-ABI communicator (None: back to torch.distributed)."""
    global _abi_comm
    _abi_comm = comm


def reduce_gradient_(flat: torch.Tensor, reduced_prefix: int = 0, pending=None) -> float:
    """
    reduced_prefix / pending: the first `reduced_prefix` floats are already being reduced by begin_reduce_ (handle
    `pending`): only the rest is exchanged here, then the handle is waited for.

    The ONE exchange of a data-parallel step: in-place all-reduce (sum) of the flat gradient — over the C-ABI
    RCCL communicator when use_abi_comm() installed one, else over the default torch.distributed process group
    (RCCL on GPUs, gloo on CPU tensors), issued whenever a group exists, also with a single rank, so the collective
    path is exercised by every torchrun launch.  Returns the factor 1/world that the caller folds into Adam
    (grad_scale) and into the logged grad_norm.
    """
    if pending is None:
        reduced_prefix = 0
    rest = flat[reduced_prefix:] if reduced_prefix else flat
    if _abi_comm is not None:
        _abi_comm.all_reduce_sum_(rest)
        return 1.0 / _abi_comm.world
    d = dist_module()
    if d is None:
        return 1.0
    d.all_reduce(rest)
    if pending is not None:
        pending.wait()  # the compute stream waits for the early slice (gloo: the host does)
    return 1.0 / d.get_world_size()


def begin_reduce_(part: torch.Tensor):
    """
    Start the all-reduce (sum) of a finished slice of the gradient while the rest of the backward still runs: with
    torch.distributed the collective is issued asynchronously (RCCL's own stream, ordered after what the current stream
    has enqueued so far) and the returned handle is waited for right before the optimizer (reduce_gradient_(...,
    pending=handle)).  Returns None when there is nothing to overlap — no process group, or the C-ABI communicator,
    whose collective is stream-ordered on the compute stream: the caller then reduces the whole buffer at the end.
    """
    if _abi_comm is not None:
        return None
    d = dist_module()
    if d is None:
        return None
    return d.all_reduce(part, async_op=True)


def grad_scale() -> float:
    """1/world: applied inside the fused Adam kernel and to the logged grad_norm."""
    return 1.0 / world_info()[1]
