"""
Data-parallel path on CPU (gloo, world_size 2): the sharding / all-reduce / 1-over-world scaling
used by TrainLoop reproduces the single-process gradient of the global batch, and the Philox
ray_offset convention makes sharded sampling noise identical to the unsharded run.
The per-rank arithmetic here is the oracle (no GPU in this test); the collective plumbing is the
product's learn_nerf.parallel module.
"""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "learn-nerf_amd"))

F64 = torch.float64
N, TC, TF, SEED = 16, 8, 8, 99
HID, COL = 32, 16


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _batch():
    gen = torch.Generator().manual_seed(0)
    o = torch.randn(N, 3, generator=gen, dtype=F64)
    o = 4 * o / o.norm(dim=-1, keepdim=True)
    d = -o + (torch.rand(N, 3, generator=gen, dtype=F64) - 0.5) * 0.6
    d = d / d.norm(dim=-1, keepdim=True)
    c = torch.rand(N, 3, generator=gen, dtype=F64) * 2 - 1
    return torch.stack([o, d, c], 1)


def _params():
    from oracle import model as OM

    gen = torch.Generator().manual_seed(1)
    dims = OM.nerf_layer_dims(hidden_dim=HID, color_layer_dim=COL)
    return OM.lecun_normal_init(dims, gen, dtype=F64), OM.lecun_normal_init(dims, gen, dtype=F64)


def _grad(batch, ray_offset):
    """oracle gradient of TrainLoop.losses for `batch`, sampling noise from the Philox streams"""
    from oracle import model as OM
    from oracle import philox
    from oracle import train as OT

    cf, ff = _params()
    bg = torch.tensor([-1.0, -1.0, -1.0], dtype=F64)
    n = batch.shape[0]
    uc = torch.from_numpy(philox.ray_uniforms(SEED, 0, ray_offset, n, TC)).double()
    uf = torch.from_numpy(philox.ray_uniforms(SEED + 1, 1, ray_offset, n, TF)).double()
    mk = lambda fl: OM.make_nerf_fn(fl, hidden_dim=HID, color_layer_dim=COL)
    _, _, _, grads = OT.nerf_train_step(mk, cf, ff, bg, None, 1, 1e-3, torch.tensor([-1.0] * 3, dtype=F64),
                                        torch.tensor([1.0] * 3, dtype=F64), batch, TC, TF, uc, uf)
    return torch.cat([g.reshape(-1) for g in grads])


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from learn_nerf import parallel

    r, _, w = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world) and parallel.world_info() == (rank, world)
    shard, offset = parallel.shard_rays(_batch(), rank, world)
    assert shard.shape[0] == N // world and offset == rank * (N // world)
    g = _grad(shard, offset)
    parallel.all_reduce_sum_(g)
    g *= parallel.grad_scale()  # what lnrf_adam_step does with grad_scale = 1/world
    torch.save(g, os.path.join(out_dir, f"g{rank}.pt"))
    import torch.distributed as dist

    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_equals_global_batch(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g0 = torch.load(os.path.join(tmp_path, "g0.pt"))
    g1 = torch.load(os.path.join(tmp_path, "g1.pt"))
    assert torch.equal(g0, g1), "all ranks must hold the same reduced gradient"
    full = _grad(_batch(), 0)
    assert torch.allclose(g0, full, rtol=1e-9, atol=1e-12)


def test_shard_bounds_and_errors():
    from learn_nerf import parallel

    assert parallel.shard_bounds(4096, 3, 8) == (1536, 2048)
    with pytest.raises(ValueError):
        parallel.shard_bounds(10, 0, 3)
    assert parallel.world_info() == (0, 1) and parallel.grad_scale() == 1.0
