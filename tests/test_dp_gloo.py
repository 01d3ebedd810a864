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


class CpuKernels:
    """CPU stand-in for the HIP kernel TrainLoop's update uses (ops.adam_step_ with fused norm accumulation), with
    the oracle's Adam (oracle/train.py) — so that train.apply_gradients, the PRODUCT's reduce -> 1/W -> Adam (+ norms)
    sequence, can run on CPU tensors under gloo."""

    @staticmethod
    def adam_step_(p, g, m, v, lr, b1, b2, eps, step, grad_scale=1.0, sq_norms=None):
        from oracle import train as OT

        if sq_norms is not None:  # what lnrf_adam_step_norms accumulates: the gradient as passed in, p before the update
            sq_norms[0] += float((g.double() ** 2).sum())
            sq_norms[1] += float((p.double() ** 2).sum())
        p2, m2, v2 = OT.adam_update(p.double(), g.double() * grad_scale, m.double(), v.double(), step, lr, b1, b2, eps)
        p.copy_(p2.to(p.dtype))
        m.copy_(m2.to(m.dtype))
        v.copy_(v2.to(v.dtype))


def _update_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from learn_nerf import parallel
    from learn_nerf.train import apply_gradients

    parallel.init_distributed(backend="gloo")
    cf, ff = _params()
    flat = torch.cat([cf, ff, torch.tensor([-1.0, -1.0, -1.0], dtype=F64)]).float()
    shard, offset = parallel.shard_rays(_batch(), rank, world)
    grad = _grad(shard, offset).float()  # this rank's gradient of ITS shard's mean loss
    m, v = torch.zeros_like(flat), torch.zeros_like(flat)
    logs = []
    for step in (1, 2):  # two steps: the second one sees non-zero moments
        g = grad.clone() * (1.0 if step == 1 else 0.5)
        sq = torch.zeros(2, dtype=torch.float64)
        if step == 1:  # one collective over the whole buffer
            scale = apply_gradients(flat, g, m, v, step, 1e-3, 0.9, 0.999, 1e-7, sq_norms=sq, kernels=CpuKernels)
        else:  # the product's overlapped form: the coarse slice starts early, the rest follows, same sums
            n_early = cf.numel()
            pending = parallel.begin_reduce_(g[:n_early])
            assert pending is not None
            scale = apply_gradients(flat, g, m, v, step, 1e-3, 0.9, 0.999, 1e-7, sq_norms=sq, kernels=CpuKernels,
                                    reduced_prefix=n_early, pending=pending)
        assert scale == 1.0 / world
        logs.append(dict(grad_norm=float(sq[0].sqrt()) * scale, param_norm=float(sq[1].sqrt())))
    torch.save(dict(flat=flat, m=m, v=v, logs=logs), os.path.join(out_dir, f"u{rank}.pt"))
    import torch.distributed as dist

    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_product_update_equals_global_batch_step(tmp_path):
    """train.apply_gradients (what TrainLoop._step runs after the backward) under gloo, world_size 2:
    all-reduce(sum) -> grad_norm * 1/W -> Adam with grad_scale 1/W  ==  the single-process step on the
    global batch (train.py:99-106)."""
    from oracle import train as OT

    world = 2
    mp.spawn(_update_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "u0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "u1.pt"))
    for k in ("flat", "m", "v"):
        assert torch.equal(r0[k], r1[k]), f"ranks disagree on {k} after the update"
    assert r0["logs"] == r1["logs"]
    # single process, global batch
    cf, ff = _params()
    p = torch.cat([cf, ff, torch.tensor([-1.0, -1.0, -1.0], dtype=F64)]).float().double()
    full = _grad(_batch(), 0).float().double()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step, logs in zip((1, 2), r0["logs"]):
        g = full * (1.0 if step == 1 else 0.5)
        assert abs(logs["grad_norm"] - float(g.norm())) < 1e-5 * float(g.norm())
        assert abs(logs["param_norm"] - float(p.norm())) < 1e-5 * float(p.norm())
        p, m, v = OT.adam_update(p, g, m, v, step, 1e-3, 0.9, 0.999, 1e-7)
    # fp32 state on the ranks vs float64 here: Adam normalises steps to ~lr, so compare at a fraction of lr
    assert (r0["flat"].double() - p).abs().max().item() < 2e-5
    assert torch.allclose(r0["m"].double(), m, rtol=1e-4, atol=1e-9)


def _shuffle_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    import numpy as np

    from learn_nerf import parallel
    from learn_nerf.dataset import ModelMetadata, NeRFDataset, NeRFView

    class View(NeRFView):
        def __init__(self, img, **kw):
            super().__init__(**kw)
            self._img = img

        def image(self):
            return self._img

    rng = np.random.default_rng(0)
    views = [View((rng.random((40, 40, 3)) * 255).astype(np.uint8), camera_direction=(0.0, 1.0, 0.0),
                  camera_origin=(float(i), 2.0, 2.0), x_axis=(-1.0, 0.0, 0.0), y_axis=(0.0, 0.0, 1.0), x_fov=1.0,
                  y_fov=1.0) for i in range(6)]
    parallel.init_distributed(backend="gloo")
    ds = NeRFDataset(metadata=ModelMetadata((0.0, 0.0, 0.0), (1.0, 1.0, 1.0)), views=views)
    it = ds.iterate_batches(os.path.join(out_dir, "shuffled"), 77, batch_size=333, repeat=True)
    got = torch.stack([next(it) for _ in range(40)])  # 40 x 333 rays: more than one epoch of 9600
    torch.save(got, os.path.join(out_dir, f"b{rank}.pt"))
    import torch.distributed as dist

    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_share_a_fresh_shuffle_dir(tmp_path):
    """on a fresh dataset every rank used to truncate and rewrite the same 32 shard files.  Now rank 0
    alone deals the rays (temporary names, renamed, marker last) and the others wait: both ranks must read
    identical, complete batches from an initially EMPTY shuffle directory."""
    world = 2
    mp.spawn(_shuffle_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    b0 = torch.load(os.path.join(tmp_path, "b0.pt"))
    b1 = torch.load(os.path.join(tmp_path, "b1.pt"))
    assert torch.equal(b0, b1)
    first_epoch = b0.reshape(-1, 9)[:9600]
    assert len({tuple(r.tolist()) for r in first_epoch}) == 9600, "an epoch visits every ray exactly once"
    names = sorted(os.listdir(tmp_path / "shuffled"))
    assert len(names) == 33 and not any(n.endswith(".tmp") for n in names)


def test_shard_bounds_and_errors():
    from learn_nerf import parallel

    assert parallel.shard_bounds(4096, 3, 8) == (1536, 2048)
    with pytest.raises(ValueError):
        parallel.shard_bounds(10, 0, 3)
    assert parallel.world_info() == (0, 1) and parallel.grad_scale() == 1.0
