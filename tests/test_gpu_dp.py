"""
Data-parallel step on the GPU: the product's TrainLoop._step with two ranks sharing one
GPU (gloo process group carrying device tensors) must equal the single-process step on the global batch — the
sharding, Philox ray offsets, all-reduce(sum), grad_norm * 1/W and Adam's grad_scale = 1/W all in the loop.
The exact-fp32 model path is used so that the comparison is tight.  Also: the RCCL entries of the C ABI
(lnrf_comm_*) with a one-rank communicator, and the 8,192-rays-per-GPU point of BASELINE configs[4].
"""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, TC, TF, SEED, LR = 64, 16, 32, 17, 1e-3
BMIN, BMAX = (-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)


def global_batch():
    gen = torch.Generator().manual_seed(3)
    o = torch.randn(N, 3, generator=gen)
    o = 4 * o / o.norm(dim=-1, keepdim=True)
    d = -o + (torch.rand(N, 3, generator=gen) - 0.5) * 0.6
    d = d / d.norm(dim=-1, keepdim=True)
    c = torch.rand(N, 3, generator=gen) * 2 - 1
    return torch.stack([o, d, c], 1).float().contiguous()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(world, out_dir):
    port = str(_free_port())
    worker = os.path.join(ROOT, "tests", "dp_gpu_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), port, out_dir], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out[-3000:]
    return [torch.load(os.path.join(out_dir, f"r{world}_{r}.pt")) for r in range(world)]


@pytest.mark.timeout(900)
def test_two_rank_train_step_equals_global_batch_step(tmp_path):
    single = _run(1, str(tmp_path))[0]
    r0, r1 = _run(2, str(tmp_path))
    assert torch.equal(r0["flat"], r1["flat"]) and torch.equal(r0["m"], r1["m"]), "ranks diverged"
    for a, b in zip(r0["logs"], r1["logs"]):  # the norm kernel combines per-block partial sums by fp32 atomics
        assert abs(a["grad_norm"] - b["grad_norm"]) < 1e-6 * a["grad_norm"]
        assert abs(a["param_norm"] - b["param_norm"]) < 1e-6 * a["param_norm"]
    for it, (dp, ref) in enumerate(zip(r0["logs"], single["logs"])):
        print(f"step {it}: grad_norm dp {dp['grad_norm']:.6f} single {ref['grad_norm']:.6f}; "
              f"rank-0 shard loss {dp['fine']:.5f}, global loss {ref['fine']:.5f}")
        assert abs(dp["grad_norm"] - ref["grad_norm"]) < 2e-4 * ref["grad_norm"]
        assert abs(dp["param_norm"] - ref["param_norm"]) < 1e-6 * ref["param_norm"]
    # first moment after step 2 = a linear function of both steps' averaged gradients
    rel_m = ((r0["m"] - single["m"]).norm() / single["m"].norm()).item()
    upd = (r0["flat"] - single["flat"]).abs().max().item()
    print(f"Adam first moment rel diff {rel_m:.2e}, max parameter diff {upd:.2e} (lr {LR})")
    assert rel_m < 1e-3
    assert upd <= 2.01 * LR  # Adam normalises steps to ~lr: only sign flips of ~zero gradients may differ


def test_comm_abi_single_rank_and_errors():
    import ctypes

    from learn_nerf import _lib as L
    from learn_nerf import parallel

    comm = parallel.AbiComm.from_process_group()  # no process group here: a one-rank RCCL communicator
    assert (comm.rank, comm.world) == (0, 1)
    rank, world = ctypes.c_int32(-1), ctypes.c_int32(-1)
    L.check(L.lib().lnrf_comm_info(comm._handle, ctypes.byref(rank), ctypes.byref(world)))
    assert (rank.value, world.value) == (0, 1)
    g = torch.randn(1_187_851, device="cuda")
    before = g.clone()
    parallel.use_abi_comm(comm)
    try:
        scale = parallel.reduce_gradient_(g)
    finally:
        parallel.use_abi_comm(None)
    torch.cuda.synchronize()
    assert scale == 1.0 and torch.equal(g, before)  # sum over one rank
    comm.destroy()
    assert L.lib().lnrf_comm_allreduce(None, L.ptr(g), g.numel(), L.stream()) == -1
    assert b"null communicator" in L.lib().lnrf_last_error()
    assert L.lib().lnrf_comm_init(None, 0, 1, None) == -1
    with pytest.raises(ValueError):
        parallel.AbiComm(b"short", 0, 1)


def test_train_step_at_8192_rays_per_gpu():
    """BASELINE configs[4]: 65,536 rays over 8 GPUs = 8,192 rays per GPU per step (64 + 128 samples)."""
    from learn_nerf.model import NeRFModel
    from learn_nerf.rng import Key
    from learn_nerf.train import TrainLoop

    sys.path.insert(0, ROOT)
    from bench import synthetic_batch

    loop = TrainLoop(NeRFModel(), NeRFModel(), init_rng=0, lr=1e-4, coarse_ts=64, fine_ts=128)
    batch = synthetic_batch(8192, 1, torch.device("cuda"))
    step = loop.step_fn(BMIN, BMAX)
    first = step(Key(0), batch)
    for i in range(1, 8):
        last = step(Key(i), batch)
    torch.cuda.synchronize()
    assert all(torch.isfinite(v) for v in last.values())
    assert float(last["fine"]) < float(first["fine"])
    # the same rays in two half batches give the same per-ray outputs (rays are independent units)
    from learn_nerf.render import NeRFRenderer

    p = loop.state.params
    r = NeRFRenderer(coarse=loop.coarse, fine=loop.fine, coarse_params=p["coarse"], fine_params=p["fine"],
                     background=p["background"], bbox_min=BMIN, bbox_max=BMAX, coarse_ts=64, fine_ts=128)
    rays = batch[:, :2].contiguous()
    whole = r.render_rays(Key(5), rays)["fine"]["outputs"]
    half = r.render_rays(Key(5, ray_offset=4096), rays[4096:].contiguous())["fine"]["outputs"]
    assert torch.equal(whole[4096:], half)


@pytest.mark.parametrize("backward", ["ls", "split"])
def test_single_rank_rccl_group_step_equals_plain_step(backward):
    """The data-parallel sequence over RCCL itself (backend "nccl", one rank on this GPU): the step with a process group —
    all-reduce of the flat gradient on RCCL's stream, for the two-launch backward also the EARLY asynchronous reduce of
    the coarse slice underneath the fine backward (train.py: parallel.begin_reduce_ / reduce_gradient_) — must leave
    bit-identical parameters and Adam moments to the step without a group.  Covers the stream ordering of the async
    collective against grad.zero_(), the backward kernels and Adam, which the gloo runs cannot."""
    import torch.distributed as dist

    from learn_nerf.model import NeRFModel
    from learn_nerf.rng import Key
    from learn_nerf.train import TrainLoop

    batch = global_batch().cuda()

    def run():
        loop = TrainLoop(NeRFModel(backward_kernel=backward), NeRFModel(backward_kernel=backward), init_rng=SEED, lr=LR,
                         coarse_ts=TC, fine_ts=TF)
        step = loop.step_fn(BMIN, BMAX)
        for i in range(3):
            step(Key(i), batch)
        torch.cuda.synchronize()
        return loop.flat.clone(), loop.state.opt_m.clone(), loop.state.opt_v.clone()

    plain = run()
    assert not dist.is_initialized()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        grouped = run()
    finally:
        dist.destroy_process_group()
    for a, b in zip(plain, grouped):
        assert torch.equal(a, b)


def test_abi_comm_file_bootstrap_ignores_a_stale_id(tmp_path):
    """AbiComm.from_file: an id file left by an earlier launch (other nonce, or the same path without one) must not be
    picked up, and the file of this launch is gone once the communicator works."""
    from learn_nerf.parallel import AbiComm

    base = str(tmp_path / "uid")
    with open(base + ".old", "wb") as fh:
        fh.write(b"\\x03\\x00\\x00\\x00old" + b"\\x00" * 128)
    with open(base + ".new", "wb") as fh:  # same name as this launch's file, but written by "an earlier crash"
        fh.write(b"\\x03\\x00\\x00\\x00xyz" + b"\\x00" * 128)
    comm = AbiComm.from_file(base, 0, 1, nonce="new")
    v = torch.arange(8, dtype=torch.float32, device="cuda")
    comm.all_reduce_sum_(v)
    torch.cuda.synchronize()
    assert torch.equal(v.cpu(), torch.arange(8, dtype=torch.float32))
    comm.destroy()
    assert not os.path.exists(base + ".new") and os.path.exists(base + ".old")
