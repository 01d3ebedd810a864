"""
Helper of tests/test_gpu_dp.py: one rank of a 2-rank data-parallel run of the PRODUCT train step
(TrainLoop._step: HIP forward/backward on this rank's ray shard -> all-reduce -> 1/world -> fused Adam) on one GPU.
The process group is gloo with device tensors (two ranks may share a GPU under gloo, not under RCCL); everything
else is the code path a torchrun launch uses.  argv: rank world port out_dir
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "learn-nerf_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch
    import torch.distributed as dist

    from learn_nerf import parallel
    from learn_nerf.model import NeRFModel
    from learn_nerf.rng import Key
    from learn_nerf.train import TrainLoop
    from test_gpu_dp import BMAX, BMIN, LR, N, SEED, TC, TF, global_batch

    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    loop = TrainLoop(NeRFModel(precision="fp32"), NeRFModel(precision="fp32"), init_rng=SEED, lr=LR, coarse_ts=TC,
                     fine_ts=TF)
    step = loop.step_fn(BMIN, BMAX)
    batch = global_batch().cuda()
    logs = []
    for it in range(2):
        mine, first = parallel.shard_rays(batch, rank, world)
        log = step(Key(100 + it, ray_offset=first), mine)
        logs.append({k: float(v) for k, v in log.items()})
    torch.save(dict(flat=loop.flat.cpu(), m=loop.state.opt_m.cpu(), logs=logs), os.path.join(out_dir, f"r{world}_{rank}.pt"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
