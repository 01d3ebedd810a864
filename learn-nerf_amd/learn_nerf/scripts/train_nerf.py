"""
Train a NeRF model on a scene (reference: learn_nerf/scripts/train_nerf.py).

Same flags, defaults, positional argument, stdout line format (``step {i}: k=v ...``), auto-resume
and save cadence as the reference; the step itself runs in the HIP kernels.  Additive flags:
--precision {bf16,fp32} and, for multi-GPU runs under torchrun, data-parallel sharding of each batch.
"""
import argparse
import os
import random
from functools import partial
from typing import Any, Dict, Tuple

import torch

from learn_nerf import parallel
from learn_nerf.dataset import ModelMetadata, load_dataset
from learn_nerf.model import ModelBase, NeRFModel
from learn_nerf.rng import Key
from learn_nerf.train import TrainLoop


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--seed", type=int, default=None)
    parser.add_argument("--lr", type=float, default=1e-4)
    parser.add_argument("--batch_size", type=int, default=4096, help="rays per batch")
    parser.add_argument("--test_batch_size", type=int, default=None, help="rays per test batch")
    parser.add_argument("--coarse_samples", type=int, default=64, help="samples per coarse ray")
    parser.add_argument("--fine_samples", type=int, default=128,
                        help="samples per fine ray (not including coarse samples)")
    parser.add_argument("--density_penalty", type=float, default=None,
                        help="penalty coefficient for density at random points")
    parser.add_argument("--density_penalty_batch_size", type=int, default=128,
                        help="batch size for computing density penalty")
    parser.add_argument("--save_interval", type=int, default=1000)
    parser.add_argument("--save_path", type=str, default="nerf.pkl")
    parser.add_argument("--one_view", action="store_true")
    parser.add_argument("--test_data_dir", type=str, default=None)
    parser.add_argument("--max_steps", type=int, default=None, help="(additive) stop after this many steps")
    add_model_args(parser)
    parser.add_argument("data_dir", type=str)
    args = parser.parse_args()

    if args.test_batch_size is None:
        args.test_batch_size = args.batch_size

    rank, local_rank, world = parallel.init_distributed()
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    log = print if rank == 0 else (lambda *a, **k: None)

    log("loading dataset...")
    data = load_dataset(args.data_dir)
    if args.one_view:
        data.views = data.views[:1]
    if args.test_data_dir is not None:
        log("loading test dataset...")
        test_data = load_dataset(args.test_data_dir)
        if args.one_view:
            test_data.views = test_data.views[:1]
    else:
        test_data = None

    seed = args.seed if args.seed is not None else random.randint(0, 2 ** 32 - 1)
    if world > 1:  # every rank must draw the same batches and initial parameters
        t = torch.tensor([seed], dtype=torch.int64, device="cuda")
        torch.distributed.broadcast(t, src=0)
        seed = int(t.item())
    init_key, key = Key(seed).split(2)

    log("creating model and train loop...")
    coarse, fine, train_kwargs = create_model(args, data.metadata)
    loop = TrainLoop(coarse, fine, init_rng=init_key.seed, lr=args.lr, coarse_ts=args.coarse_samples,
                     fine_ts=args.fine_samples, density_penalty=args.density_penalty,
                     density_penalty_batch_size=args.density_penalty_batch_size, **train_kwargs)
    if os.path.exists(args.save_path):
        log(f"loading from checkpoint: {args.save_path}")
        loop.load(args.save_path)
    step_fn = loop.step_fn(data.metadata.bbox_min, data.metadata.bbox_max)

    log("training...")
    data_key, test_data_key, key = key.split(3)
    shuffle_dir = os.path.join(args.data_dir, "shuffled")
    if test_data:
        test_shuffle_dir = os.path.join(args.test_data_dir, "shuffled")
        test_iterator = test_data.iterate_batches(test_shuffle_dir, test_data_key.seed, args.test_batch_size)
    for i, batch in enumerate(data.iterate_batches(shuffle_dir, data_key.seed, args.batch_size)):
        step_key, test_key, key = key.split(3)
        shard, offset = parallel.shard_rays(batch, rank, world)
        if test_data is not None:
            test_batch = next(test_iterator).to(loop.device)
            _, tl = loop.losses(test_key, data.metadata.bbox_min, data.metadata.bbox_max, test_batch)
            test_losses = {f"test_{k}": v for k, v in tl.items()}
        losses = step_fn(Key(step_key.seed, ray_offset=offset), shard.to(loop.device, non_blocking=True))
        if test_data is not None:
            losses.update(test_losses)
        loss_str = " ".join(f"{k}={float(v):.05}" for k, v in losses.items())
        log(f"step {i}: {loss_str}")
        if i and i % args.save_interval == 0 and rank == 0:
            loop.save(args.save_path)
        if args.max_steps is not None and i + 1 >= args.max_steps:
            if rank == 0:
                loop.save(args.save_path)
            break


def add_model_args(parser: argparse.ArgumentParser):
    parser.add_argument("--instant_ngp", action="store_true")
    parser.add_argument("--ref_nerf", action="store_true")
    parser.add_argument("--precision", choices=["bf16", "fp32"], default="bf16",
                        help="(additive) NeRFModel arithmetic: fused bf16 MFMA or exact fp32")


def create_model(args: argparse.Namespace, metadata: ModelMetadata) -> Tuple[ModelBase, ModelBase, Dict[str, Any]]:
    """Model hyper-parameters exactly as scripts/train_nerf.py:141-170."""
    if args.instant_ngp:
        from learn_nerf.instant_ngp import InstantNGPModel, InstantNGPRefNERFModel

        if args.ref_nerf:
            model_cls = partial(InstantNGPRefNERFModel, sh_degree=4)
        else:
            model_cls = InstantNGPModel
        coarse = model_cls(table_sizes=[2 ** 18] * 6, grid_sizes=[2 ** (4 + i // 2) for i in range(6)],
                           bbox_min=tuple(metadata.bbox_min), bbox_max=tuple(metadata.bbox_max))
        fine = model_cls(table_sizes=[2 ** 18] * 16, grid_sizes=[2 ** (4 + i // 2) for i in range(16)],
                         bbox_min=tuple(metadata.bbox_min), bbox_max=tuple(metadata.bbox_max))
        train_kwargs = dict(adam_eps=1e-15, adam_b1=0.9, adam_b2=0.99)
    else:
        if args.ref_nerf:
            from learn_nerf.ref_nerf import RefNERFModel

            model_cls = partial(RefNERFModel, sh_degree=4)
        else:
            model_cls = partial(NeRFModel, precision=getattr(args, "precision", "bf16"))
        coarse = model_cls()
        fine = model_cls()
        train_kwargs = dict()
    return coarse, fine, train_kwargs


if __name__ == "__main__":
    main()
