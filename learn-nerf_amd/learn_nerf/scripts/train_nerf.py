"""
Training CLI with the command line of the reference's scripts/train_nerf.py (same flag names, defaults,
positional `data_dir`, `step {i}: k=v ...` log lines, resume-if-checkpoint-exists, periodic save), driving
the HIP train step.  Extra, additive flags: --precision, --max_steps, --table_log2.  Under torchrun every global batch is
sharded over the ranks and gradients are all-reduced over RCCL.
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

# (flag, type, default, help) — values as in scripts/train_nerf.py:21-55
TRAIN_FLAGS = (
    ("--seed", int, None, None),
    ("--lr", float, 1e-4, None),
    ("--batch_size", int, 4096, "rays per batch"),
    ("--test_batch_size", int, None, "rays per test batch"),
    ("--coarse_samples", int, 64, "samples per coarse ray"),
    ("--fine_samples", int, 128, "samples per fine ray (not including coarse samples)"),
    ("--density_penalty", float, None, "penalty coefficient for density at random points"),
    ("--density_penalty_batch_size", int, 128, "batch size for computing density penalty"),
    ("--save_interval", int, 1000, None),
    ("--save_path", str, "nerf.pkl", None),
    ("--test_data_dir", str, None, None),
    ("--max_steps", int, None, "(additive) stop, and save, after this many steps"),
)


def add_model_args(parser: argparse.ArgumentParser):
    """Model selection switches shared with render_nerf.py (scripts/train_nerf.py:136-138)."""
    for switch in ("--instant_ngp", "--ref_nerf"):
        parser.add_argument(switch, action="store_true")
    parser.add_argument("--precision", choices=("bf16", "fp32"), default="bf16",
                        help="(additive) NeRFModel arithmetic: fused bf16 MFMA kernels or exact fp32")
    parser.add_argument("--table_log2", type=int, default=18,
                        help="(additive) --instant_ngp: log2 of the hash-table size per level; the reference "
                             "hard-codes 2**18 (scripts/train_nerf.py:150-161), BASELINE configs[2] uses 19")


def create_model(args: argparse.Namespace, metadata: ModelMetadata) -> Tuple[ModelBase, ModelBase, Dict[str, Any]]:
    """
    (coarse, fine, extra TrainLoop kwargs) with the hyper-parameters hard-coded by the reference
    (scripts/train_nerf.py:141-170): hash grids of 6 / 16 levels, 2^18 entries (--table_log2), grid 2^(4 + i//2),
    Adam(0.9, 0.99, eps 1e-15) for --instant_ngp; sh_degree 4 for --ref_nerf.
    """
    use_ref = bool(getattr(args, "ref_nerf", False))
    precision = getattr(args, "precision", "bf16")
    if getattr(args, "instant_ngp", False):
        from learn_nerf.instant_ngp import InstantNGPModel, InstantNGPRefNERFModel

        factory = partial(InstantNGPRefNERFModel, sh_degree=4) if use_ref else InstantNGPModel
        box = dict(bbox_min=tuple(metadata.bbox_min), bbox_max=tuple(metadata.bbox_max), precision=precision)
        table = 2 ** int(getattr(args, "table_log2", 18))
        pair = [factory(table_sizes=[table] * levels, grid_sizes=[2 ** (4 + i // 2) for i in range(levels)], **box)
                for levels in (6, 16)]
        return pair[0], pair[1], dict(adam_eps=1e-15, adam_b1=0.9, adam_b2=0.99)
    if use_ref:
        from learn_nerf.ref_nerf import RefNERFModel

        return RefNERFModel(sh_degree=4, precision=precision), RefNERFModel(sh_degree=4, precision=precision), {}
    return NeRFModel(precision=precision), NeRFModel(precision=precision), {}


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description=__doc__)
    for flag, typ, default, text in TRAIN_FLAGS:
        parser.add_argument(flag, type=typ, default=default, help=text)
    parser.add_argument("--one_view", action="store_true")
    add_model_args(parser)
    parser.add_argument("data_dir", type=str)
    return parser


def _maybe_first_view_only(dataset, one_view: bool):
    if one_view:
        dataset.views = dataset.views[:1]
    return dataset


def main():
    args = build_parser().parse_args()
    test_bs = args.test_batch_size if args.test_batch_size is not None else args.batch_size

    rank, local_rank, world = parallel.init_distributed()
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    say = print if rank == 0 else (lambda *a, **k: None)

    say("loading dataset...")
    data = _maybe_first_view_only(load_dataset(args.data_dir), args.one_view)
    test_data = None
    if args.test_data_dir is not None:
        say("loading test dataset...")
        test_data = _maybe_first_view_only(load_dataset(args.test_data_dir), args.one_view)
    bbox = (data.metadata.bbox_min, data.metadata.bbox_max)

    seed = args.seed if args.seed is not None else random.randint(0, 2 ** 32 - 1)
    if world > 1:  # all ranks must agree on the initial parameters and on the batch order
        shared = torch.tensor([seed], dtype=torch.int64, device="cuda")
        torch.distributed.broadcast(shared, src=0)
        seed = int(shared.item())
    init_key, key = Key(seed).split(2)

    say("creating model and train loop...")
    coarse, fine, extra = create_model(args, data.metadata)
    loop = TrainLoop(coarse, fine, init_rng=init_key.seed, lr=args.lr, coarse_ts=args.coarse_samples,
                     fine_ts=args.fine_samples, density_penalty=args.density_penalty,
                     density_penalty_batch_size=args.density_penalty_batch_size, **extra)
    if os.path.exists(args.save_path):
        say(f"loading from checkpoint: {args.save_path}")
        loop.load(args.save_path)
    step_fn = loop.step_fn(*bbox)

    say("training...")
    data_key, test_data_key, key = key.split(3)
    test_batches = None
    if test_data is not None:
        test_batches = test_data.iterate_batches(os.path.join(args.test_data_dir, "shuffled"), test_data_key.seed,
                                                 test_bs, device=loop.device)
    # shards are uploaded once and every batch is gathered on the GPU (dataset.ShuffledDataset._iterate_on_device)
    batches = data.iterate_batches(os.path.join(args.data_dir, "shuffled"), data_key.seed, args.batch_size,
                                   device=loop.device)
    for i, batch in enumerate(batches):
        step_key, test_key, key = key.split(3)
        report = {}
        if test_batches is not None:  # evaluated before the update, like the reference
            _, held_out = loop.losses(test_key, *bbox, next(test_batches).to(loop.device))
            report = {f"test_{name}": value for name, value in held_out.items()}
        mine, first_ray = parallel.shard_rays(batch, rank, world)
        losses = step_fn(Key(step_key.seed, ray_offset=first_ray), mine.to(loop.device, non_blocking=True))
        losses.update(report)
        say(f"step {i}: " + " ".join(f"{name}={float(value):.05}" for name, value in losses.items()))
        last = args.max_steps is not None and i + 1 >= args.max_steps
        if rank == 0 and ((i and i % args.save_interval == 0) or last):
            loop.save(args.save_path)
        if last:
            break


if __name__ == "__main__":
    main()
