"""
north_star quality gate: "PSNR within 0.1 dB of the reference".  The reference trains and renders in fp32
(model.py:72, render.py:140); the product trains on bf16 MFMA operands and renders with the split-precision
kernel.  This test trains the synthetic cube scene (scripts/make_cube_dataset.py's analytic ray-caster, the
stand-in for BASELINE config 1's STL cube) twice from ONE seed and ONE batch order — fused bf16 path and the
exact-fp32 dense path (the oracle-validated fp32 restatement of the reference arithmetic) — renders held-out
views with each, and compares PSNR = -10 log10(mean(((out - target) / 2)^2)).

"parity unpinned" applies as everywhere: the fp32 leg is this build's fp32 path (validated against the float64
oracle to 5e-6), not JAX itself, which is absent.
"""
import math
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learn-nerf_amd", "learn_nerf", "scripts"))

SIZE, FOV, RADIUS = 32, math.radians(40.0), 2.5
BMIN, BMAX = (-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)
TC, TF, BATCH = 16, 32, 512


def cube_views(count, seed):
    """[(rays[H*W,3,3] fp32 incl. colours in [-1,1])] for `count` random cameras around the cube."""
    import make_cube_dataset as mk
    from learn_nerf.dataset import CameraView

    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        origin, x, y, z = mk.random_camera(rng, RADIUS)
        rgba = mk.render_cube(origin, x, y, z, FOV, SIZE).astype(np.float64)
        rgb = np.round(rgba[..., :3] * (rgba[..., 3:] / 255)).astype(np.uint8)  # dataset.py:108-111
        view = CameraView(camera_direction=tuple(z), camera_origin=tuple(origin), x_axis=tuple(x), y_axis=tuple(y),
                          x_fov=FOV, y_fov=FOV)
        geo = view.bare_rays(SIZE, SIZE).numpy()
        col = rgb.reshape(-1, 1, 3).astype(np.float32) / 127.5 - 1
        out.append(torch.from_numpy(np.concatenate([geo, col], axis=1).astype(np.float32)))
    return out


def held_out_psnr(loop, views):
    from learn_nerf.render import NeRFRenderer
    from learn_nerf.rng import Key

    p = loop.state.params
    renderer = NeRFRenderer(coarse=loop.coarse, fine=loop.fine, coarse_params=p["coarse"], fine_params=p["fine"],
                            background=p["background"], bbox_min=BMIN, bbox_max=BMAX, coarse_ts=TC, fine_ts=TF)
    se, cnt = 0.0, 0
    for i, rays in enumerate(views):
        out = renderer.render_rays(Key(1000 + i), rays[:, :2].contiguous().cuda())["fine"]["outputs"]
        se += float((((out.cpu().double() - rays[:, 2].double()) / 2) ** 2).sum())
        cnt += rays.shape[0] * 3
    return -10 * math.log10(se / cnt)


# Adam at a constant step size keeps the held-out PSNR jittering by +-0.5 dB from checkpoint to checkpoint (measured:
# the SIGN of bf16 - fp32 flips between checkpoints), far more than the 0.1 dB to be resolved.  The comparison is
# therefore made after annealing the step size (the reference has no schedule, train.py:59; TrainLoop.lr is a plain
# attribute) and on the mean squared error accumulated over the last EVAL_POINTS checkpoints.
SCHEDULE = ((900, 5e-4), (500, 1e-4), (300, 2e-5))
EVAL_EVERY, EVAL_POINTS = 50, 8

# Independent initialisations / batch orders.  Both arithmetics are bit-reproducible (tests/test_gpu_train_step.py::
# test_train_steps_are_bit_reproducible), so every per-seed difference is a fixed number for a given build; seeds are added
# until the standard error of their mean is at most SE_TARGET (measured: sample std 0.08-0.11 dB, i.e. 8-14 seeds of ~13 s).
MIN_SEEDS, MAX_SEEDS, SE_TARGET = 8, 24, 0.03


def train(precision, train_rays, test_views, key_offset=0, init_seed=5):
    from learn_nerf.model import NeRFModel
    from learn_nerf.rng import Key
    from learn_nerf.train import TrainLoop

    loop = TrainLoop(NeRFModel(precision=precision), NeRFModel(precision=precision), init_rng=init_seed,
                     lr=SCHEDULE[0][1],
                     coarse_ts=TC, fine_ts=TF)
    step = loop.step_fn(BMIN, BMAX)
    gen = torch.Generator().manual_seed(99 + init_seed)  # one batch order for the runs that are compared
    all_rays = train_rays.cuda()
    total = sum(n for n, _ in SCHEDULE)
    tail, i = [], 0
    for n_steps, lr in SCHEDULE:
        loop.lr = lr
        for _ in range(n_steps):
            idx = torch.randint(0, all_rays.shape[0], (BATCH,), generator=gen)
            step(Key(i + key_offset), all_rays[idx.cuda()].contiguous())
            i += 1
            if i > total - EVAL_EVERY * EVAL_POINTS and (total - i) % EVAL_EVERY == 0:
                tail.append(held_out_psnr(loop, test_views))
    assert len(tail) == EVAL_POINTS
    mean_mse = sum(10 ** (-p / 10) for p in tail) / len(tail)
    return -10 * math.log10(mean_mse), tail


def test_bf16_training_matches_fp32_psnr():
    """
    north_star: PSNR within 0.1 dB of the (fp32) reference.  Training is chaotic — two runs that differ in ANY rounding end
    0.1-0.2 dB apart — so one bf16 - fp32 difference cannot resolve 0.1 dB; the mean over independent seeds can.  Seeds
    are added until the standard error of the mean difference is <= 0.03 dB, then the gate is the plain one:
    |mean difference| <= 0.1 dB.
    """
    train_rays = torch.cat(cube_views(24, seed=0), dim=0)
    test_views = cube_views(8, seed=1234)
    deltas, se = [], float("inf")
    for k in range(MAX_SEEDS):
        init_seed = 100 + k
        bf16, _ = train("bf16", train_rays, test_views, init_seed=init_seed)
        fp32, _ = train("fp32", train_rays, test_views, init_seed=init_seed)
        print(f"seed {init_seed}: held-out PSNR (mean over the last {EVAL_POINTS} checkpoints) bf16-trained {bf16:.3f} dB, "
              f"fp32-trained {fp32:.3f} dB, delta {bf16 - fp32:+.3f} dB")
        assert fp32 > 25.0, "the scene must actually be learnt for the comparison to mean anything"
        assert abs(bf16 - fp32) < 0.75, "a single run this far off is not noise"
        deltas.append(bf16 - fp32)
        n = len(deltas)
        if n >= MIN_SEEDS:
            mean_delta = sum(deltas) / n
            std = math.sqrt(sum((d - mean_delta) ** 2 for d in deltas) / (n - 1))
            se = std / math.sqrt(n)
            if se <= SE_TARGET:
                break
    print(f"mean delta over {n} seeds: {mean_delta:+.3f} dB, sample std {std:.3f} dB, standard error {se:.3f} dB")
    assert se <= SE_TARGET, "the comparison lost its resolution"
    assert abs(mean_delta) <= 0.1  # north_star: PSNR within 0.1 dB
