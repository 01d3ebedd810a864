#!/usr/bin/env python
"""
bench.py — throughput of the NeRF train step (learn_nerf/train.py:78-112 restated as HIP kernels)
on N GPUs of one node.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1], the metric's own config): vanilla NeRF, 4096 rays per GPU,
64 coarse + 128 fine samples per ray (192 ray-samples per ray, 256 MLP evaluations per ray),
bf16 MFMA with fp32 accumulate / master weights, synthetic rays aimed at bbox [-1,1]^3, random
targets, Flax-default initial weights.  One step = forward + backward + [RCCL all-reduce] + Adam.
Rays shard data-parallel: every rank draws its own 4096 rays (weak scaling), gradients are
all-reduced (sum) over RCCL and averaged inside the fused Adam kernel.

Rank 0 prints ONE JSON line (see the task contract) with two extra objects:
  roofline     — the dominant kernel family, algorithmic FLOPs / measured HIP-event time vs the
                 dense bf16 MFMA peak of MI355X (2.5 PFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md)
  cpu_baseline — the oracle's torch-CPU fp32 restatement of the same step on a bounded sample
                 (rank 0, N = 1 only); a reported baseline, not the optimisation target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "learn-nerf_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

RAYS_PER_GPU = 4096
COARSE, FINE = 64, 128
BBOX_MIN, BBOX_MAX = (-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)
PEAK_BF16_FLOPS = 2.5e15  # dense bf16 MFMA, MI355X
FLOP_FWD_PER_EVAL = 2 * 591_488
FLOP_TRAIN_PER_RAY_SAMPLE = 4_641_792
MAC_DGRAD_PER_EVAL = 557_696
MAC_WGRAD_PER_EVAL = 591_488


def synthetic_batch(n, seed, device):
    gen = torch.Generator().manual_seed(seed)
    o = torch.randn(n, 3, generator=gen)
    o = 4 * o / o.norm(dim=-1, keepdim=True)  # cameras on a radius-4 sphere (Blender-Lego-like)
    d = -o + (torch.rand(n, 3, generator=gen) - 0.5) * 0.6
    d = d / d.norm(dim=-1, keepdim=True)
    c = torch.rand(n, 3, generator=gen) * 2 - 1
    return torch.stack([o, d, c], 1).float().contiguous().to(device)


def cpu_baseline(n_rays=1024, steps=4):
    """Oracle (torch-CPU fp32 restatement of the reference step) on the host cores."""
    from oracle import model as OM
    from oracle import train as OT

    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(16, avail))  # the 1-GPU box grants a 16-core share; more threads oversubscribe it
    torch.set_num_threads(threads)
    gen = torch.Generator().manual_seed(0)
    dims = OM.nerf_layer_dims()
    cf = OM.lecun_normal_init(dims, gen)
    ff = OM.lecun_normal_init(dims, gen)
    bg = torch.tensor([-1.0, -1.0, -1.0])
    batch = synthetic_batch(n_rays, 0, "cpu")
    uc = torch.rand(n_rays, COARSE, generator=gen)
    uf = torch.rand(n_rays, FINE, generator=gen)
    bmin, bmax = torch.tensor(BBOX_MIN), torch.tensor(BBOX_MAX)
    opt = None
    p = (cf, ff, bg)
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        p, opt, _, _ = OT.nerf_train_step(lambda fl: OM.make_nerf_fn(fl), p[0], p[1], p[2], opt, it + 1, 1e-4, bmin,
                                          bmax, batch, COARSE, FINE, uc, uf)
        dt = time.perf_counter() - t0
        if it > 0:
            times.append(dt)
    sec = sum(times) / len(times)
    return dict(value=n_rays * (COARSE + FINE) / sec, unit="ray-samples/s", cores=threads, kind="port",
                sample=f"{n_rays} rays x ({COARSE}+{FINE}) samples, fp32 torch-CPU restatement of the reference "
                       f"step (oracle/), mean of {steps} steps after 1 warm-up, {sec:.2f} s/step")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=RAYS_PER_GPU, help="rays per GPU")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--workload", default="nerf", choices=["nerf", "ngp", "refnerf"],
                    help="nerf = BASELINE configs[1] (the metric's config, default); ngp = configs[2] hash-grid path; "
                         "refnerf = configs[3] RefNERFModel on the generic dense path (no roofline model: reported as null)")
    ap.add_argument("--table_log2", type=int, default=19, help="ngp: log2 of the hash table size (configs[2]: 19)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true")
    args = ap.parse_args()

    # Libraries (e.g. RCCL's start-up banner) write to the C-level stdout; keep stdout clean for the one
    # JSON line by pointing fd 1 at stderr until the result is printed.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:  # launched by torch.distributed.run
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)  # "nccl" is RCCL on ROCm
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE", file=sys.stderr)

    from learn_nerf import _prof
    from learn_nerf.model import NeRFModel
    from learn_nerf.train import TrainLoop

    n = args.rays
    if args.workload == "ngp":
        from learn_nerf.instant_ngp import InstantNGPModel

        def ngp(levels):  # scripts/train_nerf.py:150-161 with the table size of BASELINE configs[2]
            return InstantNGPModel(table_sizes=[2 ** args.table_log2] * levels,
                                   grid_sizes=[2 ** (4 + i // 2) for i in range(levels)], bbox_min=BBOX_MIN,
                                   bbox_max=BBOX_MAX, precision=args.precision)

        loop = TrainLoop(ngp(6), ngp(16), init_rng=0, lr=1e-4, coarse_ts=COARSE, fine_ts=FINE, adam_eps=1e-15,
                         adam_b1=0.9, adam_b2=0.99, device=device)
    elif args.workload == "refnerf":
        from learn_nerf.ref_nerf import RefNERFModel

        loop = TrainLoop(RefNERFModel(sh_degree=4, precision=args.precision),
                         RefNERFModel(sh_degree=4, precision=args.precision), init_rng=0, lr=1e-4, coarse_ts=COARSE,
                         fine_ts=FINE, device=device)
    else:
        loop = TrainLoop(NeRFModel(precision=args.precision), NeRFModel(precision=args.precision), init_rng=0,
                         lr=1e-4, coarse_ts=COARSE, fine_ts=FINE, device=device)
    if world > 1:  # same initial parameters everywhere (init is seeded, broadcast for safety)
        dist.broadcast(loop.flat, src=0)
    step = loop.step_fn(BBOX_MIN, BBOX_MAX)
    batch = synthetic_batch(n, 1000 + rank, device)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    from learn_nerf.rng import Key

    for i in range(args.warmup):
        step(Key(i, ray_offset=rank * n), batch)
    barrier()
    _prof.enable(not args.no_kernel_timers)
    t0 = time.perf_counter()
    for i in range(args.steps):
        log = step(Key(args.warmup + i, ray_offset=rank * n), batch)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = _prof.summary()
    _prof.enable(False)
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * n * (COARSE + FINE) * args.steps / elapsed

    if rank == 0:
        m_c, m_f = n * COARSE, n * (COARSE + FINE)
        fams = {}
        for name, (cnt, ms) in prof.items():
            lvl = "coarse" if name.startswith("coarse") else ("fine" if name.startswith("fine") else None)
            flops = None
            if lvl:
                m = m_c if lvl == "coarse" else m_f
                if name.endswith("_fwd"):
                    flops = m * FLOP_FWD_PER_EVAL
                elif name.endswith("_bwd_chain"):
                    flops = m * 2 * MAC_DGRAD_PER_EVAL
                elif name.endswith("_bwd_weights"):
                    flops = m * 2 * MAC_WGRAD_PER_EVAL
            if args.workload != "nerf":
                flops = None
            fams[name] = dict(ms=round(ms, 4), calls_per_step=cnt / args.steps,
                              tflops=None if flops is None else round(flops / (ms * 1e-3) / 1e12, 1))
        roofline = None
        timed = {k: v for k, v in fams.items() if v["tflops"] is not None}
        if args.workload == "ngp":
            # gather/scatter roofline: L*8 corners*F*4 B per evaluation; scatter-add counted as 2x
            for name, v in fams.items():
                lv = 6 if name.startswith("coarse") else 16
                m = m_c if name.startswith("coarse") else m_f
                if name.endswith("hashgrid_fwd"):
                    v["GBps"] = round(m * lv * 8 * 2 * 4 / (v["ms"] * 1e-3) / 1e9, 1)
                elif name.endswith("hashgrid_bwd"):
                    v["GBps"] = round(2 * m * lv * 8 * 2 * 4 / (v["ms"] * 1e-3) / 1e9, 1)
            hg = {k: v for k, v in fams.items() if "GBps" in v}
            if hg:
                dom = max(hg, key=lambda k: hg[k]["ms"])
                roofline = dict(bound="hbm", kernel=dom, achieved=hg[dom]["GBps"], peak=8000.0, unit="GB/s",
                                frac=round(hg[dom]["GBps"] / 8000.0, 4), traffic=None)
        elif timed:
            # Algorithmic HBM bytes per 32-evaluation tile of each kernel family:
            # forward writes the 167 KiB save block, the backward chain reads 9 KiB of masks and writes
            # 156 KiB of dy, the weight-gradient kernel reads X and dy fragments (344 KiB).
            tile_bytes = {"_fwd": 167 * 1024, "_bwd_chain": (156 + 9) * 1024, "_bwd_weights": 344 * 1024}
            for name, v in timed.items():
                m = m_c if name.startswith("coarse") else m_f
                for suffix, b in tile_bytes.items():
                    if name.endswith(suffix):
                        v["GBps"] = round((m / 32) * b / (v["ms"] * 1e-3) / 1e9, 1)
            dom = max(timed, key=lambda k: timed[k]["ms"])
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
            if os.path.exists(pmc) and n == RAYS_PER_GPU:
                traffic = json.load(open(pmc)).get(dom, {}).get("hbm_bytes_per_launch")
            if dom.endswith("_bwd_weights"):
                # the dominant kernel streams its operands once from HBM (128 FLOP/B on bf16 dumps): HBM roofline
                roofline = dict(bound="hbm", kernel=dom, achieved=timed[dom]["GBps"], peak=8000.0, unit="GB/s",
                                frac=round(timed[dom]["GBps"] / 8000.0, 4), traffic=traffic,
                                mfma_tflops=timed[dom]["tflops"])
            else:
                achieved = timed[dom]["tflops"]
                roofline = dict(bound="mfma", kernel=dom, achieved=achieved, peak=PEAK_BF16_FLOPS / 1e12,
                                unit="TFLOP/s", frac=round(achieved / (PEAK_BF16_FLOPS / 1e12), 4), traffic=traffic)
        step_tflops = (value / world) * FLOP_TRAIN_PER_RAY_SAMPLE / 1e12
        out = dict(
            metric="ray-samples/s (NeRF train step: fwd + bwd + Adam, 4096 rays x 192 samples per GPU)",
            value=value, unit="ray-samples/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
            ms_per_step=ms_per_step, higher_is_better=True, scaling="weak", vs_baseline=None,
            dtype="bf16" if args.precision == "bf16" else "f32",
            data="synthetic",
            config=dict(workload={"nerf": "vanilla NeRF coarse64+fine128 train step (BASELINE.json configs[1])",
                                  "ngp": f"instant_ngp hash-grid L=6/16, T=2^{args.table_log2} train step "
                                         "(BASELINE.json configs[2])",
                                  "refnerf": "ref_nerf.py RefNERFModel sh_degree 4 train step incl. normal losses "
                                             "(BASELINE.json configs[3])"}[args.workload],
                        rays_per_gpu=n, coarse_samples=COARSE, fine_samples=FINE, global_batch_rays=world * n,
                        parallelism=f"dp{world}", precision=args.precision + " MFMA, fp32 accumulate/master weights"
                        if args.precision == "bf16" else "fp32 (f32 MFMA)"),
            roofline=roofline,
            step_mfma=dict(achieved=round(step_tflops, 1), peak=PEAK_BF16_FLOPS / 1e12, unit="TFLOP/s",
                           frac=round(step_tflops / (PEAK_BF16_FLOPS / 1e12), 4),
                           note="whole step per GPU: ray-samples/s x 4,641,792 algorithmic FLOP (SURVEY 8d)"),
            kernels=fams,
            losses={k: round(float(v), 5) for k, v in log.items()},
        )
        if args.workload != "nerf":
            del out["step_mfma"]  # the FLOP model is the vanilla NeRFModel's; the hash-grid step is gather/scatter bound
        if args.workload == "nerf" and args.precision == "bf16":
            # the same fused MLP kernel without the activation dumps (what render_nerf.py runs): compute-bound
            from learn_nerf import ops as _ops

            _, _, _, ts_c = _ops.ray_aabb_stratified(batch, BBOX_MIN, BBOX_MAX, COARSE + FINE, seed=1)
            c_flat = loop._slices(loop.flat)[1]
            for _ in range(3):
                loop.fine.forward_rays(c_flat, batch, ts_c, save=False)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                loop.fine.forward_rays(c_flat, batch, ts_c, save=False)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            tf = m_f * FLOP_FWD_PER_EVAL / (ms * 1e-3) / 1e12
            out["inference_mlp"] = dict(ms=round(ms, 4), tflops=round(tf, 1), frac_of_mfma_peak=round(tf / 2500.0, 4),
                                        note="fine-pass NeRFModel forward, no activation save (render path)")
        if world == 1 and not args.no_cpu_baseline and args.workload == "nerf":
            out["cpu_baseline"] = cpu_baseline()
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
