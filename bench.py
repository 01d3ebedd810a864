#!/usr/bin/env python
"""
bench.py — throughput of the NeRF train step (learn_nerf/train.py:78-112 restated as HIP kernels)
on N GPUs of one node.

  python bench.py --gpus N --steps K --warmup W
  N > 1: either the driver's launch (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py
  --gpus N ...), or plain `python bench.py --gpus N`, which starts that launcher itself as a child process before
  anything touches the GPU.  The run fails if the RCCL process group does not have exactly N ranks.

Workload (BASELINE.json configs[1], the metric's own config): vanilla NeRF, 4096 rays per GPU,
64 coarse + 128 fine samples per ray (192 ray-samples per ray, 256 MLP evaluations per ray),
bf16 MFMA with fp32 accumulate / master weights, synthetic rays aimed at bbox [-1,1]^3, random
targets, Flax-default initial weights.  One step = forward + backward + [RCCL all-reduce] + Adam.
Rays shard data-parallel: every rank draws its own 4096 rays (weak scaling), gradients are
all-reduced (sum) over RCCL and averaged inside the fused Adam kernel.

Rank 0 prints ONE JSON line (see the task contract) with these extra objects:
  roofline     — SURVEY.md 8(d): configs[1] is MFMA-bound, so the dominant kernel's algorithmic FLOP per launch / its
                 mean HIP-event time in the timed region vs the dense bf16 MFMA peak of MI355X (2.5 PFLOP/s,
                 /opt/skills/guides/MI355X_MICROARCH.md).  The dominant launch is nerf_bwd_ls_kernel, the persistent
                 layer-stationary pipeline of BOTH models' backward (section bwd_ls_pipeline: Dense_8 .. Dense_1, input
                 and weight gradients = 2,097,152 algorithmic FLOP per model evaluation); `traffic` = HBM bytes per
                 launch from the rocprofv3 PMC summary named in `traffic_source` (a separate profiling run, not this one)
  roofline_hbm — the same kernel against the HBM roof (its design bytes per launch / time / 8 TB/s)
  cpu_baseline — the oracle's torch-CPU fp32 restatement of the same step on a bounded sample
                 (rank 0, N = 1 only); a reported baseline, not the optimisation target.
  other_workloads (N = 1) — short legs of BASELINE configs[2] (instant_ngp, HBM roofline) and configs[3] (ref_nerf)
  inference_mlp / inference_mlp_split — the render-path forward: plain bf16 and the split-precision kernel that
                 NeRFRenderer runs (rendered RGB within 1e-3 of fp32)
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
# layer-stationary backward: the persistent pipeline launch holds Dense_8 .. Dense_1 (input gradient + weight gradient of
# the 256 x 256 kernels), the head launches Dense_11 / 10 / 9 -> dz, the finish launches the five small weight-gradient
# problems (z x dy10m, x_emb x dy0, x_emb x dy5, d_emb x dy10m, h10 x dy11)
MAC_LS_PIPELINE_PER_EVAL = 2 * 8 * 65_536
MAC_LS_HEAD_PER_EVAL = 128 * 3 + 256 * 128 + 256
MAC_LS_FINISH_PER_EVAL = MAC_DGRAD_PER_EVAL + MAC_WGRAD_PER_EVAL - MAC_LS_PIPELINE_PER_EVAL - MAC_LS_HEAD_PER_EVAL


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


def self_launch(n_gpus: int) -> int:
    """`python bench.py --gpus N` without a launcher: start one rank per GPU through torch.distributed.run as a
    CHILD process (nothing in this process has touched the GPU yet) and pass its exit code on."""
    import socket
    import subprocess

    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def comm_abi_probe(dist, rank, world, local_rank, device):
    """Exercise the C-ABI RCCL entries (lnrf_comm_*) across the real ranks: the same vector reduced through them
    must equal the closed-form sum.  Called by every rank AFTER the timed region and after every other measurement,
    so that it can never disturb them: the unique id is agreed on collectively (a failure on rank 0 reaches every
    rank as None) and the communicator part runs in a daemon thread that the main thread abandons after 60 s — the
    caller then prints its line and leaves with os._exit, without another device synchronisation."""
    import threading

    from learn_nerf import parallel as _par

    uid = None
    if rank == 0:
        try:
            uid = _par.AbiComm.new_unique_id()
        except Exception:
            uid = None
    box = [uid]
    dist.broadcast_object_list(box, src=0)
    if box[0] is None:
        return "unavailable: lnrf_comm_get_unique_id failed on rank 0", True
    result = {}

    def _probe():
        try:
            torch.cuda.set_device(local_rank)
            comm = _par.AbiComm(box[0], rank, world)
            probe = torch.arange(4096, dtype=torch.float32, device=device) * (rank + 1)
            comm.all_reduce_sum_(probe)
            torch.cuda.synchronize()
            expect = torch.arange(4096, dtype=torch.float32, device=device) * (world * (world + 1) / 2)
            result["msg"] = "ok" if torch.equal(probe, expect) else "MISMATCH vs the expected sum"
            comm.destroy()
        except Exception as exc:
            result["msg"] = f"{type(exc).__name__}: {exc}"

    th = threading.Thread(target=_probe, daemon=True)
    th.start()
    th.join(timeout=60.0)
    if "msg" in result:
        return result["msg"], True
    return "timeout after 60 s (abandoned)", False


def build_loop(workload, precision, table_log2, device, backward="ls"):
    from learn_nerf.model import NeRFModel
    from learn_nerf.train import TrainLoop

    if workload == "ngp":
        from learn_nerf.instant_ngp import InstantNGPModel

        def ngp(levels):  # scripts/train_nerf.py:150-161 with the table size of BASELINE configs[2]
            return InstantNGPModel(table_sizes=[2 ** table_log2] * levels,
                                   grid_sizes=[2 ** (4 + i // 2) for i in range(levels)], bbox_min=BBOX_MIN,
                                   bbox_max=BBOX_MAX, precision=precision)

        return TrainLoop(ngp(6), ngp(16), init_rng=0, lr=1e-4, coarse_ts=COARSE, fine_ts=FINE, adam_eps=1e-15,
                         adam_b1=0.9, adam_b2=0.99, device=device)
    if workload == "refnerf":
        from learn_nerf.ref_nerf import RefNERFModel

        return TrainLoop(RefNERFModel(sh_degree=4, precision=precision),
                         RefNERFModel(sh_degree=4, precision=precision), init_rng=0, lr=1e-4, coarse_ts=COARSE,
                         fine_ts=FINE, device=device)
    return TrainLoop(NeRFModel(precision=precision, backward_kernel=backward),
                     NeRFModel(precision=precision, backward_kernel=backward), init_rng=0,
                     lr=1e-4, coarse_ts=COARSE, fine_ts=FINE, device=device)


def leg_traffic(workload, kernel_names, launch_in_step):
    """HBM bytes per launch of the dominant kernel family of a secondary workload, from the committed rocprofv3 PMC summary
    (profiles/r03_<workload>_pmc_summary.json; launch_in_step: 0 = coarse pass, 1 = fine pass)."""
    path = os.path.join(ROOT, "profiles", f"r03_{workload}_pmc_summary.json")
    if not os.path.exists(path):
        return None, None
    d = json.load(open(path))
    total = 0.0
    for k in kernel_names:
        v = d.get(k)
        if not v:
            continue
        per = v.get("hbm_bytes_by_launch_in_step")
        if per and launch_in_step < len(per) and len(per) == 2:
            total += per[launch_in_step]
        elif per and len(per) > 2:  # several launches of the kernel per pass: the pass's share of the step
            half = len(per) // 2
            total += sum(per[launch_in_step * half:(launch_in_step + 1) * half])
        else:
            return None, None
    src = (f"profiles/r03_{workload}_pmc_summary.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of "
           f"`bench.py --workload {workload}` (gfx950 FETCH x2 correction), kernels {kernel_names}, NOT measured in this run")
    return (total if total > 0 else None), src


def ngp_roofline(fams, m_c, m_f):
    """gather/scatter roofline: L * 8 corners * F * 4 B per evaluation; the scatter-add counts 2x"""
    for name, v in fams.items():
        lv = 6 if name.startswith("coarse") else 16
        m = m_c if name.startswith("coarse") else m_f
        if name.endswith("hashgrid_fwd"):
            v["GBps"] = round(m * lv * 8 * 2 * 4 / (v["ms"] * 1e-3) / 1e9, 1)
        elif name.endswith("hashgrid_bwd"):
            v["GBps"] = round(2 * m * lv * 8 * 2 * 4 / (v["ms"] * 1e-3) / 1e9, 1)
    hg = {k: v for k, v in fams.items() if "GBps" in v}
    if not hg:
        return None
    dom = max(hg, key=lambda k: hg[k]["ms"])
    names = (["lnrf::hashgrid_bin_kernel", "lnrf::hashgrid_reduce_kernel"] if dom.endswith("hashgrid_bwd")
             else ["lnrf::hashgrid_fwd_xcd_kernel", "lnrf::hashgrid_fwd_kernel"])
    traffic, src = leg_traffic("ngp", names, 0 if dom.startswith("coarse") else 1)
    return dict(bound="hbm", kernel=dom, achieved=hg[dom]["GBps"], peak=8000.0, unit="GB/s",
                frac=round(hg[dom]["GBps"] / 8000.0, 4), traffic=traffic, traffic_source=src)


# RefNERFModel, MACs per evaluation (ref_nerf.py:92-107 widths): trunk Dense_0..8 = 60*256 + 4*256^2 + 316*256 + 3*256^2
REF_TRUNK_MAC = 60 * 256 + 4 * 65536 + 316 * 256 + 3 * 65536   # 555,008: forward, = normal pass, = weight gradients
REF_TRUNK_DGRAD_MAC = 8 * 65536                                  # 524,288: first-order input-gradient chain
REF_TANGENT_MAC = REF_TRUNK_MAC - 65536                          # 489,472: second-order tangent chain (Dense_0..7)
REF_DIR_MAC = 273 * 128 + 128 * 3                                # 35,328: directional block, x3 (fwd, dgrad, wgrad)
REF_STEP_MAC = (2 * REF_TRUNK_MAC + (REF_TRUNK_DGRAD_MAC + REF_TRUNK_MAC) + (REF_TANGENT_MAC + REF_TRUNK_MAC)
                + 3 * REF_DIR_MAC)                               # 3,339,776 MAC = 6.68 MFLOP per evaluation


def refnerf_roofline(fams, n, ms_step):
    """configs[3] is MFMA-bound like configs[1]: algorithmic FLOP of the dominant fused kernel family / its HIP-event
    time / bf16 peak, and the whole step's algorithmic FLOP rate."""
    flop = {"_spatial_fwd": 2 * REF_TRUNK_MAC, "_normal_pass": 2 * REF_TRUNK_MAC,
            "_spatial_bwd": 2 * (REF_TRUNK_DGRAD_MAC + REF_TRUNK_MAC), "_normal_bwd": 2 * (REF_TANGENT_MAC + REF_TRUNK_MAC)}
    best = None
    for name, v in fams.items():
        m = n * (COARSE if name.startswith("coarse") else COARSE + FINE)
        for suffix, per_eval in flop.items():
            if name.endswith(suffix):
                v["tflops"] = round(m * per_eval / (v["ms"] * 1e-3) / 1e12, 1)
                if best is None or v["ms"] > fams[best]["ms"]:
                    best = name
    roofline = None
    if best is not None:
        kn = {"_spatial_fwd": ["lnrf::refnerf_trunk_fwd_kernel"], "_normal_pass": ["lnrf::refnerf_normal_kernel"],
              "_spatial_bwd": ["lnrf::refnerf_dy8_kernel", "lnrf::nerf_bwd_ls_kernel"],
              "_normal_bwd": ["lnrf::refnerf_tangent_kernel"]}
        names = next(v for k, v in kn.items() if best.endswith(k))
        traffic, src = leg_traffic("refnerf", names, 0 if best.startswith("coarse") else 1)
        roofline = dict(bound="mfma", kernel=best, achieved=fams[best]["tflops"], peak=PEAK_BF16_FLOPS / 1e12,
                        unit="TFLOP/s", frac=round(fams[best]["tflops"] / (PEAK_BF16_FLOPS / 1e12), 4), traffic=traffic,
                        traffic_source=src,
                        ms_per_launch=fams[best]["ms"],
                        note="dominant fused-trunk kernel family (chain + weight-gradient launches)")
    tf = n * (COARSE + COARSE + FINE) * 2 * REF_STEP_MAC / (ms_step * 1e-3) / 1e12
    step = dict(achieved=round(tf, 1), peak=PEAK_BF16_FLOPS / 1e12, unit="TFLOP/s", frac=round(tf / (PEAK_BF16_FLOPS / 1e12), 4),
                note="whole step: 6,679,552 algorithmic FLOP per model evaluation incl. the second-order normal term")
    return roofline, step


def short_leg(workload, n, device, table_log2, steps=50, warmup=10):
    """Another BASELINE config on this GPU (N = 1 only), so that the driver's record covers it.  Every step is timed on
    its own (host clock around a synchronised step) so that a one-off stall shows up as `max` / `max_at` instead of
    hiding in a mean: the headline `ms_per_step` is the back-to-back loop, `step_ms` the per-step distribution."""
    from learn_nerf import _prof
    from learn_nerf.rng import Key

    loop = build_loop(workload, "bf16", table_log2, device)
    step = loop.step_fn(BBOX_MIN, BBOX_MAX)
    batch = synthetic_batch(n, 1000, device)
    for i in range(warmup):
        step(Key(i), batch)
    torch.cuda.synchronize()
    per_step = []
    for i in range(steps):  # pass 1: one synchronised step at a time
        t1 = time.perf_counter()
        step(Key(warmup + i), batch)
        torch.cuda.synchronize()
        per_step.append(1e3 * (time.perf_counter() - t1))
    _prof.enable(True)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):  # pass 2: back to back (what the headline workload measures)
        step(Key(warmup + steps + i), batch)
        marks[i + 1].record()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    gaps = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    prof = _prof.summary()
    _prof.enable(False)
    srt = sorted(per_step)
    step_ms = dict(median=round(srt[len(srt) // 2], 3), min=round(srt[0], 3), max=round(srt[-1], 3),
                   max_at=int(per_step.index(srt[-1])), note="synchronised single steps (host clock), before the timed loop")
    gs = sorted(gaps)
    step_ms["back_to_back"] = dict(median=round(gs[len(gs) // 2], 3), min=round(gs[0], 3), max=round(gs[-1], 3),
                                   max_at=int(gaps.index(gs[-1])), first5=[round(g, 3) for g in gaps[:5]],
                                   last5=[round(g, 3) for g in gaps[-5:]],
                                   note="HIP-event time between consecutive steps of the timed loop")
    fams = {k: dict(ms=round(v[1], 4), calls_per_step=v[0] / steps) for k, v in prof.items()}
    out = dict(ms_per_step=round(ms, 3), value=round(n * (COARSE + FINE) / (ms * 1e-3), 1), unit="ray-samples/s",
               steps=steps, warmup=warmup, step_ms=step_ms)
    if workload == "ngp":
        # whole step against the HBM roof: 663,552 algorithmic bytes per ray
        out["step_hbm"] = dict(achieved=round(n * 663_552 / (ms * 1e-3) / 1e9, 1), peak=8000.0, unit="GB/s",
                               frac=round(n * 663_552 / (ms * 1e-3) / 8e12, 4))
        out["roofline"] = ngp_roofline(fams, n * COARSE, n * (COARSE + FINE))
        out["config"] = f"instant_ngp hash-grid L=6/16, T=2^{table_log2} (BASELINE configs[2])"
    else:
        out["roofline"], out["step_mfma"] = refnerf_roofline(fams, n, ms)
        out["config"] = "ref_nerf.py RefNERFModel sh_degree 4 incl. normal losses (BASELINE configs[3])"
    out["kernels"] = fams
    del loop, step
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rays", type=int, default=RAYS_PER_GPU, help="rays per GPU")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--workload", default="nerf", choices=["nerf", "ngp", "refnerf"],
                    help="nerf = BASELINE configs[1] (the metric's config, default); ngp = configs[2] hash-grid path; "
                         "refnerf = configs[3] RefNERFModel (fused spatial block)")
    ap.add_argument("--table_log2", type=int, default=19, help="ngp: log2 of the hash table size (configs[2]: 19)")
    ap.add_argument("--backward", default="ls", choices=["ls", "split"],
                    help="nerf: NeRFModel backward = layer-stationary pipeline (default) or the older chain + weight-gradient "
                         "launches (A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the short ngp / refnerf legs")
    ap.add_argument("--comm-abi-check", action="store_true",
                    help="N > 1: after every measurement, also reduce a probe vector through the C-ABI RCCL entries "
                         "(lnrf_comm_*) across the ranks and report it as comm_abi_check (opt-in: a second RCCL "
                         "communicator next to torch.distributed's has only been exercised with one rank so far)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))  # before any GPU call in this process

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
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE {world} ranks")
    if dist is not None and dist.get_world_size() != args.gpus:
        raise SystemExit(f"bench.py: the RCCL process group has {dist.get_world_size()} ranks, expected {args.gpus}")
    if torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPUs visible")

    from learn_nerf import _prof

    n = args.rays
    loop = build_loop(args.workload, args.precision, args.table_log2, device, args.backward)
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
            # layer-stationary backward of BOTH models: head launches, ONE persistent pipeline launch, finish launches
            ls_mac = {"bwd_ls": MAC_DGRAD_PER_EVAL + MAC_WGRAD_PER_EVAL, "bwd_ls_head": MAC_LS_HEAD_PER_EVAL,
                      "bwd_ls_pipeline": MAC_LS_PIPELINE_PER_EVAL, "bwd_ls_finish": MAC_LS_FINISH_PER_EVAL}
            if name in ls_mac:
                flops = (m_c + m_f) * 2 * ls_mac[name]
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
        roofline, roofline_hbm = None, None
        timed = {k: v for k, v in fams.items() if v["tflops"] is not None}
        if args.workload == "ngp":
            roofline = ngp_roofline(fams, m_c, m_f)
        elif args.workload == "refnerf":
            roofline, ref_step = refnerf_roofline(fams, n, ms_per_step)
        elif timed:
            # Design HBM bytes per 32-evaluation tile of each kernel family:
            # forward writes the 167 KiB save block, the backward chain reads 9 KiB of masks and writes
            # 156 KiB of dy, the weight-gradient kernel reads X and dy fragments (344 KiB).
            # layer-stationary backward: head (9 + 28 KiB), pipeline (reads 128 KiB of X + 16 of dy8, writes 128 of dy; the
            # dy reads behind the producer are served on chip), small problems (88 KiB)
            tile_bytes = {"_fwd": 167 * 1024, "_bwd_chain": (156 + 9) * 1024, "_bwd_weights": 344 * 1024,
                          "bwd_ls": (30 + 272 + 88) * 1024, "bwd_ls_head": 30 * 1024, "bwd_ls_pipeline": 272 * 1024,
                          "bwd_ls_finish": 88 * 1024}
            for name, v in timed.items():
                m = m_c if name.startswith("coarse") else (m_c + m_f if name.startswith("bwd_ls") else m_f)
                for suffix, b in tile_bytes.items():
                    if name.endswith(suffix):
                        v["GBps"] = round((m / 32) * b / (v["ms"] * 1e-3) / 1e9, 1)
            dom = max(timed, key=lambda k: timed[k]["ms"])
            traffic, traffic_source = None, None
            for cand in ("r03_pmc_summary.json", "r02_pmc_summary.json", "r01_pmc_summary.json"):
                pmc = os.path.join(ROOT, "profiles", cand)
                if os.path.exists(pmc) and n == RAYS_PER_GPU:
                    traffic = json.load(open(pmc)).get(dom, {}).get("hbm_bytes_per_launch")
                    if traffic is None and dom == "bwd_ls_pipeline":
                        traffic = json.load(open(pmc)).get("lnrf::nerf_bwd_ls_kernel", {}).get("hbm_bytes_per_launch")
                    if traffic is not None:
                        traffic_source = (f"profiles/{cand}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an "
                                          "earlier run of this command (gfx950 FETCH x2 correction), NOT measured in "
                                          "this run")
                        break
            # configs[1] is MFMA-bound -> algorithmic FLOP of the dominant kernel / its time / bf16 peak
            achieved = timed[dom]["tflops"]
            roofline = dict(bound="mfma", kernel=dom, achieved=achieved, peak=PEAK_BF16_FLOPS / 1e12,
                            unit="TFLOP/s", frac=round(achieved / (PEAK_BF16_FLOPS / 1e12), 4), traffic=traffic,
                            traffic_source=traffic_source, ms_per_launch=timed[dom]["ms"])
            if "GBps" in timed[dom]:
                roofline_hbm = dict(bound="hbm", kernel=dom, achieved=timed[dom]["GBps"], peak=8000.0, unit="GB/s",
                                    frac=round(timed[dom]["GBps"] / 8000.0, 4),
                                    note="design bytes per launch (DESIGN.md section 4) / time: the dump traffic "
                                         "this kernel streams, not an algorithmic minimum")
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
            roofline_hbm=roofline_hbm,
            step_mfma=dict(achieved=round(step_tflops, 1), peak=PEAK_BF16_FLOPS / 1e12, unit="TFLOP/s",
                           frac=round(step_tflops / (PEAK_BF16_FLOPS / 1e12), 4),
                           note="whole step per GPU: ray-samples/s x 4,641,792 algorithmic FLOP (SURVEY 8d)"),
            kernels=fams,
            losses={k: round(float(v), 5) for k, v in log.items()},
        )
        if args.workload == "refnerf":
            out["step_mfma"] = ref_step
        elif args.workload != "nerf":
            del out["step_mfma"]  # the FLOP model is the vanilla NeRFModel's; the hash-grid step is gather/scatter bound
        if args.workload == "nerf" and args.precision == "bf16":
            # the render path: the fused forward without activation dumps, plain bf16 and split precision
            from learn_nerf import ops as _ops

            _, _, _, ts_c = _ops.ray_aabb_stratified(batch, BBOX_MIN, BBOX_MAX, COARSE + FINE, seed=1)
            c_flat = loop._slices(loop.flat)[1]
            for key, rp, note in (("inference_mlp", "bf16", "fine-pass NeRFModel forward, no activation save, plain "
                                   "bf16 operands (render_precision='bf16')"),
                                  ("inference_mlp_split", "bf16x3", "the kernel NeRFRenderer / render_nerf.py run: bf16 "
                                   "hi+lo operands, 3 MFMAs per product (rendered RGB within 1e-3 of fp32); tflops = "
                                   "algorithmic fp32-equivalent FLOP, MFMA work is 3x that")):
                loop.fine.render_precision = rp
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
                out[key] = dict(ms=round(ms, 4), tflops=round(tf, 1), frac_of_mfma_peak=round(tf / 2500.0, 4), note=note)
            loop.fine.render_precision = "bf16x3"
        if world == 1 and args.workload == "nerf" and not args.no_other_workloads:
            del step
            loop = None
            torch.cuda.empty_cache()
            out["other_workloads"] = {}
            for wl in ("ngp", "refnerf"):
                try:
                    out["other_workloads"][wl] = short_leg(wl, n, device, args.table_log2)
                except Exception as exc:  # a broken secondary leg must not hide the headline measurement
                    out["other_workloads"][wl] = dict(error=f"{type(exc).__name__}: {exc}")
        if world == 1 and not args.no_cpu_baseline and args.workload == "nerf":
            out["cpu_baseline"] = cpu_baseline()
    clean = True
    if world > 1 and args.comm_abi_check:  # last: nothing measured above can be disturbed by it
        msg, clean = comm_abi_probe(dist, rank, world, local_rank, device)
        if rank == 0:
            out["comm_abi_check"] = msg
    if rank == 0:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if not clean:  # an abandoned probe may have left a spinning collective behind: no further device calls
        sys.stderr.flush()
        os._exit(0)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
