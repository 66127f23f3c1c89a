#!/usr/bin/env python
"""bench.py -- guided trajectories/sec of the MI355X sampler on BASELINE.json's headline config.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is ONE planning round of the hot path: (all-gather of the robots' best paths ->) device-side soft-constraint
table -> one guided DDPM sampling call (T=100 denoise steps + 1 no-noise step, 20 guide iterations on the 51 guided
steps) for every local robot's B=64 samples -> best-path selection for the next round.  N=1 is BASELINE.json's headline
workload: 32 robots on the Empty map (circle r=0.8) = 2048 trajectories, each robot soft-constrained by the other 31
(31 x 63 = 1953 points).  N>1, `--scaling strong` (default): the SAME 32-robot instance, the metric's own workload,
sharded 32/N robots per GPU (512 trajectories per GPU at N=4, 256 at N=8).  `--scaling weak`: 32 robots PER GPU of one
32N-robot instance (the pairwise term grows with N).  Either way ONE RCCL all-gather of [robots/GPU,64,2] fp32 per rank
per round.  UNet weights are synthetic random-init (numpy PCG64), Gaussian noise is drawn in-kernel (Philox keyed by the
global trajectory index, so every rank's rows equal the unsharded run's), inputs are resident in HBM before the timed
region.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time
from math import ceil

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402

H, D = 64, 4
PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2516.6     # dense bf16 MFMA peak (1024 FLOP/clk/SIMD): 16x the fp32 MFMA rate
DOMINANT_KERNEL = "UNET"           # unet_kernel: the whole TemporalUnet forward in one launch (all 25 convs + GN/Mish)
HEADLINE_ROBOTS = 32               # BASELINE.json: 32-robot Empty map


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N>1: strong = the metric's 32-robot instance sharded 32/N robots per GPU; weak = 32 robots per "
                         "GPU of a 32N-robot instance")
    ap.add_argument("--robots-per-gpu", type=int, default=0, help="override (0 = 32/N for strong, 32 for weak)")
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--diffusion-steps", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    return ap.parse_args()


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(T, B, n_robots, budget_s):
    """The oracle's reference-SHAPED path (dense (n,B,H,2) CostConstraint broadcast + one autograd pass per cost term,
    torch-CPU UNet) for ONE robot of the headline instance, timed on this host's cores on a bounded sample and
    extrapolated to the full 101-step call.  Robots are planned sequentially by the reference, so trajectories/s of
    one robot's call is the whole-round rate.  The thread count is swept upwards from 8 (an oversubscribed pool is slower
    for these small tensors) and the best setting is reported."""
    import cases_for_bench as cb
    from oracle import mmd_oracle as O
    sd, tb, gp, grp, hc = cb.oracle_headline_robot(T, n_robots)
    x = torch.from_numpy(cb.synth.synth_noise(91, (B, H, D))) * 0.5
    x = O.apply_hard_conditioning(x, hc)
    noise = torch.from_numpy(cb.synth.synth_noise(92, (B, H, D)))
    guide = lambda y: O.guide_grad_dense_autograd(y, gp, [grp])     # noqa: E731
    tsg = ceil(0.5 * T)

    def timed(i, g):
        t0 = time.perf_counter()
        with torch.no_grad():
            O.ddpm_sample_step(sd, tb, x.clone(), hc, i, guide=g, n_guide_steps=20, t_start_guide=tsg, noise=noise,
                               noise_std_extra=0.5)
        return time.perf_counter() - t0

    n_cpus = os.cpu_count() or 1
    # oversubscribed pools are catastrophically slow for these small tensors (256 threads: 100x slower than 8), so the
    # sweep climbs from 8 and stops as soon as a setting is clearly worse than the best so far
    sweep = [t for t in (8, 16, 32, 64, 128) if t <= n_cpus] or [n_cpus]
    n_guided, n_unguided = tsg + 1, T - tsg                          # i = tsg-1 ... -1 guided; the rest unguided
    results, t_start = [], time.perf_counter()
    for nt in sweep:
        if results and time.perf_counter() - t_start > budget_s:
            break
        torch.set_num_threads(nt)
        timed(T - 1, None)                                           # warm-up (thread pool, allocator)
        t_u = min(timed(T - 1 - k, None) for k in range(2))
        t_g = [timed(tsg - 1 - k, guide) for k in range(2)]
        est = n_guided * float(np.mean(t_g)) + n_unguided * t_u
        results.append({"threads": nt, "guided_step_s": float(np.mean(t_g)), "unguided_step_s": t_u,
                        "est_seconds_per_robot_call": est, "trajectories_per_s": B / est})
        if B / est < 0.7 * max(r["trajectories_per_s"] for r in results):
            break
    best = max(results, key=lambda r: r["trajectories_per_s"])
    return {"value": best["trajectories_per_s"], "unit": "trajectories/s", "cores": best["threads"], "kind": "port",
            "cpu_model": cpu_model_name(), "logical_cpus": n_cpus,
            "sample": f"1 of {n_robots} robots (B={B}, {grp.q.shape[0]} soft-constraint points): per thread count 2 guided "
                      f"+ 2 unguided DDPM steps timed and extrapolated to the {n_guided}+{n_unguided}-step call (robots "
                      f"are sequential in the reference); best of the thread sweep reported",
            "est_seconds_per_robot_call": best["est_seconds_per_robot_call"], "thread_sweep": results}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (there is no CPU fallback for the product path)"
    # MMD_BENCH_REHEARSAL=1: rehearse the N>1 path on ONE GPU (all ranks on cuda:0, gloo) -- not a measurement
    rehearsal = os.environ.get("MMD_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from mmd_amd import _lib, synth
    from mmd_amd.diffusion_model import GaussianDiffusionModel
    from mmd_amd.multi_robot import MultiRobotSampler
    from mmd_amd.temporal_unet import TemporalUnet

    T, B = args.diffusion_steps, args.samples
    if args.robots_per_gpu:
        RPG = args.robots_per_gpu
    elif args.scaling == "strong":
        if HEADLINE_ROBOTS % world:
            raise SystemExit(f"--scaling strong needs {HEADLINE_ROBOTS} % gpus == 0")
        RPG = HEADLINE_ROBOTS // world
    else:
        RPG = HEADLINE_ROBOTS
    n_robots = RPG * world
    unet = TemporalUnet(state_dim=4, n_support_points=H, unet_input_dim=32, dim_mults=(1, 2, 4))
    unet.load_state_dict(synth.synth_unet_state_dict(0))
    model = GaussianDiffusionModel(model=unet, variance_schedule="exponential", n_diffusion_steps=T, predict_epsilon=True)
    starts, goals = synth.start_goal_circle(n_robots, 0.8)
    sampler = MultiRobotSampler(model, starts, goals, env_id="EnvEmpty2D", n_samples=B, rank=rank, world_size=world,
                                device=dev)
    # round 0 input: straight-line paths stand in for "previous best paths" 
    paths_local = torch.from_numpy(synth.straight_line_paths(starts, goals, H)[sampler.robot0:sampler.robot0 + RPG]).to(dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    lib = _lib.load()
    import ctypes as C
    n_traj_local = RPG * B
    flops = lib.mmd_unet_flops_per_trajectory() * n_traj_local              # algorithmic (direct-conv) FLOPs per launch
    mfma_flops = lib.mmd_unet_mfma_flops_per_trajectory() * n_traj_local    # fp32 GEMM FLOPs the matrix pipe runs
    bf_flops = lib.mmd_unet_f16x2_flops_per_trajectory() * n_traj_local    # ... of which as bf16x3 on the bf16 pipe

    for w in range(args.warmup):
        _, paths_local = sampler.plan_round(paths_local, seed=1000 + w)
    # roofline of the dominant kernel: its launches INSIDE the timed region are bracketed by HIP event pairs on the stream
    # it is launched on (mmd_amd_debug.h profiler attached to the sampler).  Every 13th of the 101 launches per step is
    # bracketed: ~8 event pairs per step, < 0.5 % of the timed region.
    prof = C.c_void_p()
    _lib.check(lib.mmd_profiler_create(C.byref(prof), (T + 1) * args.steps, 13))
    model.profiler = prof
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        trajs, paths_local = sampler.plan_round(paths_local, seed=k)
    barrier()
    dt = time.perf_counter() - t0
    model.profiler = None
    dom_ms_c, dom_n = C.c_double(), C.c_int()
    _lib.check(lib.mmd_profiler_read(prof, C.byref(dom_ms_c), C.byref(dom_n)))
    _lib.check(lib.mmd_profiler_destroy(prof))
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(trajs).all()
    value = args.steps * n_traj_local * world / dt
    dom_ms = dom_ms_c.value
    # outside the timed region: the same kernel as ONE launch of all local trajectories, back to back on one stream -- the
    # per-launch figure that does not depend on how the sampler chunks its batch (kernel quality from round to round)
    xs = torch.randn(n_traj_local, H, 4, device=dev)
    for _ in range(3):
        unet(xs, 50)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        unet(xs, 50)
    e1.record()
    torch.cuda.synchronize()
    solo_ms = e0.elapsed_time(e1) / 20
    # mmd_p_sample_loop splits the robots into `chunks` concurrent launch chains (HIP streams, default 2): at any time
    # `chunks` unet_kernel launches of n_traj_local / chunks trajectories each share the GPU.  launch_ms is the mean
    # duration of ONE such launch (what rocprofv3 --stats reports for unet_kernel); the rate the GPU sustains is that of
    # all `chunks` launches in flight.
    chunks = int(os.environ.get("MMD_AMD_STREAMS", "0") or 0) or (2 if n_traj_local >= 2048 else 1)
    chunks = max(1, min(chunks, 4, RPG))
    launch_flops, launch_mfma, launch_bf = flops / chunks, mfma_flops / chunks, bf_flops / chunks
    issued_tf = mfma_flops / (dom_ms * 1e-3) / 1e12
    alg_tf = flops / (dom_ms * 1e-3) / 1e12

    # HBM traffic of the dominant kernel comes from rocprofv3 PMC passes (separate runs; tools/gpu_round.sh), committed
    # as profiles/pmc_latest.json: traffic = (2 * FETCH_SIZE + WRITE_SIZE) KiB (gfx950 FETCH_SIZE half-count correction)
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path) and n_traj_local // chunks == 1024:
        with open(pmc_path) as f:
            pmc = json.load(f).get(DOMINANT_KERNEL)
        if pmc:
            traffic = (2.0 * pmc["FETCH_SIZE_KiB"] + pmc["WRITE_SIZE_KiB"]) * 1024.0
    # `achieved` / `frac`: the fp32 GEMM FLOPs the kernel actually runs on the matrix pipe per launch (Winograd F(4,5) for
    # the k=5 convs, i.e. 0.45x the multiplies of the direct form) / launch time, against the fp32 MFMA peak -- the
    # roofline of fp32 arithmetic on this chip.  55 % of those FLOPs (downs.2 + mid, ups.0 conv A) run as bf16x3 on the bf16 pipe (an
    # exact three-way split of both operands, six bf16 MFMAs per fp32 chunk: fp32-accurate and 2.7x the fp32 MFMA rate),
    # so the pipe's BUSY fraction is lower than `frac`: `pipe_busy_model` prices every MFMA at its issue cycles.  The
    # ALGORITHMIC (direct-convolution) rate is reported separately and may exceed the peak.
    busy_s = ((mfma_flops - bf_flops) / (PEAK_FP32_MFMA_TFLOPS * 1e12) + 6.0 * bf_flops / (PEAK_BF16_MFMA_TFLOPS * 1e12))
    roofline = {"bound": "mfma", "kernel": "unet_kernel: whole TemporalUnet forward for 4 trajectories per workgroup (12 ResidualTemporalBlocks, 2 down / 2 up convs, final block; the 25 k=5 convs as Winograd F(4,5): fp32 MFMA GEMMs, the seven 128->128 convs and ups.0 conv A as bf16x3 on the bf16 MFMA; GroupNorm + Mish + time bias + residuals fused, activations in LDS/registers)",
                "achieved": issued_tf, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": issued_tf / PEAK_FP32_MFMA_TFLOPS,
                "traffic": traffic, "launch_ms": dom_ms, "launches_timed": dom_n.value,
                "concurrent_launches": chunks, "trajectories_per_launch": n_traj_local // chunks,
                "flops_per_launch": launch_mfma, "flops_per_launch_as_bf16x3": launch_bf,
                "achieved_single_launch": launch_mfma / (dom_ms * 1e-3) / 1e12,
                "pipe_busy_model": busy_s / (dom_ms * 1e-3),
                "whole_batch_single_launch": {"trajectories": n_traj_local, "launch_ms": solo_ms,
                                              "achieved": mfma_flops / (solo_ms * 1e-3) / 1e12,
                                              "frac": mfma_flops / (solo_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                                              "note": "the same kernel as one launch of all local trajectories, 20 back to back on one stream, outside the timed region"},
                "note": f"flops = fp32 GEMM FLOPs run on the matrix pipe (Winograd-domain; {100 * bf_flops / mfma_flops:.0f} % of them as bf16x3: 6 bf16 MFMA FLOPs per fp32 FLOP). "
                        "`concurrent_launches` launches of `trajectories_per_launch` trajectories share the GPU at any time (the sampler's stream chunks); "
                        "launch_ms = mean duration of one of them (HIP events on its stream; = rocprofv3's average for unet_kernel). "
                        "achieved = concurrent_launches x flops_per_launch / launch_ms = the rate the GPU sustains, vs the fp32 MFMA peak; "
                        "achieved_single_launch = flops_per_launch / launch_ms is one launch's share of it. "
                        "pipe_busy_model = MFMA issue time at spec clock (fp32 MFMAs at 157.3 TF, bf16 ones at 2516.6 TF) / launch time",
                "algorithmic": {"flops_per_launch": launch_flops, "achieved": alg_tf, "ratio_to_peak": alg_tf / PEAK_FP32_MFMA_TFLOPS,
                                "note": "direct-conv FLOPs (2*C_out*taps*C_in*L_out), all concurrent launches / launch time; Winograd F(4,5) issues 0.45x of them, so this can exceed the MFMA peak"},
                "launches_per_forward": chunks}

    out = {
        "metric": "guided trajectories/sec (H=64, 100 denoise steps), 32-robot Empty map",
        "value": value, "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.scaling}-scaling over {world} GPU(s): "
                               f"{n_robots}-robot EnvEmpty2D circle r=0.8, {RPG} robots/GPU x B={B} samples, H=64, "
                               f"T={T}+1 DDPM steps, 20 guide iterations on {ceil(0.5 * T) + 1} guided steps, "
                               f"{n_robots - 1} x 63 soft-constraint points per robot",
                   "n_robots": n_robots, "robots_per_gpu": RPG, "samples_per_robot": B, "horizon": H,
                   "diffusion_steps": T, "trajectories_per_step": n_traj_local * world,
                   "parallelism": f"robots sharded x{world}; 1 all-gather of [{RPG},64,2] fp32 per round" if world > 1
                   else "single GPU", "noise": "in-kernel Philox4x32-10 keyed by global trajectory index", "weights": "random-init (numpy PCG64 seed 0)"},
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        out["cpu_baseline"] = cpu_baseline(T, B, n_robots, args.cpu_budget_s)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
