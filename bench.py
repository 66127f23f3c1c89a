#!/usr/bin/env python
"""bench.py -- guided trajectories/sec of the MI355X sampler on BASELINE.json's headline config.

    python bench.py --gpus N --steps K --warmup W          (N > 1 re-launches itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is ONE planning round of the hot path: (all-gather of the robots' best paths ->) device-side soft-constraint
table -> one guided DDPM sampling call (T=100 denoise steps + 1 no-noise step, 20 guide iterations on the 51 guided
steps) for every local robot's B=64 samples -> best-path selection for the next round.  N=1 is BASELINE.json's headline
workload: 32 robots on the Empty map (circle r=0.8) = 2048 trajectories, each robot soft-constrained by the other 31
(31 x 63 = 1953 points).  N>1, `--scaling strong` (default, the headline): the SAME 32-robot instance, the metric's own
workload, sharded 32/N robots per GPU (512 trajectories per GPU at N=4, 256 at N=8); the weak-scaling variant (32 robots
PER GPU of one 32N-robot instance; the pairwise term grows with N) is timed right after it and reported in the same line as
`weak_scaling` (`--scaling weak` makes it the headline instead).  Either way ONE RCCL all-gather of [robots/GPU,64,2] fp32
per rank per round.  UNet weights are synthetic random-init (numpy PCG64), Gaussian noise is drawn in-kernel (Philox keyed
by the global trajectory index, so every rank's rows equal the unsharded run's), inputs are resident in HBM before the
timed region.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import glob
import socket
import subprocess
import sys
import threading
import time
from math import ceil

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402

H, D = 64, 4
PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
PEAK_F16_MFMA_TFLOPS = 2516.6      # dense fp16 / bf16 MFMA peak (1024 FLOP/clk/SIMD): 16x the fp32 MFMA rate
PEAK_HBM_GBPS = 8000.0             # HBM3E spec (6.3 TB/s achievable, same guide)
DOMINANT_KERNEL = "UNET"           # unet_kernel: the whole TemporalUnet forward in one launch (all 25 convs + GN/Mish)
HEADLINE_ROBOTS = 32               # BASELINE.json: 32-robot Empty map
PROF_UNET, PROF_STEP_GUIDED, PROF_STEP_PLAIN, PROF_UNET_FUSED = 0, 1, 2, 3     # include/mmd_amd_debug.h


# unet_kernel runs at the package power limit (profiles/r03_power_probe.txt): a loop of nothing but v_mfma_f32_16x16x32_f16 at
# the kernel's occupancy throttles to 2.05 GHz / 1.28 kW and sustains this rate, not the 2.4 GHz spec peak
POWER_LIMITED_F16_MFMA_TFLOPS = 1981.0
SPEC_SCLK_MHZ = 2400.0


class PowerSampler(threading.Thread):
    """Shader clock (MHz) and socket power (W) of the GPU while something runs: amdgpu sysfs every 20 ms (pp_dpm_sclk's
    current level, hwmon power1_input / power1_average), or `rocm-smi --showclocks --showpower` when sysfs is not
    there.  Of several cards, the one drawing the most power is reported.  Everything is best effort: no reading -> None."""

    def __init__(self, period=0.02, drm_root="/sys/class/drm"):
        super().__init__(daemon=True)
        self.period, self.halt, self.samples = period, threading.Event(), {}
        self.cards = []
        for dev in sorted(glob.glob(os.path.join(drm_root, "card[0-9]*", "device"))):
            pw = sorted(glob.glob(dev + "/hwmon/hwmon*/power1_input")) or sorted(glob.glob(dev + "/hwmon/hwmon*/power1_average"))
            if os.path.exists(dev + "/pp_dpm_sclk") and pw:
                self.cards.append((dev + "/pp_dpm_sclk", pw[0], os.path.join(os.path.dirname(pw[0]), "power1_cap")))
        self.source = "amdgpu sysfs (pp_dpm_sclk, hwmon power1)" if self.cards else "rocm-smi --showclocks --showpower"

    @staticmethod
    def _sysfs(card):
        sclk = None
        with open(card[0]) as f:
            for line in f:
                if "*" in line:
                    sclk = float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        with open(card[1]) as f:
            watts = float(f.read().strip()) * 1e-6
        return sclk, watts

    @staticmethod
    def _smi():
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
        sclk = watts = None
        for line in out.splitlines():
            if "sclk" in line and "(" in line:
                sclk = float(line.split("(")[1].lower().split("mhz")[0])
            elif "Power (W)" in line:
                watts = float(line.split(":")[-1])
        return sclk, watts

    def run(self):
        while not self.halt.is_set():
            try:
                if self.cards:
                    for i, card in enumerate(self.cards):
                        self.samples.setdefault(i, []).append(self._sysfs(card))
                else:
                    self.samples.setdefault(0, []).append(self._smi())
            except Exception:      # noqa: BLE001  (a monitoring read must never take the benchmark down)
                pass
            self.halt.wait(self.period)

    def finish(self, skip_s=0.0):
        """Stop; mean clock / power of the busiest card, ignoring the first skip_s seconds (power readings lag)."""
        self.halt.set()
        self.join(timeout=15)
        best = None
        for i, rows in self.samples.items():
            rows = [r for r in rows[int(skip_s / self.period) if self.cards else 0:] if r[0] is not None and r[1] is not None]
            if rows and (best is None or np.mean([r[1] for r in rows]) > best["package_watts"]):
                best = {"sclk_mhz": float(np.mean([r[0] for r in rows])), "package_watts": float(np.mean([r[1] for r in rows])),
                        "samples": len(rows), "source": self.source}
                try:
                    with open(self.cards[i][2]) as f:
                        best["package_cap_watts"] = float(f.read().strip()) * 1e-6
                except Exception:      # noqa: BLE001
                    pass
        return best


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N>1 headline: strong = the metric's 32-robot instance sharded 32/N robots per GPU; weak = 32 robots "
                         "per GPU of a 32N-robot instance (the other one is timed too and reported alongside)")
    ap.add_argument("--robots-per-gpu", type=int, default=0, help="override (0 = 32/N for strong, 32 for weak)")
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--diffusion-steps", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-second-scaling", action="store_true", help="N>1: time only the headline scaling mode")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    ap.add_argument("--cpu-baseline-only", action="store_true", help="(internal) print the cpu_baseline object and exit: no GPU touched")
    ap.add_argument("--no-power-probe", action="store_true",
                    help="skip the 1.5 s of whole-batch launches timed with the clock / power sampled (outside the timed region; "
                         "~6000 extra unet_kernel launches that would swamp a rocprofv3 trace of the command)")
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
    for these small tensors); the rest of the budget then goes into more steps at the best thread count (spread over the
    guided and the unguided half of the schedule), and that longer sample is what is reported."""
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
        if results and time.perf_counter() - t_start > 0.6 * budget_s:
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
    # the longer sample at the best thread count: guided / unguided steps spread over their halves of the schedule
    torch.set_num_threads(best["threads"])
    tg, tu = [], []
    gi = list(np.linspace(tsg - 1, 0, 8).astype(int))
    ui = list(np.linspace(T - 1, tsg, 8).astype(int))
    k = 0
    while time.perf_counter() - t_start < budget_s and k < 8:
        tg.append(timed(int(gi[k]), guide))
        tu.append(timed(int(ui[k]), None))
        k += 1
    if tg:
        est = n_guided * float(np.mean(tg)) + n_unguided * float(np.mean(tu))
    else:
        est = best["est_seconds_per_robot_call"]
    return {"value": B / est, "unit": "trajectories/s", "cores": best["threads"], "kind": "port",
            "cpu_model": cpu_model_name(), "logical_cpus": n_cpus,
            "sample": f"1 of {n_robots} robots (B={B}, {grp.q.shape[0]} soft-constraint points): {len(tg)} guided + {len(tu)} "
                      f"unguided DDPM steps at {best['threads']} threads (the best of a thread sweep that timed 2 + 2 steps per "
                      f"count), {sum(tg) + sum(tu):.1f} s of CPU work, extrapolated to the {n_guided}+{n_unguided}-step call "
                      f"(robots are sequential in the reference)",
            "est_seconds_per_robot_call": est, "thread_sweep": results}


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: re-exec under torch.distributed.run, one rank per GPU."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execvp(sys.executable, cmd)


def union_length(intervals):
    tot, cur_s, cur_e = 0.0, None, None
    for a, b in sorted(intervals):
        if cur_e is None or a > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = a, b
        else:
            cur_e = max(cur_e, b)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def run_mode(args, scaling, rank, world, dev, rehearsal, with_roofline, cpu_job=None):
    """Time args.steps planning rounds of one scaling mode with NOTHING attached (pass 1: `value`); with_roofline: the same
    rounds once more with the launch profiler and the clock / power sampler attached (pass 2: secondary launch statistics) and
    the whole-batch kernel back to back (power probe).  Returns (value, ms_per_step, config, roofline objects or None)."""
    import ctypes as C
    from mmd_amd import _lib, synth
    from mmd_amd.diffusion_model import GaussianDiffusionModel
    from mmd_amd.multi_robot import MultiRobotSampler
    from mmd_amd.temporal_unet import TemporalUnet

    T, B = args.diffusion_steps, args.samples
    if args.robots_per_gpu:
        RPG = args.robots_per_gpu
    elif scaling == "strong":
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

    def timed_rounds(seed0):
        nonlocal paths_local
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            trajs, paths_local = sampler.plan_round(paths_local, seed=seed0 + k)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            dt = float(tt.item())
        assert torch.isfinite(trajs).all()
        return dt

    lib = _lib.load()
    n_traj_local = RPG * B
    # the launch shape the library will use: concurrent stream chunks per UNet / step launch (2 from 2048 trajectories on)
    chunks = lib.mmd_sampler_stream_chunks(sampler.n_streams, RPG, B)
    for w in range(args.warmup):
        _, paths_local = sampler.plan_round(paths_local, seed=1000 + w)
    # ---- pass 1: the timed region.  No profiler, no sampler thread: exactly args.steps rounds between two barriers.
    dt = timed_rounds(0)
    ms_per_step = dt / args.steps * 1e3
    value = args.steps * n_traj_local * world / dt
    config = {"workload": f"{scaling}-scaling over {world} GPU(s): "
                          f"{n_robots}-robot EnvEmpty2D circle r=0.8, {RPG} robots/GPU x B={B} samples, H=64, "
                          f"T={T}+1 DDPM steps, 20 guide iterations on {ceil(0.5 * T) + 1} guided steps, "
                          f"{n_robots - 1} x 63 soft-constraint points per robot",
              "n_robots": n_robots, "robots_per_gpu": RPG, "samples_per_robot": B, "horizon": H,
              "diffusion_steps": T, "trajectories_per_step": n_traj_local * world,
              "parallelism": f"robots sharded x{world}; 1 all-gather of [{RPG},64,2] fp32 per round" if world > 1
              else "single GPU", "noise": "in-kernel Philox4x32-10 keyed by global trajectory index",
              "weights": "random-init (numpy PCG64 seed 0)"}
    if not with_roofline:
        return value, ms_per_step, config, None, None

    # ---- roofline of the dominant kernel from the timed region alone: MFMA issue time of one round at the spec clock / the
    # round's wall time.  Every conv runs as f16x2: 3 fp16 MFMA FLOPs per fp32 GEMM FLOP, (T + 1) forwards of every local
    # trajectory per round, whatever the launch shape.
    n_fwd = T + 1
    flops_traj = lib.mmd_unet_flops_per_trajectory()                 # algorithmic (direct-conv, fp32) FLOPs per forward
    mfma_traj = lib.mmd_unet_mfma_flops_per_trajectory()             # fp32 GEMM FLOPs the kernel runs (padding included)
    h_traj = lib.mmd_unet_f16x2_flops_per_trajectory()               # ... of which as f16x2 on the fp16 pipe (all of it)
    assert mfma_traj == h_traj, "an fp32 MFMA path is back: price it at the fp32 MFMA peak"
    issued_round = 3.0 * h_traj * n_traj_local * n_fwd               # fp16 MFMA FLOPs issued per round (per GPU)
    issue_ms_round = issued_round / (PEAK_F16_MFMA_TFLOPS * 1e12) * 1e3
    frac = issue_ms_round / ms_per_step
    launches_round = n_fwd * chunks
    n_launch = n_traj_local // chunks                                # trajectories per launch
    busy_s = issue_ms_round * 1e-3 / launches_round                  # MFMA issue time of ONE launch at spec clock

    # ---- pass 2 (secondary): the same rounds with every UNet launch and step-kernel launch of two consecutive DDPM steps out
    # of every twelve (all stream chunks) bracketed by a HIP event pair on the stream it is launched on, and the shader clock /
    # socket power sampled.  The brackets slow the bracketed launches down (they serialise against the other chunk), which is
    # why none of this enters `value` or `frac`.
    prof = C.c_void_p()
    window, stride = 2 * chunks, 6
    per_call = n_fwd * chunks
    max_pairs = 2 * (per_call // (window * stride) + 1) * window * max(args.steps, 1)
    _lib.check(lib.mmd_profiler_create_windowed(C.byref(prof), max_pairs, stride, window, per_call))
    model.profiler = prof
    watch = PowerSampler() if rank == 0 and not rehearsal else None
    if watch:
        watch.start()
    dt2 = timed_rounds(10_000)
    power_timed = watch.finish() if watch else None
    model.profiler = None

    def intervals(kind):
        a = (C.c_double * max_pairs)()
        b = (C.c_double * max_pairs)()
        n = C.c_int()
        _lib.check(lib.mmd_profiler_intervals(prof, kind, a, b, max_pairs, C.byref(n)))
        return [(a[i] * 1e-3, b[i] * 1e-3) for i in range(n.value)]       # seconds

    iv_plain, iv_fused = intervals(PROF_UNET), intervals(PROF_UNET_FUSED)
    iv_g, iv_p = intervals(PROF_STEP_GUIDED), intervals(PROF_STEP_PLAIN)
    _lib.check(lib.mmd_profiler_destroy(prof))
    iv_unet = iv_plain + iv_fused

    def mean_ms(iv):
        return float(np.mean([b - a for a, b in iv])) * 1e3 if iv else None

    dur = [b - a for a, b in iv_unet]
    union_s = union_length(iv_unet)                      # wall time during which at least one bracketed UNet launch runs
    bracketed = {
        "note": "pass 2 (same rounds, profiler attached; NOT the timed region): HIP-event intervals of the bracketed launches; "
                "bracketed launches run slower than unbracketed ones, so union x launches may exceed the round time",
        "ms_per_step_with_profiler": dt2 / args.steps * 1e3,
        "unet_launch_ms": {"forward_only": mean_ms(iv_plain), "forward_plus_fused_unguided_step": mean_ms(iv_fused)},
        "launches_timed": len(dur), "measured_concurrency": sum(dur) / union_s if dur else None,
        "union_ms_per_launch": union_s / len(dur) * 1e3 if dur else None,
        "pipe_busy_in_union": busy_s * len(dur) / union_s if dur else None,
    }
    # outside the timed region: the same kernel as ONE launch of all local trajectories, back to back on one stream, with the
    # shader clock / socket power / throttler residencies sampled (and the CPU baseline running beside it in its own process:
    # it needs the host cores only, and the GPU stays busy for whoever watches it)
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
    solo_s, solo_n = e0.elapsed_time(e1) / 20 * 1e-3, 20
    sustained, cpu_result = None, None
    if rank == 0 and not rehearsal and not args.no_power_probe:
        job = cpu_job() if cpu_job else None               # subprocess handle (or None)
        throttle0 = throttle_snapshot()
        watch = PowerSampler()
        watch.start()
        t_start, reps, batch = time.perf_counter(), 0, max(20, int(0.25 / solo_s))
        e0.record()
        while True:
            for _ in range(batch):
                unet(xs, 50)
            reps += batch
            torch.cuda.synchronize()                       # (bounds the launch queue; one sync per ~0.25 s of kernels)
            el = time.perf_counter() - t_start
            if (el >= 1.5 and (job is None or job.poll() is not None)) or el > args.cpu_budget_s + 90:
                break
        e1.record()
        torch.cuda.synchronize()
        sustained = watch.finish(skip_s=0.5)
        throttle1 = throttle_snapshot()
        if sustained:
            sustained.update({"launch_ms": e0.elapsed_time(e1) / reps, "launches": reps, "trajectories": n_traj_local,
                              "seconds": time.perf_counter() - t_start})
            if throttle0 and throttle1:
                import gpu_throttle
                sustained["throttle_residency"] = gpu_throttle.residency(throttle0, throttle1)
                sustained["throttle_active_at_end"] = {k: v for k, v in throttle1.items() if k.startswith("active_")}
                sustained["power_cap_info"] = throttle1.get("power_cap")
        solo_s, solo_n = e0.elapsed_time(e1) / reps * 1e-3, reps
        if job is not None:
            cpu_result = cpu_job_result(job)
    elif cpu_job:
        cpu_result = cpu_job_result(cpu_job())
    pmc_ref = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path):
        with open(pmc_path) as f:
            pmc_ref = json.load(f)
    roofline = {
        "bound": "mfma",
        "kernel": "unet_kernel<4>: whole TemporalUnet forward for 4 trajectories per workgroup (launches of <= 512 trajectories: unet_kernel<2>, two per workgroup; 12 ResidualTemporalBlocks, 2 down / 2 up convs, final block; every conv a direct convolution as an fp16 two-piece split of fp32 (f16x2, 3 MFMAs per product, fp32 accumulate) on the fp16 matrix pipe; the unguided DDPM steps ride in its tail; GroupNorm + Mish + time bias + residuals fused, activations in LDS/registers)",
        "achieved": frac * PEAK_F16_MFMA_TFLOPS, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": frac,
        "frac_definition": "fp16 MFMA FLOPs the kernel issues per planning round (3 per fp32 GEMM FLOP of an f16x2 conv x "
                           "(T + 1) forwards x local trajectories) / 2516.6 TFLOP/s = MFMA issue time per round at the spec "
                           "clock, / ms_per_step of the timed region (pass 1, nothing attached): the matrix pipe's busy "
                           "fraction over the WHOLE round, step kernels and launch gaps included; achieved = frac x peak",
        "mfma_issue_ms_per_round": issue_ms_round, "ms_per_step": ms_per_step, "unet_launches_per_round": launches_round,
        "mfma_issue_ms_per_launch": busy_s * 1e3, "trajectories_per_launch": n_launch, "stream_chunks": chunks,
        "useful_frac_of_f16_pipe": flops_traj * n_traj_local * n_fwd / (PEAK_F16_MFMA_TFLOPS * 1e12) / (ms_per_step * 1e-3),
        "flops_per_trajectory_forward": {"algorithmic_direct_conv_fp32": flops_traj, "fp32_gemm_issued": mfma_traj,
                                         "fp16_mfma_issued": 3.0 * h_traj},
        "traffic": None,
        "traffic_note": "not measured by this run (PMC counters need rocprofv3 passes of their own); the last committed passes are "
                        "under pmc_reference",
        "pmc_reference": pmc_ref,
        "bracketed_launches": bracketed,
        "whole_batch_single_launch": {"trajectories": n_traj_local, "launch_ms": solo_s * 1e3,
                                      "pipe_busy": busy_s * chunks / solo_s,
                                      "note": f"the same kernel as one launch of all local trajectories, {solo_n} back to back on one stream, outside the timed region"},
        "power": {
            "pass2_rounds": power_timed, "kernel_back_to_back": sustained,
            "frac_at_measured_clock": None if not (sustained and sustained.get("sclk_mhz")) else
            busy_s * chunks * SPEC_SCLK_MHZ / sustained["sclk_mhz"] / (sustained["launch_ms"] * 1e-3),
            "note": "shader clock / socket power (amdgpu sysfs) and the firmware's throttler residencies (amdsmi violation "
                    "accumulators: ppt_pwr = package power tracker, *_thrm = thermal) while the whole-batch launch runs back to "
                    "back; frac_at_measured_clock = MFMA issue time at the sampled clock / launch time"},
    }
    # second kernel: the fused DDPM-step + guide kernel, HBM roofline on its algorithmic bytes
    step_bytes = 3.0 * 1024.0 * n_launch                     # x read + eps read + x write per trajectory and step
    guide = {"kernel": "ddpm_guide_kernel: posterior mean + 20 guide iterations (SDF gather, workspace walls, GP prior, 31 x 63 soft-constraint points) + noise + hard conditioning, one wave per trajectory",
             "bound": "hbm", "peak": PEAK_HBM_GBPS, "unit": "GB/s", "bytes_per_launch": step_bytes,
             "note": "algorithmic bytes = 3 KiB per trajectory and step; the kernel is bound by the latency of its 20 dependent guide iterations (VALU issue), not by HBM; launch times are pass-2 brackets"}
    for name, iv in (("guided", iv_g), ("unguided", iv_p)):
        if iv:
            d = float(np.mean([b - a for a, b in iv]))
            guide[name] = {"launch_ms": d * 1e3, "launches_timed": len(iv), "achieved": step_bytes / d / 1e9,
                           "frac": step_bytes / d / 1e9 / PEAK_HBM_GBPS}
    if not iv_p:
        guide["unguided"] = {"fused": "steps without guidance run inside the tail of the unet_kernel launch that produces their eps (no step-kernel launch)"}
    return value, ms_per_step, config, roofline, (guide, cpu_result)


def throttle_snapshot():
    """amdsmi violation accumulators (tools/gpu_throttle.py), or None when the binding / driver does not provide them."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import gpu_throttle
        return gpu_throttle.snapshot()
    except Exception:      # noqa: BLE001  (best effort, like PowerSampler)
        return None


def cpu_job_result(job):
    out, _ = job.communicate(timeout=600)
    for line in reversed(out.decode().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return {"error": "cpu baseline subprocess printed no JSON", "returncode": job.returncode}


def main():
    args = parse()
    if args.cpu_baseline_only:                              # (the subprocess of cpu_job below: host cores only)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        print(json.dumps(cpu_baseline(args.diffusion_steps, args.samples, HEADLINE_ROBOTS, args.cpu_budget_s)), flush=True)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args)                                   # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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

    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline

    def cpu_job():
        """The CPU baseline in a process of its own (host cores only), started when the GPU part's timed region is over."""
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--diffusion-steps", str(args.diffusion_steps),
               "--samples", str(args.samples), "--cpu-budget-s", str(args.cpu_budget_s)]
        return subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))

    value, ms, config, roofline, (guide, cpu_result) = run_mode(args, args.scaling, rank, world, dev, rehearsal, with_roofline=True,
                                                                cpu_job=cpu_job if want_cpu else None)
    out = {
        "metric": "guided trajectories/sec (H=64, 100 denoise steps), 32-robot Empty map",
        "value": value, "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": config,
        "roofline": roofline,
        "roofline_step_kernel": guide,
    }
    if rehearsal:
        out["rehearsal"] = "all ranks on ONE GPU over gloo (MMD_BENCH_REHEARSAL=1): exercises the N>1 code path, NOT a measurement"
    if world > 1 and not args.no_second_scaling and not args.robots_per_gpu:
        other = "weak" if args.scaling == "strong" else "strong"
        v2, ms2, cfg2, _, _ = run_mode(args, other, rank, world, dev, rehearsal, with_roofline=False)
        out[f"{other}_scaling"] = {"value": v2, "unit": "trajectories/s", "ms_per_step": ms2, "scaling": other, "config": cfg2}
    if want_cpu:
        out["cpu_baseline"] = cpu_result
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
