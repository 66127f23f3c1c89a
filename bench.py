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

`--workload {headline,config2,config3,config4,config5}` (default: headline) runs the same measurement on BASELINE.json's other
configs (`config.workload` carries BASELINE.json's own wording): config2 = 6-robot Empty circle without the inter-robot term, config3
= 10-robot Highways with it, config4 = the 4 robots of the 1x2 Empty-tile ensemble (MPDEnsemble planner calls: two tile models
composed along the horizon, a trajectory = 128 support points), config5 = the 64-robot Conveyor instance -- 8 robots per GPU under
`--gpus 8`, one 4096-trajectory instance under `--gpus 1`.  All with B = 64 samples, T = 100 + 1 steps.

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
HEADLINE_METRIC = "guided trajectories/sec (H=64, 100 denoise steps), 32-robot Empty map"


def _baseline_configs():
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["configs"]
    except Exception:      # noqa: BLE001  (the wording is then the table's own)
        return [None] * 5


_CFG = _baseline_configs()
# the synthetic instances of SURVEY 8d (tests/test_gpu_configs.py holds each against the oracle at this size)
WORKLOADS = {
    "headline": dict(robots=32, env="EnvEmpty2D", formation=("circle", 0.8), inter_robot=True, label=None,
                     ref="mmd/common/multi_agent_utils.py:146-154, env_empty_2d.py:25-54"),
    "config2": dict(robots=6, env="EnvEmpty2D", formation=("circle", 0.8), inter_robot=False,
                    label=_CFG[1] or "6-robot Empty circle map, no inter-robot term", ref="env_empty_2d.py:25-54"),
    "config3": dict(robots=10, env="EnvHighways2D", formation=("circle", 0.45), inter_robot=True,
                    label=_CFG[2] or "10-robot Highways map with inter-robot soft-constraint guidance",
                    ref="env_highways_2d.py:38-103, mmd_experiment_configs.py:142-156"),
    "config4": dict(robots=4, env="EnvEmptyNoWait2D", formation=("tiles_1x2",), inter_robot=False, ensemble=True,
                    label=_CFG[3] or "MPDEnsemble multi_tile 1x2 Empty grid, 4 robots", ref="inference_multi_agent.py:418-431, mpd_ensemble.py:335-429"),
    "config5": dict(robots=64, env="EnvConveyor2D", formation=("boundary",), inter_robot=True,
                    label=_CFG[4] or "64-robot Conveyor map, per-robot batch sharded across 8 GPUs",
                    ref="env_conveyor_2d.py:37-80, multi_agent_utils.py:157-173"),
}


def workload_starts_goals(w, n_robots):
    from mmd_amd import synth
    f = w["formation"]
    if f[0] == "circle":
        return synth.start_goal_circle(n_robots, f[1])
    if f[0] == "boundary":
        return synth.start_goal_boundary(n_robots)
    raise ValueError(f)


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
    ap.add_argument("--workload", choices=tuple(WORKLOADS), default="headline",
                    help="headline = BASELINE.json's metric (32-robot Empty map); config2..config5 = BASELINE.json's configs[1..4]")
    ap.add_argument("--robots-per-gpu", type=int, default=0, help="override (0 = robots/N for strong, all robots per GPU for weak)")
    ap.add_argument("--sequential-planners", action="store_true",
                    help="config4: the planner calls of a round one after the other (the reference's loop) instead of batched")
    ap.add_argument("--concurrent-planners", action="store_true",
                    help="config4: the planner calls of a round on one host thread + stream each (plan_concurrently) instead of batched")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over the dominant kernel that fill roofline.traffic")
    ap.add_argument("--streams", type=int, default=0, help="mmd_sampler_desc.n_streams of the sharded sampler (0 = the library's choice: 2 chunks above 512 trajectories, A/B: tools/gpu_streams.sh)")
    ap.add_argument("--guide-coop-max", type=int, default=0, help="mmd_sampler_desc.guide_coop_max (A/B: launches up to this size run the four-waves-per-trajectory guide kernel; 0 = the library's 512)")
    ap.add_argument("--ns2-max", type=int, default=0, help="mmd_unet_options.two_per_workgroup_max (A/B: launches up to this size run two trajectories per workgroup; 0 = the library's 512)")
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--diffusion-steps", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-second-scaling", action="store_true", help="N>1: time only the headline scaling mode")
    ap.add_argument("--cpu-budget-s", type=float, default=25.0)
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


def cpu_baseline(T, B, workload, budget_s):
    """The oracle's reference-SHAPED path (dense (n,B,H,2) CostConstraint broadcast + one autograd pass per cost term,
    torch-CPU UNet) for ONE robot of the workload's instance, timed on this host's cores on a bounded sample and
    extrapolated to the full 101-step call.  Robots are planned sequentially by the reference, so trajectories/s of
    one robot's call is the whole-round rate.  The thread count is swept upwards from 8 (an oversubscribed pool is slower
    for these small tensors); the rest of the budget then goes into more steps at the best thread count (spread over the
    guided and the unguided half of the schedule), and that longer sample is what is reported."""
    import cases_for_bench as cb
    from oracle import mmd_oracle as O
    w = WORKLOADS[workload]
    n_robots = w["robots"]
    sd, tb, gp, groups, hc = cb.oracle_workload_robot(w, T)
    tiles = 2 if w.get("ensemble") else 1                         # the ensemble steps its K tile models one after the other
    x = torch.from_numpy(cb.synth.synth_noise(91, (B, H, D))) * 0.5
    x = O.apply_hard_conditioning(x, hc)
    noise = torch.from_numpy(cb.synth.synth_noise(92, (B, H, D)))
    guide = lambda y: O.guide_grad_dense_autograd(y, gp, groups)     # noqa: E731
    tsg = ceil(0.5 * T)

    def timed(i, g):
        t0 = time.perf_counter()
        with torch.no_grad():
            O.ddpm_sample_step(sd, tb, x.clone(), hc, i, guide=g, n_guide_steps=20, t_start_guide=tsg, noise=noise,
                               noise_std_extra=0.5)
        return (time.perf_counter() - t0) * tiles

    n_cpus = os.cpu_count() or 1
    # oversubscribed pools are catastrophically slow for these small tensors (256 threads: 100x slower than 8), so the
    # sweep climbs from 8 and stops as soon as a setting is clearly worse than the best so far
    sweep = [t for t in (8, 16, 32, 64, 128) if t <= n_cpus] or [n_cpus]
    n_guided, n_unguided = tsg + 1, T - tsg                          # i = tsg-1 ... -1 guided; the rest unguided
    results, t_start = [], time.perf_counter()
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(n_cpus))

    def pin(nt):
        """the process (and the torch pool it starts) on the first nt logical CPUs it may use; returns that list"""
        cpus = allowed[:nt]
        if hasattr(os, "sched_setaffinity"):
            os.sched_setaffinity(0, cpus)
        torch.set_num_threads(nt)
        return cpus
    for nt in sweep:
        if results and time.perf_counter() - t_start > 0.4 * budget_s:
            break
        pin(nt)
        timed(T - 1, None)                                           # warm-up (thread pool, allocator)
        t_u = min(timed(T - 1 - k, None) for k in range(2))
        t_g = [timed(tsg - 1 - k, guide) for k in range(2)]
        est = n_guided * float(np.mean(t_g)) + n_unguided * t_u
        results.append({"threads": nt, "guided_step_s": float(np.mean(t_g)), "unguided_step_s": t_u,
                        "est_seconds_per_robot_call": est, "trajectories_per_s": B / est})
        if B / est < max(r["trajectories_per_s"] for r in results):      # past the best thread count: stop climbing
            break
    best = max(results, key=lambda r: r["trajectories_per_s"])
    # the longer sample at the best thread count: guided / unguided steps spread over their halves of the schedule
    pinned = pin(best["threads"])
    tg, tu = [], []
    n_pairs = 24                                                      # (as many of them as the budget allows: less to extrapolate)
    gi = list(np.linspace(tsg - 1, 0, n_pairs).astype(int))
    ui = list(np.linspace(T - 1, tsg, n_pairs).astype(int))
    order = [int(j) for j in np.argsort([(j * 7) % n_pairs for j in range(n_pairs)])]   # spread over the schedule whatever the count
    gi, ui = [gi[j] for j in order], [ui[j] for j in order]
    k = 0
    t_sample = time.perf_counter()                                    # (the sample has 60 % of the budget to itself, at least 3 pairs)
    while (time.perf_counter() - t_sample < 0.6 * budget_s or k < 3) and k < n_pairs:
        tg.append(timed(int(gi[k]), guide))
        tu.append(timed(int(ui[k]), None))
        k += 1
    if tg:
        est = n_guided * float(np.mean(tg)) + n_unguided * float(np.mean(tu))
    else:
        est = best["est_seconds_per_robot_call"]
    return {"value": B / est, "unit": "trajectories/s", "cores": best["threads"], "kind": "port",
            "cpu_model": cpu_model_name(), "logical_cpus": n_cpus,
            "pinned_cpus": f"{pinned[0]}-{pinned[-1]}" if pinned == list(range(pinned[0], pinned[-1] + 1)) else pinned,
            "host_state": "the GPU process is idle (blocked on this subprocess): every GPU measurement of the run is finished before it starts",
            "sample": f"1 of {n_robots} robots of the {workload} instance (B={B}, {sum(g.q.shape[0] for g in groups)} soft-constraint points"
                      f"{', x 2 tile models per step' if tiles == 2 else ''}): {len(tg)} guided + {len(tu)} "
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
    W = WORKLOADS[args.workload]
    if args.robots_per_gpu:
        RPG = args.robots_per_gpu
    elif scaling == "strong":
        if W["robots"] % world:
            raise SystemExit(f"--scaling strong needs {W['robots']} % gpus == 0")
        RPG = W["robots"] // world
    else:
        RPG = W["robots"]
    n_robots = RPG * world
    unet = TemporalUnet(state_dim=4, n_support_points=H, unet_input_dim=32, dim_mults=(1, 2, 4), two_per_workgroup_max=args.ns2_max)
    unet.load_state_dict(synth.synth_unet_state_dict(0))
    model = GaussianDiffusionModel(model=unet, variance_schedule="exponential", n_diffusion_steps=T, predict_epsilon=True)
    model.guide_coop_max = args.guide_coop_max
    starts, goals = workload_starts_goals(W, n_robots)
    sampler = MultiRobotSampler(model, starts, goals, env_id=W["env"], n_samples=B, rank=rank, world_size=world,
                                device=dev, inter_robot=W["inter_robot"], n_streams=args.streams)
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
    # the launch shape the library will use: concurrent stream chunks per UNet / step launch (2 above 512 trajectories)
    chunks = lib.mmd_sampler_stream_chunks(sampler.n_streams, RPG, B)
    for w in range(args.warmup):
        _, paths_local = sampler.plan_round(paths_local, seed=1000 + w)
    # ---- pass 1: the timed region.  No profiler, no sampler thread: exactly args.steps rounds between two barriers.
    dt = timed_rounds(0)
    ms_per_step = dt / args.steps * 1e3
    value = args.steps * n_traj_local * world / dt
    detail = (f"{scaling}-scaling over {world} GPU(s): {n_robots}-robot {W['env']} {' '.join(str(v) for v in W['formation'])}, "
              f"{RPG} robots/GPU x B={B} samples, H=64, T={T}+1 DDPM steps, 20 guide iterations on {ceil(0.5 * T) + 1} guided steps, "
              + (f"{n_robots - 1} x 63 soft-constraint points per robot" if W["inter_robot"] else "no inter-robot term"))
    config = {"workload": detail if W["label"] is None else W["label"], "workload_detail": detail, "workload_key": args.workload,
              "reference_shapes": W["ref"], "n_robots": n_robots, "robots_per_gpu": RPG, "samples_per_robot": B, "horizon": H,
              "diffusion_steps": T, "trajectories_per_step": n_traj_local * world,
              "parallelism": "single GPU" if world == 1 else
              (f"robots sharded x{world}; 1 all-gather of [{RPG},64,2] fp32 per round" if W["inter_robot"]
               else f"robots sharded x{world}; no exchange (the workload has no inter-robot term)"),
              "noise": "in-kernel Philox4x32-10 keyed by global trajectory index",
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
    # shader clock / socket power / throttler residencies sampled
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
        job = None                                         # (the CPU baseline runs AFTER every GPU measurement, the host idle: VERDICT r5)
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
    pmc_ref = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path):
        with open(pmc_path) as f:
            pmc_ref = json.load(f)
    # ---- HBM-side traffic of the dominant kernel, measured NOW: separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; the SQ
    # instruction counters) over the two kernels alone at the sampler's launch size, outside the timed region, rank 0 at N = 1 only
    weight_bytes = float(lib.mmd_unet_weight_bytes(unet.handle(T, dev)))
    algo_bytes = 2.0 * 1024.0 * n_launch + weight_bytes            # 1 KiB in + 1 KiB out per trajectory, the weight pack once
    pmc_live, pmc_log = (None, ["skipped (--no-pmc / N > 1 / rehearsal)"])
    if rank == 0 and world == 1 and not rehearsal and not args.no_pmc:
        pmc_live, pmc_log = measure_pmc(args.workload, n_launch, T)
    traffic, traffic_src = None, None
    u = (pmc_live or {}).get("UNET", {})
    if "FETCH_SIZE" in u and "WRITE_SIZE" in u:
        traffic = (2.0 * u["FETCH_SIZE"] + u["WRITE_SIZE"]) * 1024.0
        traffic_src = "this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over tools/pmc_kernels.py"
    elif pmc_ref and pmc_ref.get("UNET", {}).get("FETCH_SIZE_KiB") and pmc_ref["UNET"].get("trajectories_per_launch") == n_launch:
        traffic = (2.0 * pmc_ref["UNET"]["FETCH_SIZE_KiB"] + pmc_ref["UNET"]["WRITE_SIZE_KiB"]) * 1024.0
        traffic_src = "NOT this run (the live passes failed: see pmc_log): the last committed passes, profiles/pmc_latest.json"
    # ---- the roofline object (SURVEY 8d / VERDICT r5 #5).  frac = ALGORITHMIC FLOPs / time / peak: the direct-convolution fp32 FLOPs of
    # the (T + 1) forwards of a round over the round's wall time, against the dense fp16 MFMA peak (the pipe the kernel runs on).
    # pipe_occupancy is a different question -- how busy the matrix pipe is, counting the THREE fp16 MFMAs the f16x2 split issues per
    # fp32 product -- and is reported beside it with the hardware counter of the same run.
    algo_tflops_round = flops_traj * n_traj_local * n_fwd / (ms_per_step * 1e-3) / 1e12
    busy_pmc = None
    if u.get("SQ_VALU_MFMA_BUSY_CYCLES") and u.get("GRBM_GUI_ACTIVE"):
        # per-SIMD MFMA-busy cycles / the launch's cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs)
        busy_pmc = u["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * u["GRBM_GUI_ACTIVE"] / 8.0)
    detail = {
        "frac_definition": "algorithmic FLOPs (direct convolution, fp32: mmd_unet_flops_per_trajectory x (T + 1) forwards x local "
                           "trajectories) per round / ms_per_step of the timed region (pass 1, nothing attached) / 2516.6 TFLOP/s",
        "pipe_occupancy_definition": "issue_frac = fp16 MFMA FLOPs ISSUED per round (3 per fp32 GEMM FLOP of an f16x2 conv, padding "
                                     "included) / 2516.6 TFLOP/s / ms_per_step: the matrix pipe's busy fraction over the whole round at "
                                     "the spec clock; mfma_busy_pmc = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) of "
                                     "this run's PMC pass over the kernel alone at the sampler's launch size",
        "mfma_issue_ms_per_round": issue_ms_round, "unet_launches_per_round": launches_round, "mfma_issue_ms_per_launch": busy_s * 1e3,
        "flops_per_trajectory_forward": {"algorithmic_direct_conv_fp32": flops_traj, "fp32_gemm_issued": mfma_traj,
                                         "fp16_mfma_issued": 3.0 * h_traj},
        "traffic_unit": "bytes per launch of the dominant kernel (HBM / fabric side of the L2s)",
        "traffic_definition": "(2 x FETCH_SIZE + WRITE_SIZE) KiB: gfx950's FETCH_SIZE tallies the 128-byte requests of wide coalesced "
                              "reads at 64 bytes (MI355X_MICROARCH.md, HBM section), WRITE_SIZE is taken as reported",
        "algorithmic_bytes_definition": f"{n_launch} trajectories x (1 KiB in + 1 KiB out) + the packed weight / parameter block once "
                                        f"({weight_bytes / 1e6:.2f} MB: two fp16 pieces per weight, MFMA fragment order)",
        "wasted_note": "the weight block is fetched once per XCD L2 (8 x), not once per launch: the kernel keeps every activation on "
                       "chip, so what exceeds the algorithmic bytes is the 8 L2s' copies of the weights",
        "pmc_log": pmc_log, "pmc_live": pmc_live, "pmc_reference": pmc_ref, "bracketed_launches": bracketed,
        "power": {
            "pass2_rounds": power_timed, "kernel_back_to_back": sustained,
            "frac_at_measured_clock": None if not (sustained and sustained.get("sclk_mhz")) else
            busy_s * chunks * SPEC_SCLK_MHZ / sustained["sclk_mhz"] / (sustained["launch_ms"] * 1e-3),
            "note": "shader clock / socket power (amdgpu sysfs) and the firmware's throttler residencies (amdsmi violation "
                    "accumulators: ppt_pwr = package power tracker, *_thrm = thermal) while the whole-batch launch runs back to "
                    "back; frac_at_measured_clock = MFMA issue time at the sampled clock / launch time"},
    }
    roofline = {
        "bound": "mfma",
        "kernel": "unet_kernel<4> (<= 512 trajectories per launch: unet_kernel<2>, <= 256: unet_kernel<1>): the whole TemporalUnet forward in one launch, every conv "
                  "a direct convolution as an fp16 two-piece split of fp32 on the fp16 matrix pipe, fp32 accumulate",
        "achieved": algo_tflops_round, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": algo_tflops_round / PEAK_F16_MFMA_TFLOPS,
        "frac_is": "algorithmic fp32 direct-conv FLOPs of the round / round time / dense fp16 MFMA peak (not the issued f16x2 FLOPs: those are pipe_occupancy)",
        "pipe_occupancy": {"issue_frac": frac, "mfma_busy_pmc": busy_pmc, "f16_mfma_per_fp32_product": 3},
        "kernel_alone": {"trajectories": n_traj_local, "launch_ms": solo_s * 1e3, "launches_timed": solo_n,
                         "frac": flops_traj * n_traj_local / solo_s / 1e12 / PEAK_F16_MFMA_TFLOPS, "issue_frac": busy_s * chunks / solo_s,
                         "timed_with": "HIP events around back-to-back launches of the whole local batch, outside the timed region"},
        "in_loop_launch": {"trajectories": n_launch, "stream_chunks": chunks, "avg_launch_ms": mean_ms(iv_unet),
                           "measured_concurrency": bracketed["measured_concurrency"],
                           "timed_with": "HIP event pairs on the launching stream (pass 2: the same rounds with the profiler attached)"},
        "ms_per_step": ms_per_step, "trajectories_per_launch": n_launch, "stream_chunks": chunks,
        "traffic": traffic, "algorithmic_bytes_per_launch": algo_bytes, "wasted_ratio": None if traffic is None else traffic / algo_bytes,
        "traffic_source": traffic_src,
        "hbm_gbps_in_round": None if traffic is None else traffic * launches_round / (ms_per_step * 1e-3) / 1e9,
        "detail": detail,
    }
    # second kernel: the fused DDPM-step + guide kernel.  It moves 3 KiB per trajectory and step (0.4 % of the HBM roof):
    # its bound is the VALU issue port -- a SIMD issues one wave64 VALU instruction per 4 cycles -- so the fraction reported is
    # 4 cycles x wave-level VALU instructions per launch / (1024 SIMDs x launch cycles), alone (the PMC pass's own dispatches) and
    # in the loop (pass-2 brackets: beside the other stream chunk's UNet launch, which shares the issue port).
    step_bytes = 3.0 * 1024.0 * n_launch                     # x read + eps read + x write per trajectory and step
    gq = (pmc_live or {}).get("GUIDE") or {}
    insts = gq.get("SQ_INSTS_VALU") or ((pmc_ref or {}).get("GUIDE") or {}).get("SQ_INSTS_VALU")
    insts_src = "this run's PMC pass" if gq.get("SQ_INSTS_VALU") else "profiles/pmc_latest.json (the live pass failed or was skipped)"
    n_simd = 1024.0
    sclk_hz = (power_timed or {}).get("sclk_mhz", SPEC_SCLK_MHZ) * 1e6

    def valu_frac(seconds, hz):
        return None if not (insts and seconds) else 4.0 * insts / (n_simd * seconds * hz)
    guide = {"kernel": "ddpm_guide_kernel: posterior mean + 20 guide iterations (SDF gather, workspace walls, GP prior, soft-constraint slots) + noise + hard conditioning, one wave per trajectory",
             "bound": "valu", "peak": 1.0, "unit": "fraction of the VALU issue slots (one wave64 instruction per SIMD and 4 cycles)",
             "valu_instructions_per_launch": insts, "valu_instructions_source": insts_src, "trajectories_per_launch": n_launch,
             "hbm": {"bytes_per_launch": step_bytes, "note": "3 KiB per trajectory and step: not what bounds it"},
             "note": "20 dependent guide iterations of ~600 VALU instructions per wave; one wave per SIMD at 1024 trajectories per launch"}
    if gq.get("SQ_BUSY_CYCLES") or gq.get("GRBM_GUI_ACTIVE"):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8 = the launch's cycles
        cyc = gq["GRBM_GUI_ACTIVE"] / 8.0 if gq.get("GRBM_GUI_ACTIVE") else None
        guide["alone"] = {"launch_cycles": cyc, "frac": None if not (cyc and insts) else 4.0 * insts / (n_simd * cyc),
                          "valu_busy_of_wave_cycles": (gq["SQ_ACTIVE_INST_VALU"] / gq["SQ_WAVE_CYCLES"]) if gq.get("SQ_ACTIVE_INST_VALU") and gq.get("SQ_WAVE_CYCLES") else None,
                          "source": "this run's PMC pass over tools/pmc_kernels.py (the kernel with the GPU to itself)"}
    for name, iv in (("guided", iv_g), ("unguided", iv_p)):
        if iv:
            d = float(np.mean([b - a for a, b in iv]))
            guide[name if name == "unguided" else "in_loop"] = {
                "launch_ms": d * 1e3, "launches_timed": len(iv), "frac": valu_frac(d, sclk_hz) if name == "guided" else None,
                "sclk_mhz_used": sclk_hz / 1e6, "hbm_gbps": step_bytes / d / 1e9,
                "note": "pass-2 brackets: the launch runs beside the other stream chunk's UNet launch"}
    guide["achieved"] = (guide.get("in_loop") or {}).get("frac")
    guide["frac"] = guide["achieved"]
    if not iv_p:
        guide["unguided"] = {"fused": "steps without guidance run inside the tail of the unet_kernel launch that produces their eps (no step-kernel launch)"}
    return value, ms_per_step, config, roofline, (guide, cpu_result)


PMC_PASSES = ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE",
              "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE")


def measure_pmc(workload, n_launch, T, timeout_s=150):
    """rocprofv3 PMC passes over tools/pmc_kernels.py (the UNet forward and the guided step kernel alone, at the sampler's launch
    size), each counter set in a run of its own with --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes
    (FETCH_SIZE and WRITE_SIZE do not fit one pass).  Returns {kernel family: {counter: mean per dispatch}} + the log of what ran.
    Best effort: a pass that fails or times out leaves its counters out."""
    import csv
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, ["rocprofv3 not found"]
    out, log = {}, []
    base = tempfile.mkdtemp(prefix="mmd_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", REPS="6")
    for k, counters in enumerate(PMC_PASSES):
        d = os.path.join(base, f"p{k}")
        cmd = [exe, "--pmc", *counters.split(), "--kernel-trace", "-f", "csv", "-d", d, "-o", "pmc", "--", sys.executable,
               os.path.join(ROOT, "tools", "pmc_kernels.py"), workload, str(n_launch), str(T)]
        try:
            p = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, cwd="/tmp", env=env, start_new_session=True)
            try:
                _, err = p.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, 9)                          # (exactly the group this call started)
                p.communicate()
                log.append(f"{counters}: timed out after {timeout_s} s")
                continue
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                log.append(f"{counters}: rc {p.returncode}, {(err or b'').decode(errors='replace')[-200:]}")
                continue
            acc = {}
            with open(files[0]) as f:
                for row in csv.DictReader(f):
                    name = row.get("Kernel_Name", "")
                    fam = "UNET" if "unet_kernel" in name else "GUIDE" if "ddpm_guide" in name else None
                    if fam:
                        acc.setdefault((fam, row["Counter_Name"]), []).append(float(row["Counter_Value"]))
            for (fam, c), v in acc.items():
                v = v[len(v) // 3:]                          # (drop the first launches: cold L2 / instruction cache)
                out.setdefault(fam, {})[c] = sum(v) / len(v)
                out[fam]["dispatches"] = len(v)
            log.append(f"{counters}: ok")
        except Exception as e:      # noqa: BLE001  (a profiling pass must never take the benchmark down)
            log.append(f"{counters}: {type(e).__name__}: {e}")
    shutil.rmtree(base, ignore_errors=True)
    return out, log


def run_ensemble(args, rank, world, dev, rehearsal=False):
    """config4: the 4 robots of the 1x2 Empty-tile instance, each an MPDEnsemble planner (two tile models composed along the horizon,
    cross-conditioned at the tile boundary every step, mpd_ensemble.py:335-429, diffusion_ensemble.py:55-106) called once per step --
    the reference's own granularity: one planner per agent (inference_multi_agent.py:225-237), B = 64 samples a call, a trajectory =
    2 x 64 support points.  A step = the four planner calls, selection included.  Multi-GPU: robots sharded (no exchange)."""
    from mmd_amd import _lib, synth
    from mmd_amd.planners import MPDEnsemble, plan_batched, plan_concurrently
    T, B = args.diffusion_steps, args.samples
    W = WORKLOADS["config4"]
    if W["robots"] % world:
        raise SystemExit(f"config4 needs {W['robots']} % gpus == 0")
    RPG = W["robots"] // world
    sd = synth.synth_unet_state_dict(0)
    # the reference's multi_tile example (inference_multi_agent.py:418-431): skeletons alternate [[0,0],[0,1]] / [[0,1],[0,0]] (tile
    # transforms [col * 2, -row * 2], :148-151 -- the relative tile transform is +2 for agents 0 and 2, -2 for agents 1 and 3), starts /
    # goals given in the frame of the first / last tile and moved to the global frame (:196-199)
    starts_l = torch.tensor([[0, 0.8], [0, 0.3], [0, -0.3], [0, -0.8]])
    goals_l = torch.tensor([[0, -0.8], [0, -0.3], [0, 0.3], [0, 0.8]])
    planners = []
    for r in range(rank * RPG, (rank + 1) * RPG):
        sk = [[0, 0], [0, 1]] if r % 2 == 0 else [[0, 1], [0, 0]]
        tr = {j: torch.tensor([c * 2.0, -row * 2.0]) for j, (row, c) in enumerate(sk)}
        start, goal = starts_l[r % 4] + tr[0], goals_l[r % 4] + tr[1]
        planners.append((MPDEnsemble(model_ids=("EnvEmptyNoWait2D-RobotPlanarDisk",) * 2, transforms=tr, planner_alg="mmd",
                                     start_state_pos=start, goal_state_pos=goal, n_samples=B, model_state_dicts=[sd, sd],
                                     model_args=dict(n_diffusion_steps=T), device=dev, seed=18 + r), start, goal))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    mode = "sequential" if args.sequential_planners else "concurrent" if args.concurrent_planners else "batched"

    def rounds(n):
        # the four planner calls of a round are independent.  batched (default): ONE launch sequence for all of them
        # (planners.plan_batched: a 64-trajectory UNet launch costs what a 256-trajectory one does); --concurrent-planners: one host
        # thread + one stream per call; --sequential-planners: one after the other, the reference's loop
        out = None
        for _ in range(n):
            if mode == "sequential":
                for p, start, goal in planners:
                    out = p(start, goal)
            elif mode == "concurrent":
                out = plan_concurrently(planners)[-1]
            else:
                out = plan_batched(planners)[-1]
        return out
    rounds(args.warmup)
    barrier()
    t0 = time.perf_counter()
    out = rounds(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    assert out.trajs_iters.shape[-2] == 2 * H and torch.isfinite(out.trajs_iters[-1]).all()
    ms = dt / args.steps * 1e3
    n_traj = RPG * B * world
    lib = _lib.load()
    h_traj = lib.mmd_unet_f16x2_flops_per_trajectory()
    issued = 3.0 * h_traj * RPG * B * 2 * (T + 1)                    # two tile forwards per composed trajectory and step
    issue_ms = issued / (PEAK_F16_MFMA_TFLOPS * 1e12) * 1e3
    detail = (f"{W['robots']} robots x B={B} samples on the 1x2 EnvEmptyNoWait2D tile grid (tile offset 2.0), one MPDEnsemble planner "
              f"call per robot and step ({ {'sequential': 'one after the other', 'concurrent': 'the calls of a step issued concurrently, one stream each', 'batched': 'the calls of a step packed into ONE launch sequence (plan_batched), bitwise the sequential calls'}[mode] }): "
              f"K=2 tile models, T={T}+1 DDPM steps per tile, 20 guide iterations on {ceil(0.5 * T) + 1} guided "
              f"steps, cross-conditioning of the tile boundary after every tile step, post-sampling selection; a trajectory = 128 support points")
    config = {"workload": W["label"], "workload_detail": detail, "workload_key": "config4", "reference_shapes": W["ref"],
              "n_robots": W["robots"], "robots_per_gpu": RPG, "samples_per_robot": B, "horizon": 2 * H, "diffusion_steps": T,
              "trajectories_per_step": n_traj, "parallelism": "single GPU" if world == 1 else f"robots sharded x{world}; no exchange",
              "planner_calls": {"sequential": "sequential", "concurrent": "concurrent (plan_concurrently)", "batched": "batched (plan_batched)"}[mode],
              "noise": "in-kernel Philox4x32-10", "weights": "random-init (numpy PCG64 seed 0), the same for both tiles"}
    per_launch = B * RPG if mode == "batched" else B
    algo_tflops = lib.mmd_unet_flops_per_trajectory() * RPG * B * 2 * (T + 1) / (ms * 1e-3) / 1e12   # two tile forwards per trajectory and step
    roofline = {"bound": "mfma", "kernel": f"unet_kernel<1> ({per_launch}-trajectory launches, one per tile and step)", "peak": PEAK_F16_MFMA_TFLOPS,
                "unit": "TFLOP/s", "achieved": algo_tflops, "frac": algo_tflops / PEAK_F16_MFMA_TFLOPS,
                "frac_is": "algorithmic fp32 direct-conv FLOPs of the round (2 tile forwards x (T + 1) steps x trajectories) / round time / dense "
                           "fp16 MFMA peak; a launch of <= 256 trajectories (one per workgroup) costs 72 - 75 us whatever its size: the planner "
                           "call is latency bound (one workgroup's 25 dependent convs), which is why the calls of a round are packed into one launch sequence",
                "pipe_occupancy": {"issue_frac": issue_ms / ms, "mfma_busy_pmc": None, "f16_mfma_per_fp32_product": 3},
                "ms_per_step": ms, "unet_launches_per_round": (1 if mode == "batched" else RPG) * 2 * (T + 1),
                "trajectories_per_launch": per_launch, "traffic": None,
                "traffic_source": "PMC passes are wired for the sharded sampler's launches (tools/pmc_kernels.py); not collected for the planner-call workload"}
    return n_traj * args.steps / dt, ms, config, roofline


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
        print(json.dumps(cpu_baseline(args.diffusion_steps, args.samples, args.workload, args.cpu_budget_s)), flush=True)
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
               "--samples", str(args.samples), "--cpu-budget-s", str(args.cpu_budget_s), "--workload", args.workload]
        return subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))

    W = WORKLOADS[args.workload]
    if W.get("ensemble"):
        value, ms, config, roofline = run_ensemble(args, rank, world, dev, rehearsal)
        guide = None
    else:
        value, ms, config, roofline, (guide, _) = run_mode(args, args.scaling, rank, world, dev, rehearsal, with_roofline=True)
    out = {
        "metric": HEADLINE_METRIC if W["label"] is None else f"guided trajectories/sec (H=64, {args.diffusion_steps} denoise steps), {W['label']}",
        "value": value, "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32 (f16x2 split on the fp16 MFMA pipe: every fp32 product as 3 fp16 MFMAs, fp32 accumulate)", "data": "synthetic",
        "config": config,
        "roofline": roofline,
        "roofline_step_kernel": guide,
    }
    if rehearsal:
        out["rehearsal"] = "all ranks on ONE GPU over gloo (MMD_BENCH_REHEARSAL=1): exercises the N>1 code path, NOT a measurement"
    if world > 1 and not args.no_second_scaling and not args.robots_per_gpu and not W.get("ensemble"):
        other = "weak" if args.scaling == "strong" else "strong"
        v2, ms2, cfg2, _, _ = run_mode(args, other, rank, world, dev, rehearsal, with_roofline=False)
        out[f"{other}_scaling"] = {"value": v2, "unit": "trajectories/s", "ms_per_step": ms2, "scaling": other, "config": cfg2}
    if want_cpu:
        # the CPU baseline LAST: every GPU measurement above is finished and this process only waits for the subprocess
        torch.cuda.synchronize()
        out["cpu_baseline"] = cpu_job_result(cpu_job())
    if rank == 0:
        # the long-form measurement record (definitions, per-pass PMC logs, bracketed launches, power / clock samples) goes to a side
        # file; the line keeps the numbers
        detail = {}
        for key in ("roofline", "roofline_step_kernel"):
            if isinstance(out.get(key), dict) and "detail" in out[key]:
                detail[key] = out[key].pop("detail")
        if isinstance(out.get("cpu_baseline"), dict) and "thread_sweep" in out["cpu_baseline"]:
            detail["cpu_baseline_thread_sweep"] = out["cpu_baseline"].pop("thread_sweep")
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            path = os.path.join(ROOT, "gpurun_out", f"bench_detail_{args.workload}_n{world}.json")
            with open(path, "w") as f:
                json.dump(detail, f, indent=1)
            out["detail_file"] = os.path.relpath(path, ROOT)
        except OSError:
            pass
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
