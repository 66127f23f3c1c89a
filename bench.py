#!/usr/bin/env python
"""bench.py -- guided trajectories/sec of the MI355X sampler on BASELINE.json's headline config.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is ONE planning round of the hot path: (all-gather of the robots' best paths ->) device-side soft-constraint
table -> one guided DDPM sampling call (T=100 denoise steps + 1 no-noise step, 20 guide iterations on the 51 guided
steps) for every local robot's B=64 samples -> best-path selection for the next round.  N=1 is BASELINE.json's headline
workload: 32 robots on the Empty map (circle r=0.8) = 2048 trajectories, each robot soft-constrained by the other 31
(31 x 63 = 1953 points).  N>1 is weak scaling: 32 robots PER GPU of one 32N-robot instance (the pairwise term grows
with N), one RCCL all-gather of [32,64,2] fp32 per rank per round.  UNet weights are synthetic random-init (numpy
PCG64), Gaussian noise is drawn in-kernel (Philox), inputs are resident in HBM before the timed region.

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
DOMINANT_LAYER = "UNET"            # unet_kernel: the whole TemporalUnet forward in one launch (all 25 convs + GN/Mish)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--robots-per-gpu", type=int, default=32)
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--diffusion-steps", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    return ap.parse_args()


def cpu_baseline(T, B, n_robots, budget_s):
    """The oracle's reference-SHAPED path (dense (n,B,H,2) CostConstraint broadcast + one autograd pass per cost term,
    torch-CPU UNet) for ONE robot of the headline instance, timed on this host's cores on a bounded sample and
    extrapolated to the full 101-step call.  Robots are planned sequentially by the reference, so trajectories/s of
    one robot's call is the whole-round rate."""
    import cases_for_bench as cb
    from oracle import mmd_oracle as O
    cores = torch.get_num_threads()
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

    timed(T - 1, None)                                              # warm-up (thread pool, allocator)
    t_u = min(timed(T - 1 - k, None) for k in range(3))
    t_g, n_g, spent = [], 0, 0.0
    while n_g < 3 and (n_g == 0 or spent + (spent / n_g) < budget_s):
        dt = timed(tsg - 1 - n_g, guide)
        t_g.append(dt)
        spent += dt
        n_g += 1
    n_guided = tsg + 1                                              # i = tsg-1 ... -1
    n_unguided = T - tsg
    est = n_guided * float(np.mean(t_g)) + n_unguided * t_u
    return {"value": B / est, "unit": "trajectories/s", "cores": cores, "kind": "port",
            "sample": f"1 of {n_robots} robots (B={B}, {grp.q.shape[0]} soft-constraint points), {n_g} guided + 3 unguided "
                      f"DDPM steps timed ({np.mean(t_g):.2f} s / {t_u * 1e3:.0f} ms each) and extrapolated to the "
                      f"{n_guided}+{n_unguided}-step call; robots are sequential in the reference",
            "est_seconds_per_robot_call": est}


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

    T, B, RPG = args.diffusion_steps, args.samples, args.robots_per_gpu
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
    nl = lib.mmd_unet_num_layers()
    names = [lib.mmd_unet_layer_name(i).decode() for i in range(nl)]
    dom = [i for i, nme in enumerate(names) if nme == DOMINANT_LAYER]
    n_traj_local = RPG * B
    flops = [lib.mmd_unet_layer_flops(i) * n_traj_local for i in range(nl)]
    mfma_flops = [lib.mmd_unet_layer_mfma_flops(i) * n_traj_local for i in range(nl)]

    for w in range(args.warmup):
        _, paths_local = sampler.plan_round(paths_local, seed=1000 + w)
    # roofline of the dominant kernel: every one of its launches INSIDE the timed region is bracketed by a HIP event pair
    # on the stream it is launched on (mmd_unet_profile_layer).  Every 13th of its 101 launches per step is bracketed
    # (one launch per UNet forward): ~8 event pairs per step, < 0.5 % of the timed region.
    _lib.check(lib.mmd_unet_profile_layer(unet.handle(T), dom[0], len(dom) * (T + 1) * args.steps, 13))
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        trajs, paths_local = sampler.plan_round(paths_local, seed=k)
    barrier()
    dt = time.perf_counter() - t0
    dom_ms_c, dom_n = C.c_double(), C.c_int()
    _lib.check(lib.mmd_unet_profile_read(unet.handle(T), C.byref(dom_ms_c), C.byref(dom_n)))
    _lib.check(lib.mmd_unet_profile_layer(unet.handle(T), -1, 0, 1))
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(trajs).all()
    value = args.steps * n_traj_local * world / dt
    dom_ms = dom_ms_c.value
    dom_tf = flops[dom[0]] / (dom_ms * 1e-3) / 1e12

    # HBM traffic of the dominant kernel comes from rocprofv3 PMC passes (separate runs; tools/gpu_round.sh), committed
    # as profiles/pmc_latest.json: traffic = (2 * FETCH_SIZE + WRITE_SIZE) KiB (gfx950 FETCH_SIZE half-count correction)
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path) and n_traj_local == 2048:
        with open(pmc_path) as f:
            pmc = json.load(f).get(DOMINANT_LAYER)
        if pmc:
            traffic = (2.0 * pmc["FETCH_SIZE_KiB"] + pmc["WRITE_SIZE_KiB"]) * 1024.0
    roofline = {"bound": "mfma", "kernel": "unet_kernel: whole TemporalUnet forward for 4 trajectories per workgroup (12 ResidualTemporalBlocks, 2 down / 2 up convs, final block; fp32 MFMA GEMMs, the 25 k=5 convs as Winograd F(4,5); GroupNorm + Mish + time bias + residuals fused, activations in LDS/registers)",
                "achieved": dom_tf, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": dom_tf / PEAK_FP32_MFMA_TFLOPS,
                "traffic": traffic, "launch_ms": dom_ms, "launches_timed": dom_n.value,
                "flops_per_launch": flops[dom[0]],
                # `achieved` counts ALGORITHMIC FLOPs (direct-convolution definition).  The k=5 convs run as
                # Winograd F(4,5) in fp32, so the matrix pipe issues 0.45x of them and `frac` can exceed 1: `mfma_issued` is
                # the pipe's own utilisation (PMC SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 = flops_per_launch below).
                "note": "achieved = algorithmic (direct-conv) FLOPs / time; Winograd F(4,5) issues 0.45x of them, see mfma_issued",
                "mfma_issued": {"flops_per_launch": mfma_flops[dom[0]],
                                "achieved": mfma_flops[dom[0]] / (dom_ms * 1e-3) / 1e12,
                                "frac": mfma_flops[dom[0]] / (dom_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS},
                "launches_per_forward": nl}

    out = {
        "metric": "guided trajectories/sec (H=64, 100 denoise steps), 32-robot Empty map",
        "value": value, "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{n_robots}-robot EnvEmpty2D circle r=0.8, {RPG} robots/GPU x B={B} samples, H=64, "
                               f"T={T}+1 DDPM steps, 20 guide iterations on {ceil(0.5 * T) + 1} guided steps, "
                               f"{n_robots - 1} x 63 soft-constraint points per robot",
                   "n_robots": n_robots, "robots_per_gpu": RPG, "samples_per_robot": B, "horizon": H,
                   "diffusion_steps": T, "trajectories_per_step": n_traj_local * world,
                   "parallelism": f"robots sharded x{world}; 1 all-gather of [{RPG},64,2] fp32 per round" if world > 1
                   else "single GPU", "noise": "in-kernel Philox4x32-10", "weights": "random-init (numpy PCG64 seed 0)"},
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
