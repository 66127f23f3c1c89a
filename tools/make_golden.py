#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the GENUINE reference (imported from
/root/reference; build container only -- SURVEY.md §8c).  Inputs are regenerated from numpy-PCG64 seeds by
`mmd_amd.synth`; only seeds/parameters and the reference's OUTPUTS are stored.  No reference code is stored.

    python tools/make_golden.py            # rewrites tests/golden/*.npz

Fixtures:
  g1_schedules.npz     12 diffusion buffers for T in {25,50,100}
  g2_unet.npz          TemporalUnet eps for B=4 at t in {0,37,99}
  g3_sdf.npz           per map: float64 checksums of the 400x400 SDF grid + 4096 sampled (value, grad) cells
  g4_guide_terms.npz   clipped per-term guide gradients (objects, ws-boundary, GP, constraints) on Highways, B=8
  g5_guide.npz         full guide(x): Highways+constraints (B=8), Empty + 31x63 soft constraints (B=4)
  g6_sample_*.npz      run_inference chains with injected noise
  g7_local.npz         run_local_inference (3 noising / 3 denoising steps)
  g8_ensemble.npz      2-tile DiffusionsEnsemble.run_inference
  g10_multi_agent.npz  robot-robot collisions of a best-path set + per-sample conflict totals (least_collisions scan)
  g9_post.npz          post-sampling selection: collision/free split, smoothness, path length, SavGol smoothing
  g12_boundary.npz     outer-boundary contract: check_rr_collisions / compute_collision on the shapes CBS / PP pass
  g13_split_constraints.npz  MPDEnsemble.split_cost_constraints_to_tasks + the per-tile range / transform shift
  g14_extra_objects.npz      a map with extra objects (spheres + boxes): guide, extra-objects-only guide, occupancy
  g16_options.npz            options the planners leave at their defaults: clip_grad_rule 'value' / clip_grad False, scale_grad_by_std,
                             predict_epsilon False
  g19_trained_unet.npz / g19_trained_chains.npz   a TemporalUnet trained briefly with the reference's loss (it denoises), and
                             the guided chains / sensitivities / multi-seed final rows of two constraint cases run with it
  g20_ensemble3.npz          3-tile corner-turning heterogeneous MPDEnsemble instance (x and y hops, both directions, per-tile maps
                             and weights, constraint routing, per-tile free / collision split + combine_trajs)
  g22_ensemble3_local.npz    DiffusionsEnsemble.run_local_inference (the re-plan path) on the 3-tile instance: whole-seed q_sample, per-tile split
                             into the tile frames, 3 + 1 guided denoising steps, both directions
  g21_ensemble_task.npz      PlanningTaskEnsemble.compute_collision / infer_task_id_from_q on global positions over the three tiles
  g15_distribution_*.npz     guided sampling over 32 noise seeds: final rows of every sample, their position mean / covariance per
                             support point, free / collision split and soft-constraint violation counts (distribution-level parity)
"""
import os
import sys
from math import ceil

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from ref_harness import (make_model, make_guide, make_cost_constraint, make_task, injected_noise, quiet,   # noqa: E402
                         ddpm_sample_fn, TENSOR_ARGS)
from mmd_amd import synth                                                                                 # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
MINS, MAXS = synth.NORM_MINS, synth.NORM_MAXS
H, D = 64, 4
RADIUS_SOFT = 0.05 * 2.4        # mmd_params.py:52
# number of independent 1e-6 perturbation draws whose max is stored as `sens` (the amplification of the chaotic guided
# loop is heavy-tailed: the more draws, the better the estimate of how far two fp32 implementations can drift apart)
N_SENS_DRAWS = int(os.environ.get("MMD_SENS_DRAWS", "24"))


def normalize(x):
    return 2 * (x - MINS) / (MAXS - MINS) - 1


def hard_conds_for(start, goal):
    """dataset.get_hard_conditions(..., normalize=True) (mmd/datasets/trajectories.py:216-239)."""
    s = normalize(np.concatenate([start, np.zeros(2, np.float32)]).astype(np.float32))
    g = normalize(np.concatenate([goal, np.zeros(2, np.float32)]).astype(np.float32))
    return {0: torch.from_numpy(s.astype(np.float32)), H - 1: torch.from_numpy(g.astype(np.float32))}


def soft_points(paths, agent):
    """cbs.py:468-508 for equal start times."""
    q, tr = [], []
    for j in range(paths.shape[0]):
        if j == agent:
            continue
        for t in range(1, H):
            q.append(paths[j, t])
            tr.append((t, t + 1))
    return np.array(q, np.float32), np.array(tr, np.int64), np.full(len(q), RADIUS_SOFT, np.float32)


def g1():
    out = {}
    sd = synth.synth_unet_state_dict(0)
    for T in (25, 50, 100):
        with quiet():
            m = make_model(sd, T)
        for k, v in m.state_dict().items():
            if not k.startswith("model."):
                out[f"T{T}.{k}"] = v.numpy()
    # the cosine schedule (helpers.py:16-27; GaussianDiffusionModel(variance_schedule='cosine'), diffusion_model_base.py:70-75)
    from mmd.models.diffusion_models.diffusion_model_base import GaussianDiffusionModel
    from mmd.models.diffusion_models.temporal_unet import TemporalUnet
    for T in (25, 100):
        with quiet():
            m = GaussianDiffusionModel(model=TemporalUnet(state_dim=4, n_support_points=64, unet_input_dim=32, dim_mults=(1, 2, 4)),
                                       variance_schedule="cosine", n_diffusion_steps=T, predict_epsilon=True)
        for k, v in m.state_dict().items():
            if not k.startswith("model."):
                out[f"cosine.T{T}.{k}"] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "g1_schedules.npz"), **out)


def g2():
    sd = synth.synth_unet_state_dict(0)
    with quiet():
        m = make_model(sd, 100)
    x = torch.from_numpy(synth.synth_noise(5, (4, H, D)))
    out = {"weights_seed": 0, "x_seed": 5, "ts": np.array([0, 37, 99])}
    for t in (0, 37, 99):
        with torch.no_grad():
            out[f"eps_t{t}"] = m.model(x, torch.full((4,), t, dtype=torch.long), None).numpy()
    np.savez_compressed(os.path.join(OUT, "g2_unet.npz"), **out)


def g3():
    out = {}
    rng = np.random.Generator(np.random.PCG64(33))
    for env_id in ("EnvEmpty2D", "EnvHighways2D", "EnvConveyor2D", "EnvDropRegion2D"):
        with quiet():
            env, robot, task = make_task(env_id)
        g = env.grid_map_sdf_obj_fixed
        sdf, grad = g.sdf_tensor.numpy(), g.grad_sdf_tensor.numpy()
        idx = rng.integers(0, 400, size=(4096, 2))
        out[f"{env_id}.shape"] = np.array(sdf.shape)
        out[f"{env_id}.sum_sdf"] = np.float64(sdf.astype(np.float64).sum())
        out[f"{env_id}.sum_abs_grad"] = np.float64(np.abs(grad.astype(np.float64)).sum())
        out[f"{env_id}.idx"] = idx
        out[f"{env_id}.sdf"] = sdf[idx[:, 0], idx[:, 1]]
        out[f"{env_id}.grad"] = grad[idx[:, 0], idx[:, 1]]
    np.savez_compressed(os.path.join(OUT, "g3_sdf.npz"), **out)


def highways_case(agent=3, n_agents=10):
    starts, goals = synth.start_goal_circle(n_agents, 0.45)
    paths = synth.straight_line_paths(starts, goals, H)
    q, tr, r = soft_points(paths, agent)
    hard_q = np.array([[0.1, 0.2]], np.float32)
    hard_tr = np.array([[20, 27]], np.int64)
    hard_r = np.array([RADIUS_SOFT], np.float32)
    return starts, goals, (q, tr, r), (hard_q, hard_tr, hard_r)


def g4_g5():
    x = torch.from_numpy(synth.synth_noise(7, (8, H, D))) * 0.6
    _, _, soft, hard = highways_case()
    out = {"x_seed": 7, "x_scale": 0.6}
    for name, which in (("obj", ("obj",)), ("ws", ("ws",)), ("gp", ("gp",)), ("cons", ())):
        with quiet():
            guide, robot, task, env = make_guide("EnvHighways2D", MINS, MAXS, which=which, w_coll=1.0, w_smooth=1.0)
        if name == "cons":
            guide.add_extra_costs([make_cost_constraint(robot, *soft, True), make_cost_constraint(robot, *hard, False)],
                                  [1.0, 0.0])
            out["term_cons_soft"] = (-guide(x)).numpy()
            guide.reset_extra_costs()
            guide.add_extra_costs([make_cost_constraint(robot, *hard, False)], [1.0])
            out["term_cons_hard"] = (-guide(x)).numpy()
        else:
            out[f"term_{name}"] = (-guide(x)).numpy()
    np.savez_compressed(os.path.join(OUT, "g4_guide_terms.npz"), **out)

    out = {}
    with quiet():
        guide, robot, task, env = make_guide("EnvHighways2D", MINS, MAXS)
    guide.add_extra_costs([make_cost_constraint(robot, *soft, True), make_cost_constraint(robot, *hard, False)],
                          [2e-2, 2e-1])
    out["highways_B8"] = guide(x).numpy()
    # values beyond +-1 exercise the data-dependent clip of LimitsNormalizer.unnormalize
    x2 = torch.from_numpy(synth.synth_noise(8, (8, H, D))) * 1.1
    out["highways_B8_wide"] = guide(x2).numpy()
    guide.reset_extra_costs()
    with quiet():
        guide, robot, task, env = make_guide("EnvEmpty2D", MINS, MAXS)
    starts, goals = synth.start_goal_circle(32, 0.8)
    paths = synth.straight_line_paths(starts, goals, H)
    q, tr, r = soft_points(paths, 0)
    guide.add_extra_costs([make_cost_constraint(robot, q, tr, r, True)], [2e-2])
    x3 = torch.from_numpy(synth.synth_noise(9, (4, H, D))) * 0.5
    out["empty32_B4"] = guide(x3).numpy()
    np.savez_compressed(os.path.join(OUT, "g5_guide.npz"), **out)


CHAIN_ROWS = lambda T: sorted({0, 1, 2, T // 2, T // 2 + 1, T // 2 + 2, T - 1, T, T + 1})   # noqa: E731


def run_ref_inference(env_id, T, B, start, goal, cons, seed_xT, seed_steps, weights_seed=0, cutoff=0.05,
                      n_guide_steps=20, use_guide=True, perturb=0.0, perturb_seed=0, sd=None):
    """`perturb` > 0 multiplies the reference UNet's output by (1 + perturb * N(0,1)) at every step: an fp32-rounding
    sized disturbance (a different summation order) used to MEASURE how well-conditioned the reference's own map
    noise -> trajectory is for this case (stored next to the golden rows as `sens`)."""
    sd = synth.synth_unet_state_dict(weights_seed) if sd is None else sd
    with quiet():
        model = make_model(sd, T)
        guide, robot, task, env = make_guide(env_id, MINS, MAXS, cutoff_margin=cutoff)
    if perturb:
        gen = torch.Generator().manual_seed(perturb_seed)
        model.model.register_forward_hook(
            lambda mod, inp, out: out * (1 + perturb * torch.empty(out.shape).normal_(generator=gen)))
    costs, ws = [], []
    for (q, tr, r, soft) in cons:
        costs.append(make_cost_constraint(robot, q, tr, r, soft))
        ws.append(2e-2 if soft else 2e-1)                              # mpd.py:409-412, mmd_params.py:42-43
    guide.add_extra_costs(costs, ws)
    xT = synth.synth_noise(seed_xT, (B, H, D))
    steps = synth.synth_noise(seed_steps, (T + 1, B, H, D))
    with quiet(), injected_noise([xT] + list(steps)) as q:
        chain = model.run_inference(
            None, hard_conds_for(start, goal), n_samples=B, horizon=H, return_chain=True, sample_fn=ddpm_sample_fn,
            guide=guide if use_guide else None, n_guide_steps=n_guide_steps, t_start_guide=ceil(0.5 * T),
            noise_std_extra_schedule_fn=lambda x: 0.5, n_diffusion_steps_without_noise=1)
        assert len(q) == 0
    guide.reset_extra_costs()
    return chain.numpy()          # [T+2, B, H, D]


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def save_case(fname, meta, rows, *args, **kw):
    chain = run_ref_inference(*args, **kw)
    sens = np.zeros(len(rows))
    draws = []
    for ps in range(1, N_SENS_DRAWS + 1):   # the amplification is itself random: max over N_SENS_DRAWS perturbation draws
        pert = run_ref_inference(*args, perturb=1e-6, perturb_seed=ps, **kw)
        draws.append([rel_l2(pert[r], chain[r]) for r in rows])
        sens = np.maximum(sens, draws[-1])
    np.savez_compressed(os.path.join(OUT, fname), rows=np.array(rows), chain_rows=chain[rows], sens=sens,
                        sens_draws=np.array(draws), meta=np.array(meta))
    print("  ", fname, "final-row sensitivity to a 1e-6 relative UNet perturbation:", f"{sens[-1]:.2e}", flush=True)


def g6():
    # (a) config 2 shape: Empty, 6 robots on a circle, robot 0, soft constraints from the 5 others, T=50
    starts, goals = synth.start_goal_circle(6, 0.8)
    paths = synth.straight_line_paths(starts, goals, H)
    q, tr, r = soft_points(paths, 0)
    save_case("g6_sample_empty_T50.npz", [50, 8, 6, 0, 11, 12], CHAIN_ROWS(50),
              "EnvEmpty2D", 50, 8, starts[0], goals[0], [(q, tr, r, True)], 11, 12)
    # (b) config 3 shape: Highways, 10 robots small circle, robot 3, soft + one hard constraint, T=100
    starts, goals, soft, hard = highways_case()
    save_case("g6_sample_highways_T100.npz", [100, 8, 10, 3, 13, 14], CHAIN_ROWS(100),
              "EnvHighways2D", 100, 8, starts[3], goals[3], [(*soft, True), (*hard, False)], 13, 14)
    # (c) released-checkpoint step count: Empty, no constraints, T=25
    save_case("g6_sample_empty_T25_nocons.npz", [25, 4, 10, 0, 15, 16], CHAIN_ROWS(25),
              "EnvEmpty2D", 25, 4, starts[0], goals[0], [], 15, 16)
    # (d) config 0: single robot, 1 sample, T=50
    save_case("g6_sample_cfg0_T50_B1.npz", [50, 1, 1, 0, 17, 18], list(range(52)),
              "EnvEmpty2D", 50, 1, np.array([-0.8, 0], np.float32), np.array([0.8, 0], np.float32), [], 17, 18)
    # (e) north-star per-robot shape, short: Empty, 32 robots, robot 5, 31x63 soft points, T=25, B=4
    starts, goals = synth.start_goal_circle(32, 0.8)
    paths = synth.straight_line_paths(starts, goals, H)
    q, tr, r = soft_points(paths, 5)
    save_case("g6_sample_empty32_T25.npz", [25, 4, 32, 5, 19, 20], CHAIN_ROWS(25),
              "EnvEmpty2D", 25, 4, starts[5], goals[5], [(q, tr, r, True)], 19, 20)
    # (f) Conveyor map (config 5 shape, one robot's shard): 8 robots on the boundary, robot 2, T=50
    starts, goals = synth.start_goal_boundary(8)
    paths = synth.straight_line_paths(starts, goals, H)
    q, tr, r = soft_points(paths, 2)
    save_case("g6_sample_conveyor_T50.npz", [50, 4, 8, 2, 21, 22], CHAIN_ROWS(50),
              "EnvConveyor2D", 50, 4, starts[2], goals[2], [(q, tr, r, True)], 21, 22)
    # (g) prior only (planner_alg 'diffusion_prior'): no guide at all, T=100 -- the well-conditioned end-to-end case
    starts, goals = synth.start_goal_circle(6, 0.8)
    save_case("g6_sample_prior_T100.npz", [100, 8, 6, 1, 29, 30], CHAIN_ROWS(100),
              "EnvEmpty2D", 100, 8, starts[1], goals[1], [], 29, 30, use_guide=False)


def g6_full():
    """The FULL 27-row chain of the 32-robot case (g6_sample_empty32_T25, same inputs), so that every one of its 13 guided steps
    is teacher-forced by test_single_step_teacher_forced_golden (no sensitivity draws: one reference run)."""
    starts, goals = synth.start_goal_circle(32, 0.8)
    paths = synth.straight_line_paths(starts, goals, H)
    q, tr, r = soft_points(paths, 5)
    chain = run_ref_inference("EnvEmpty2D", 25, 4, starts[5], goals[5], [(q, tr, r, True)], 19, 20)
    ref = np.load(os.path.join(OUT, "g6_sample_empty32_T25.npz"))
    assert np.array_equal(chain[ref["rows"]], ref["chain_rows"]), "the full chain must contain the stored rows of g6"
    np.savez_compressed(os.path.join(OUT, "g6_full_empty32_T25.npz"), chain=chain, meta=ref["meta"])
    print("   g6_full: chain", chain.shape)


def g7():
    """run_local_inference (diffusion_model_base.py:353-421): noise the seed batch 3 steps, denoise 3 (+1)."""
    T, B = 50, 8
    sd = synth.synth_unet_state_dict(0)
    starts, goals, soft, hard = highways_case()
    with quiet():
        model = make_model(sd, T)
        guide, robot, task, env = make_guide("EnvHighways2D", MINS, MAXS)
    guide.add_extra_costs([make_cost_constraint(robot, *soft, True), make_cost_constraint(robot, *hard, False)],
                          [2e-2, 2e-1])
    # seed trajectories: straight line + small noise, UN-normalised as the caller passes them (cbs.py:424)
    a = np.linspace(0, 1, H, dtype=np.float32)[None, :, None]
    pos = starts[3][None, None] * (1 - a) + goals[3][None, None] * a
    seed = np.concatenate([np.repeat(pos, B, 0), np.zeros((B, H, 2), np.float32)], -1)
    seed = (seed + 0.02 * synth.synth_noise(23, (B, H, D))).astype(np.float32)
    qn = synth.synth_noise(24, (B, H, D))
    steps = synth.synth_noise(25, (4, B, H, D))

    def run(perturb=0.0, ps=0):
        handle = None
        if perturb:
            gen = torch.Generator().manual_seed(ps)
            handle = model.model.register_forward_hook(
                lambda mod, inp, out: out * (1 + perturb * torch.empty(out.shape).normal_(generator=gen)))
        with quiet(), injected_noise([qn] + list(steps)) as q:
            chain = model.run_local_inference(
                torch.from_numpy(seed), 3, 3, None, hard_conds_for(starts[3], goals[3]), n_samples=B, horizon=H,
                return_chain=True, sample_fn=ddpm_sample_fn, guide=guide, n_guide_steps=20,
                t_start_guide=ceil(0.5 * T), noise_std_extra_schedule_fn=lambda x: 0.5,
                n_diffusion_steps_without_noise=1)
            assert len(q) == 0
        if handle is not None:
            handle.remove()
        return chain.numpy()

    chain = run()
    sens = np.zeros(5)
    for ps in range(1, N_SENS_DRAWS + 1):
        pert = run(1e-6, ps)
        sens = np.maximum(sens, [rel_l2(pert[r], chain[r]) for r in range(5)])
    print("   g7 sensitivity per row:", sens)
    np.savez_compressed(os.path.join(OUT, "g7_local.npz"), chain=chain, sens=sens, meta=np.array([T, B, 10, 3, 23, 24, 25]))


def g8():
    """2-tile DiffusionsEnsemble.p_sample_loop (diffusion_ensemble.py:55-106): tiles at x-offsets 0 and 2, the end of
    tile 0 cross-conditioned onto the start of tile 1 (sample_functions.py:17-31).  Stores every chain row of both tiles
    and, like g6, the reference's own sensitivity `sens` (per row, max over 24 draws; `sens_draws` keeps all of them) to a relative 1e-6 perturbation of
    the UNet output."""
    from mmd.models.diffusion_models.diffusion_ensemble import DiffusionsEnsemble
    T, B = 25, 4
    sd = synth.synth_unet_state_dict(0)
    with quiet():
        models = {0: make_model(sd, T), 1: make_model(sd, T)}
        guides = {}
        for m in (0, 1):
            # separate guide objects (own extra-cost lists) over the same map; MPDEnsemble uses cutoff 0.01
            guides[m], robot, task, env = make_guide("EnvEmptyNoWait2D", MINS, MAXS, cutoff_margin=0.01)
    transforms = {0: torch.tensor([0.0, 0.0]), 1: torch.tensor([2.0, 0.0])}
    ens = DiffusionsEnsemble(models, transforms)
    start = np.array([-0.7, 0.3], np.float32)
    goal = np.array([0.6, -0.4], np.float32)          # in tile 1's local frame
    s = hard_conds_for(start, start)[0]
    g = hard_conds_for(goal, goal)[0]
    cross_conds = {(0, 1): (H - 1, 0)}
    # one soft constraint point per tile so that the guides differ
    cons = {0: (np.array([[0.2, 0.1]], np.float32), np.array([[30, 36]]), np.array([RADIUS_SOFT], np.float32)),
            1: (np.array([[-0.3, -0.1]], np.float32), np.array([[10, 14]]), np.array([RADIUS_SOFT], np.float32))}
    for m in (0, 1):
        guides[m].add_extra_costs([make_cost_constraint(robot, *cons[m], False)], [2e-1])
    x0 = [synth.synth_noise(26 + m, (B, H, D)) for m in (0, 1)]
    steps = synth.synth_noise(28, (T + 1, 2, B, H, D))
    draws = list(x0) + [steps[k, m] for k in range(T + 1) for m in (0, 1)]
    sample_kwargs = {m: dict(guide=guides[m], n_guide_steps=20, t_start_guide=ceil(0.5 * T),
                             noise_std_extra_schedule_fn=lambda x: 0.5) for m in (0, 1)}

    def run(perturb=0.0, ps=0):
        handles = []
        if perturb:
            gen = torch.Generator().manual_seed(ps)
            for m in (0, 1):
                handles.append(models[m].model.register_forward_hook(
                    lambda mod, inp, out: out * (1 + perturb * torch.empty(out.shape).normal_(generator=gen))))
        hard_conds = {0: {0: s.repeat(B, 1)}, 1: {H - 1: g.repeat(B, 1)}}
        with quiet(), injected_noise(draws) as q:
            x, chains = ens.p_sample_loop((B, H, D), hard_conds, cross_conds, n_diffusion_steps=T, return_chain=True,
                                          sample_fn=ddpm_sample_fn, n_diffusion_steps_without_noise=1,
                                          sample_kwargs=sample_kwargs)
            assert len(q) == 0
        for h in handles:
            h.remove()
        return {m: chains[m].transpose(0, 1).numpy().copy() for m in (0, 1)}     # [T+2, B, H, D]

    chains = run()
    sens = {m: np.zeros(T + 2) for m in (0, 1)}
    for ps in range(1, N_SENS_DRAWS + 1):
        pert = run(1e-6, ps)
        for m in (0, 1):
            sens[m] = np.maximum(sens[m], [rel_l2(pert[m][r], chains[m][r]) for r in range(T + 2)])
    print("   g8 final-row sensitivity:", sens[0][-1], sens[1][-1])
    np.savez_compressed(os.path.join(OUT, "g8_ensemble.npz"), final0=chains[0][-1], final1=chains[1][-1],
                        chain0_mid=chains[0][T // 2 + 1], chain1_mid=chains[1][T // 2 + 1], chain0=chains[0],
                        chain1=chains[1], sens0=sens[0], sens1=sens[1], meta=np.array([T, B, 26, 27, 28]))


def g9():
    """Post-sampling selection: PlanningTask.get_trajs_collision_and_free (tasks.py:236-311),
    compute_smoothness / compute_path_length (trajectory/metrics.py:7-39), smooth_trajs (trajectory_utils.py:31-40) on
    a Highways batch: the reference's own final samples of case (b) + straight lines that cross obstacles / leave the map."""
    from torch_robotics.trajectory.metrics import compute_path_length, compute_smoothness
    from mmd.common.trajectory_utils import smooth_trajs
    g = np.load(os.path.join(OUT, "g6_sample_highways_T100.npz"))
    final_n = torch.from_numpy(g["chain_rows"][-1])
    with quiet():
        guide, robot, task, env = make_guide("EnvHighways2D", MINS, MAXS)
    trajs = guide.dataset.unnormalize_trajectories(final_n)
    a = torch.linspace(0, 1, H)[None, :, None]
    ends = torch.tensor([[[-0.9, -0.9], [0.9, 0.9]], [[-0.6, 0.45], [0.6, 0.45]], [[-0.45, -0.9], [-0.45, 0.9]],
                         [[-0.99, 0.45], [1.05, 0.45]], [[0.0, 0.0], [0.45, 0.45]], [[-0.45, 0.45], [0.45, -0.45]]])
    lines = ends[:, 0:1, :] * (1 - a) + ends[:, 1:2, :] * a
    vel = torch.zeros_like(lines)
    vel[:, :-1] = (lines[:, 1:] - lines[:, :-1]) / (5.0 / 64)
    trajs = torch.cat((trajs, torch.cat((lines, vel), -1)))
    coll, coll_idxs, free, free_idxs, wp = task.get_trajs_collision_and_free(trajs, return_indices=True)
    out = {"trajs": trajs.numpy(), "free_idxs": free_idxs.numpy().reshape(-1), "coll_idxs": coll_idxs.numpy().reshape(-1),
           "waypoint_collisions": wp.numpy(), "smoothness": compute_smoothness(trajs, robot).numpy(),
           "path_length": compute_path_length(trajs, robot).numpy(), "smoothed": smooth_trajs(trajs).numpy()}
    print("   g9: free", out["free_idxs"], "coll", out["coll_idxs"])
    np.savez_compressed(os.path.join(OUT, "g9_post.npz"), **out)


def g10():
    """Multi-agent layer: the reference robot's check_rr_collisions on a 6-robot best-path set, and the
    conflict count CBS.get_conflicts would see for every sample of robot 0's batch (cbs.py:166-246, :446-458)."""
    with quiet():
        env, robot, task = make_task("EnvEmpty2D")
    starts, goals = synth.start_goal_circle(6, 0.8)
    paths = torch.from_numpy(synth.straight_line_paths(starts, goals, H))
    paths = paths + 0.03 * torch.from_numpy(synth.synth_noise(31, tuple(paths.shape)))     # break the symmetry
    coll, mid = robot.check_rr_collisions(paths.permute(1, 0, 2))                           # (H, n, n)
    g = np.load(os.path.join(OUT, "g6_sample_empty_T50.npz"))
    guide, _, _, _ = make_guide("EnvEmpty2D", MINS, MAXS)
    samples = guide.dataset.unnormalize_trajectories(torch.from_numpy(g["chain_rows"][-1]))   # robot 0's batch [8,H,4]
    # the guided samples avoid everybody; make half of the batch noisy straight lines so that the counts differ
    line = torch.from_numpy(synth.straight_line_paths(starts, goals, H))[0]
    for b in range(4, 8):
        samples[b, :, :2] = line + 0.01 * (b - 3) * torch.from_numpy(synth.synth_noise(32 + b, (H, 2)))
    totals = []
    for b in range(samples.shape[0]):
        ps = paths.clone()
        ps[0] = samples[b, :, :2]
        c, _ = robot.check_rr_collisions(ps.permute(1, 0, 2))
        totals.append(int(torch.nonzero(c.int()).shape[0]))
    print("   g10: collisions in the base set", int(coll.sum()), "per-sample conflict totals", totals)
    np.savez_compressed(os.path.join(OUT, "g10_multi_agent.npz"), paths=paths.numpy(), collisions=coll.numpy(),
                        midpoints=mid.numpy(), samples=samples.numpy(), conflict_totals=np.array(totals))


def run_ref_ddim(env_id, T, B, start, goal, cons, seed_xT, use_guide=True, perturb=0.0, perturb_seed=0):
    """GaussianDiffusionModel.ddim_sample (diffusion_model_base.py:213-290) of the genuine reference with x_T injected;
    the per-step randn_like draws are multiplied by sigma = 0 (eta = 0) and are fed zeros."""
    sd = synth.synth_unet_state_dict(0)
    with quiet():
        model = make_model(sd, T)
        guide, robot, task, env = make_guide(env_id, MINS, MAXS)
    if perturb:
        gen = torch.Generator().manual_seed(perturb_seed)
        model.model.register_forward_hook(
            lambda mod, inp, out: out * (1 + perturb * torch.empty(out.shape).normal_(generator=gen)))
    costs, ws = [], []
    for (q, tr, r, soft) in cons:
        costs.append(make_cost_constraint(robot, q, tr, r, soft))
        ws.append(2e-2 if soft else 2e-1)
    guide.add_extra_costs(costs, ws)
    xT = synth.synth_noise(seed_xT, (B, H, D))
    n_pairs = T // 5 + 1
    with quiet(), injected_noise([xT] + [np.zeros((B, H, D), np.float32)] * n_pairs) as q:
        x, chain = model.ddim_sample((B, H, D), hard_conds_for(start, goal), n_diffusion_steps=T, return_chain=True,
                                     guide=guide if use_guide else None, t_start_guide=ceil(0.5 * T), n_guide_steps=20)
    guide.reset_extra_costs()
    return chain.transpose(0, 1).numpy()      # [n_pairs + 1, B, H, D]


def g11():
    """DDIM sampler: unguided on the Empty map (T=50: 10 sampling steps) and guided with constraints on
    Highways (T=100: 20 sampling steps); full chains + the reference's own sensitivity to a 1e-6 UNet perturbation."""
    starts, goals = synth.start_goal_circle(6, 0.8)
    chain = run_ref_ddim("EnvEmpty2D", 50, 8, starts[0], goals[0], [], 31, use_guide=False)
    pert = run_ref_ddim("EnvEmpty2D", 50, 8, starts[0], goals[0], [], 31, use_guide=False, perturb=1e-6, perturb_seed=1)
    np.savez_compressed(os.path.join(OUT, "g11_ddim_empty_T50.npz"), chain=chain, sens=rel_l2(pert[-1], chain[-1]),
                        meta=np.array([50, 8, 6, 0, 31]))
    print("   g11 empty: sens", rel_l2(pert[-1], chain[-1]))
    starts, goals, soft, hard = highways_case()
    cons_h = [(*soft, True), (*hard, False)]
    chain = run_ref_ddim("EnvHighways2D", 100, 8, starts[3], goals[3], cons_h, 32)
    sens = 0.0
    for ps in range(1, 4):
        pert = run_ref_ddim("EnvHighways2D", 100, 8, starts[3], goals[3], cons_h, 32, perturb=1e-6, perturb_seed=ps)
        sens = max(sens, rel_l2(pert[-1], chain[-1]))
    np.savez_compressed(os.path.join(OUT, "g11_ddim_highways_T100.npz"), chain=chain, sens=sens,
                        meta=np.array([100, 8, 10, 3, 32]))
    print("   g11 highways: sens", sens)


def g12():
    """Outer-boundary contract (what CBS / PP call on planner.robot / planner.task): RobotPlanarDisk.check_rr_collisions
    (robot_planar_disk.py:173-203) on [N,2] and [H,N,2] inputs and PlanningTask.compute_collision (tasks.py:141-234) on
    [D], [N,2] and [B,H,4] inputs of the Highways task (default margin = collision margin + cutoff), plus the
    compute_variance_waypoints metric of the g9 batch."""
    from torch_robotics.trajectory.metrics import compute_variance_waypoints
    with quiet():
        env, robot, task = make_task("EnvHighways2D")
    rng = np.random.Generator(np.random.PCG64(41))
    pts = torch.from_numpy(rng.uniform(-1.12, 1.12, size=(96, 2)).astype(np.float32))
    starts, goals = synth.start_goal_circle(10, 0.45)
    starts_t = torch.from_numpy(starts)
    close = starts_t.clone()
    close[1] = close[0] + torch.tensor([0.08, 0.0])                   # two robots closer than 2.1 r
    g9 = np.load(os.path.join(OUT, "g9_post.npz"))
    trajs = torch.from_numpy(g9["trajs"])
    paths = trajs[:6, :, :2].permute(1, 0, 2).contiguous()             # [H, 6, 2]
    paths[:, 1] = paths[:, 0] + 0.07                                  # robots 0 and 1 collide at every step
    c_pts = task.compute_collision(pts)
    c_one = task.compute_collision(pts[5])
    c_traj = task.compute_collision(trajs)
    rr_starts, mid_starts = robot.check_rr_collisions(starts_t)
    rr_close, mid_close = robot.check_rr_collisions(close)
    rr_paths, mid_paths = robot.check_rr_collisions(paths)
    out = {"points": pts.numpy(), "coll_points": c_pts.numpy(), "coll_one": c_one.numpy(), "coll_trajs": c_traj.numpy(),
           "starts": starts, "close": close.numpy(), "paths": paths.numpy(),
           "rr_starts": rr_starts.numpy(), "rr_close": rr_close.numpy(), "mid_close": mid_close.numpy(),
           "rr_paths": rr_paths.numpy(), "mid_paths": mid_paths.numpy(),
           "variance_waypoints": np.float64(compute_variance_waypoints(trajs, robot))}
    print("   g12: colliding points", int(c_pts.sum()), "of", pts.shape[0], "shapes", tuple(c_pts.shape), tuple(c_one.shape),
          tuple(c_traj.shape), "rr pairs", int(rr_close.sum()), int(rr_paths.sum()))
    np.savez_compressed(os.path.join(OUT, "g12_boundary.npz"), **out)


def g13():
    """MPDEnsemble.split_cost_constraints_to_tasks + the per-tile shift of run_constrained_inference
    (mpd_ensemble.py:431-507, 515-518), executed on the genuine class through an attribute stand-in (the method only touches
    self.task / robot / n_support_points / tensor_args; constructing an MPDEnsemble needs the dataset files).  A mixed list:
    hard + soft constraints over 3 tiles, one range starting exactly on a tile boundary (t = 64), one straddling it
    (t = 62..66 stays with tile 0: 'we do not break down long constraint intervals').  Stored per tile, in the order the
    guides receive them: qs, traj_ranges, radii (after the shift) and is_soft."""
    import types
    from mmd.planners.single_agent.mpd_ensemble import MPDEnsemble
    from mp_baselines.planners.costs.cost_functions import CostConstraint
    from torch_robotics.robots import RobotPlanarDisk
    from torch_robotics.tasks.tasks_ensemble import PlanningTaskEnsemble
    with quiet():
        robot = RobotPlanarDisk(tensor_args=TENSOR_ARGS)
    task = types.SimpleNamespace(tasks={0: "t0", 1: "t1", 2: "t2"})
    task.infer_task_id_from_q_idx = types.MethodType(PlanningTaskEnsemble.infer_task_id_from_q_idx, task)
    me = types.SimpleNamespace(task=task, robot=robot, n_support_points=H, tensor_args=TENSOR_ARGS)
    transforms = {0: torch.tensor([0.0, 0.0]), 1: torch.tensor([2.0, 0.0]), 2: torch.tensor([4.0, 0.5])}

    def cc(qs, ranges, radii, soft):
        return CostConstraint(robot, H, q_l=[torch.tensor(q, dtype=torch.float32) for q in qs], traj_range_l=ranges,
                              radius_l=radii, is_soft=soft, tensor_args=TENSOR_ARGS)
    inputs = [
        (([0.1, 0.2], [2.3, 0.1]), [(10, 14), (70, 75)], [0.12, 0.10], False),
        (([-0.4, 0.3], [1.7, -0.2], [4.4, 0.6]), [(5, 6), (64, 65), (130, 131)], [0.12, 0.12, 0.12], True),
        (([3.9, 0.4],), [(128, 140)], [0.2], False),
        (([0.9, -0.1], [0.0, 0.0]), [(62, 66), (0, 1)], [0.12, 0.15], True),
    ]
    cons = [cc(*a) for a in inputs]
    with quiet():
        split = MPDEnsemble.split_cost_constraints_to_tasks(me, cons)
    out = {"task_order": np.array(list(split.keys()), dtype=np.int64)}
    for task_id, cl in split.items():
        out[f"n_{task_id}"] = np.int64(len(cl))
        for k, c in enumerate(cl):
            c.traj_ranges -= task_id * H                                  # mpd_ensemble.py:517
            c.qs -= transforms[task_id]                                   # :518
            out[f"qs_{task_id}_{k}"] = c.qs.numpy().astype(np.float32)
            out[f"ranges_{task_id}_{k}"] = np.asarray(c.traj_ranges.numpy(), dtype=np.float32)
            out[f"radii_{task_id}_{k}"] = c.radii.numpy().astype(np.float32)
            out[f"soft_{task_id}_{k}"] = np.bool_(c.is_soft)
    print("   g13: tiles", list(split.keys()), "costs per tile", [len(v) for v in split.values()])
    np.savez_compressed(os.path.join(OUT, "g13_split_constraints.npz"), **out)


EXTRA_SPHERES = [(0.35, -0.30, 0.09), (-0.55, 0.15, 0.06)]          # (cx, cy, r)
EXTRA_BOXES = [(-0.10, 0.55, 0.30, 0.12), (0.62, 0.48, 0.10, 0.22)]    # (cx, cy, size x, size y)


def g14():
    """A map WITH extra objects (EnvBase.obj_extra_list, env_base.py:76-89; every shipped ExtraObjects map has an empty
    list): EnvHighways2D + an ObjectField of a MultiSphereField and a MultiBoxField.  Stores guide(x) with the full
    collision fields (mpd.py:220: fixed grid + extra objects in one CollisionObjectDistanceField, workspace walls, GP
    prior), guide(x) with use_guide_on_extra_objects_only (mpd.py:216-219: the extra-objects field alone + GP prior), and
    the task's occupancy (compute_collision at the guide margin and get_trajs_collision_and_free) of a batch."""
    from mp_baselines.planners.costs.cost_functions import CostCollision, CostComposite, CostGPTrajectory
    from mmd.models.diffusion_models.guides import GuideManagerTrajectoriesWithVelocity
    from ref_harness import DatasetLike
    from torch_robotics import environments
    from torch_robotics.environments.primitives import MultiBoxField, MultiSphereField, ObjectField
    from torch_robotics.robots import RobotPlanarDisk
    from torch_robotics.tasks.tasks import PlanningTask
    sph, box = np.array(EXTRA_SPHERES, np.float32), np.array(EXTRA_BOXES, np.float32)
    with quiet():
        extra = ObjectField([MultiSphereField(sph[:, :2], sph[:, 2], tensor_args=TENSOR_ARGS),
                             MultiBoxField(box[:, :2], box[:, 2:], tensor_args=TENSOR_ARGS)], "extra")
        env = environments.EnvHighways2D(obj_extra_list=[extra], tensor_args=TENSOR_ARGS)
        robot = RobotPlanarDisk(tensor_args=TENSOR_ARGS)
        task = PlanningTask(env=env, robot=robot, obstacle_cutoff_margin=0.05, tensor_args=TENSOR_ARGS)
    robot.dt = 5.0 / H

    def guide_for(fields):
        cost_l = [CostCollision(robot, H, field=f, sigma_coll=1.0, tensor_args=TENSOR_ARGS) for f in fields]
        w_l = [2e-2] * len(cost_l)
        cost_l.append(CostGPTrajectory(robot, H, robot.dt, sigma_gp=1.0, tensor_args=TENSOR_ARGS))
        w_l.append(8e-2)
        comp = CostComposite(robot, H, cost_l, weights_cost_l=w_l, tensor_args=TENSOR_ARGS)
        return GuideManagerTrajectoriesWithVelocity(DatasetLike(MINS, MAXS), comp, clip_grad=True,
                                                    interpolate_trajectories_for_collision=True,
                                                    num_interpolated_points=ceil(H * 1.5), tensor_args=TENSOR_ARGS)
    x = torch.from_numpy(synth.synth_noise(95, (8, H, D))) * 0.6
    with quiet():
        g_full = guide_for(task.get_collision_fields())(x).numpy()
        g_extra = guide_for(task.get_collision_fields_extra_objects())(x).numpy()
    xu = DatasetLike(MINS, MAXS).unnormalize_trajectories(x)
    with quiet():
        coll_pts = task.compute_collision(xu[..., :2].reshape(-1, 2))
        _, coll_idxs, _, free_idxs, _ = task.get_trajs_collision_and_free(xu, return_indices=True)
    rng = np.random.Generator(np.random.PCG64(96))
    pts = torch.from_numpy(rng.uniform(-1.0, 1.0, size=(512, 2)).astype(np.float32))
    with quiet():
        coll_rand = task.compute_collision(pts)
    print("   g14: |guide| full / extra-only", float(np.abs(g_full).max()), float(np.abs(g_extra).max()),
          "colliding points", int(coll_pts.sum()), "/", coll_pts.numel(), "random", int(coll_rand.sum()), "free trajs", free_idxs.numel())
    np.savez_compressed(os.path.join(OUT, "g14_extra_objects.npz"), spheres=sph, boxes=box, guide_full=g_full,
                        guide_extra_only=g_extra, coll_points=coll_pts.numpy(), points=pts.numpy(),
                        coll_random=coll_rand.numpy(), free_idxs=free_idxs.reshape(-1).numpy(),
                        coll_idxs=coll_idxs.reshape(-1).numpy())


N_DIST_SEEDS = int(os.environ.get("MMD_DIST_SEEDS", "32"))


def violation_counts(pos, cons):
    """pos [n,H,2] un-normalised final positions; cons: the case's (q, t_range, radius, soft) groups.  Per trajectory: number of
    (constraint point, support point) pairs inside the radius at an active time (t0 <= t < t1, cost_functions.py:305)."""
    out = np.zeros(pos.shape[0], np.int64)
    for (q, tr, r, soft) in cons:
        for k in range(len(q)):
            t0, t1 = int(tr[k][0]), int(tr[k][1])
            d = np.linalg.norm(pos[:, t0:t1].astype(np.float64) - np.asarray(q[k], np.float64)[None, None], axis=-1)
            out += (d < float(r[k])).sum(1)
    return out


def g15():
    """Distribution-level parity: the guided sampler is chaotic end to end, so beside the per-step teacher-forced
    statement the reference's OUTPUT DISTRIBUTION is pinned: for the 32-robot north-star shape (empty32_T25) and the Highways
    case with soft + hard constraints (highways_T100), N_DIST_SEEDS independent noise seeds (x_T and every step's noise from
    synth.synth_noise(base + 2 j), (base + 2 j + 1)) through the genuine reference; stored: the final normalised rows of every
    sample, per-support-point mean / covariance of the un-normalised positions over all samples, the reference's own free /
    collision split (PlanningTask.get_trajs_collision_and_free, tasks.py:236-311) and the soft / hard constraint violation
    counts per trajectory (guides.py:180-226 is what keeps them low)."""
    cases = {}
    starts, goals = synth.start_goal_circle(32, 0.8)
    paths = synth.straight_line_paths(starts, goals, H)
    q, tr, r = soft_points(paths, 5)
    cases["empty32_T25"] = ("EnvEmpty2D", 25, 4, starts[5], goals[5], [(q, tr, r, True)], 3000)
    starts, goals, soft, hard = highways_case()
    cases["highways_T100"] = ("EnvHighways2D", 100, 8, starts[3], goals[3], [(*soft, True), (*hard, False)], 4000)
    for name, (env_id, T, B, start, goal, cons, base) in cases.items():
        finals = []
        for j in range(N_DIST_SEEDS):
            chain = run_ref_inference(env_id, T, B, start, goal, cons, base + 2 * j, base + 2 * j + 1)
            finals.append(chain[-1])
            print("   g15", name, "seed", j, flush=True)
        finals = np.concatenate(finals, 0).astype(np.float32)                      # [N * B, H, D] normalised
        with quiet():
            guide, robot, task, env = make_guide(env_id, MINS, MAXS)
        trajs = guide.dataset.unnormalize_trajectories(torch.from_numpy(finals))
        coll, coll_idxs, free, free_idxs, wp = task.get_trajs_collision_and_free(trajs, return_indices=True)
        pos = trajs[..., :2].numpy().astype(np.float64)
        mean = pos.mean(0)                                                         # [H, 2]
        dev = pos - mean[None]
        cov = np.einsum("nhi,nhj->hij", dev, dev) / (pos.shape[0] - 1)             # [H, 2, 2]
        free_mask = np.zeros(finals.shape[0], bool)
        free_mask[free_idxs.numpy().reshape(-1)] = True
        viol = violation_counts(trajs[..., :2].numpy(), cons)
        np.savez_compressed(os.path.join(OUT, f"g15_distribution_{name}.npz"), finals=finals, pos_mean=mean, pos_cov=cov,
                            free_mask=free_mask, violations=viol, meta=np.array([T, B, N_DIST_SEEDS, base]))
        print("   g15", name, "free", int(free_mask.sum()), "of", finals.shape[0], "violating", int((viol > 0).sum()),
              "pairs", int(viol.sum()), flush=True)


def g16():
    """Reference options MPD / MPDEnsemble never set (missing #5): GuideManager.clip_gradient by value and switched off
    (guides.py:228-259), guide_gradient_steps' scale_grad_by_std (sample_functions.py:100-101) and GaussianDiffusionModel(
    predict_epsilon=False) (diffusion_model_base.py:131-141).  Inputs as g5 / g6: Highways guide inputs (seed 7), chains with
    injected noise (T = 25, B = 4)."""
    from ref_harness import GaussianDiffusionModel, TemporalUnet
    out = {}
    x = torch.from_numpy(synth.synth_noise(7, (8, H, D))) * 0.6
    starts, goals, soft, hard = highways_case()
    for key, kw in (("guide_clip_value", dict(clip_grad=True, clip_grad_rule="value")), ("guide_clip_off", dict(clip_grad=False))):
        with quiet():
            guide, robot, task, env = make_guide("EnvHighways2D", MINS, MAXS)
        guide.clip_grad, guide.clip_grad_rule = kw["clip_grad"], kw.get("clip_grad_rule", "norm")
        guide.add_extra_costs([make_cost_constraint(robot, *soft, True), make_cost_constraint(robot, *hard, False)], [2e-2, 2e-1])
        out[key] = guide(x).numpy()
        out["max_grad_value"] = np.float32(guide.max_grad_value)
    # scale_grad_by_std through run_inference's **diffusion_kwargs -> sample_fn
    T, B = 25, 4
    sd = synth.synth_unet_state_dict(0)
    xT = synth.synth_noise(41, (B, H, D))
    steps = synth.synth_noise(42, (T + 1, B, H, D))
    with quiet():
        model = make_model(sd, T)
        guide, robot, task, env = make_guide("EnvHighways2D", MINS, MAXS)
    guide.add_extra_costs([make_cost_constraint(robot, *soft, True), make_cost_constraint(robot, *hard, False)], [2e-2, 2e-1])
    with quiet(), injected_noise([xT] + list(steps)) as q:
        chain = model.run_inference(None, hard_conds_for(starts[3], goals[3]), n_samples=B, horizon=H, return_chain=True,
                                    sample_fn=ddpm_sample_fn, guide=guide, n_guide_steps=20, t_start_guide=ceil(0.5 * T),
                                    noise_std_extra_schedule_fn=lambda x: 0.5, n_diffusion_steps_without_noise=1,
                                    scale_grad_by_std=True)
        assert len(q) == 0
    out["chain_scale_grad_by_std"] = chain.numpy()
    # predict_epsilon = False: the network output is x_recon (no guide: the prior chain)
    unet = TemporalUnet(state_dim=4, n_support_points=64, unet_input_dim=32, dim_mults=(1, 2, 4))
    m0 = GaussianDiffusionModel(model=unet, variance_schedule="exponential", n_diffusion_steps=T, predict_epsilon=False)
    m0.load_state_dict({"model." + k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    m0.eval()
    with quiet(), injected_noise([xT] + list(steps)) as q:
        chain0 = m0.run_inference(None, hard_conds_for(starts[3], goals[3]), n_samples=B, horizon=H, return_chain=True,
                                  sample_fn=ddpm_sample_fn, guide=None, noise_std_extra_schedule_fn=lambda x: 0.5,
                                  n_diffusion_steps_without_noise=1)
        assert len(q) == 0
    out["chain_predict_x0"] = chain0.numpy()
    out["meta"] = np.array([T, B, 41, 42])
    np.savez_compressed(os.path.join(OUT, "g16_options.npz"), **out)
    print("   g16:", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim > 0})


def g17():
    """UNET_DIM_MULTS[1] = (1, 2, 4, 8) (mmd/models/__init__.py:8-11, temporal_unet.py:50-119): eps of a four-level TemporalUnet
    at t in {0, 37, 99} (inputs as g2), of a two-level one (1, 2) with unet_input_dim = 16, and a guided Highways chain (T = 25,
    B = 4, inputs as g16) through the four-level network with its sensitivity row (1e-6 relative perturbations, 8 draws)."""
    x = torch.from_numpy(synth.synth_noise(5, (4, H, D)))
    out = {"x_seed": 5, "ts": np.array([0, 37, 99])}
    for tag, uid, dm in (("d32_1248", 32, (1, 2, 4, 8)), ("d16_12", 16, (1, 2)), ("d8_1", 8, (1,)), ("d64_124", 64, (1, 2, 4))):
        sd = synth.synth_unet_state_dict(0, unet_input_dim=uid, dim_mults=dm)
        with quiet():
            m = make_model(sd, 100, dim_mults=dm, unet_input_dim=uid)
        for t in (0, 37, 99):
            with torch.no_grad():
                out[f"{tag}.eps_t{t}"] = m.model(x, torch.full((4,), t, dtype=torch.long), None).numpy()
    T, B = 25, 4
    dm = (1, 2, 4, 8)
    sd = synth.synth_unet_state_dict(0, dim_mults=dm)
    starts, goals, soft, hard = highways_case()
    xT = synth.synth_noise(41, (B, H, D))
    steps = synth.synth_noise(42, (T + 1, B, H, D))

    def run(perturb=0.0, seed=0):
        with quiet():
            model = make_model(sd, T, dim_mults=dm)
            guide, robot, task, env = make_guide("EnvHighways2D", MINS, MAXS)
        if perturb:
            gen = torch.Generator().manual_seed(seed)
            model.model.register_forward_hook(
                lambda mod, inp, o: o * (1 + perturb * torch.empty(o.shape).normal_(generator=gen)))
        guide.add_extra_costs([make_cost_constraint(robot, *soft, True), make_cost_constraint(robot, *hard, False)], [2e-2, 2e-1])
        with quiet(), injected_noise([xT] + list(steps)) as q:
            chain = model.run_inference(None, hard_conds_for(starts[3], goals[3]), n_samples=B, horizon=H, return_chain=True,
                                        sample_fn=ddpm_sample_fn, guide=guide, n_guide_steps=20, t_start_guide=ceil(0.5 * T),
                                        noise_std_extra_schedule_fn=lambda x: 0.5, n_diffusion_steps_without_noise=1)
            assert len(q) == 0
        return chain.numpy()
    chain = run()
    sens = np.zeros(chain.shape[0])
    for ps in range(1, 9):
        pert = run(1e-6, ps)
        sens = np.maximum(sens, [rel_l2(pert[r], chain[r]) for r in range(chain.shape[0])])
    out["chain"], out["sens"], out["meta"] = chain, sens, np.array([T, B, 41, 42])
    np.savez_compressed(os.path.join(OUT, "g17_unet_dim_mults.npz"), **out)
    print("   g17: chain", chain.shape, "final-row sensitivity", f"{sens[-1]:.2e}")


def g18():
    """Hard conditions on rows other than 0 / H-1 (apply_hard_conditioning takes any {row: state} dict, sample_functions.py:8-14)
    and DDIM with GaussianDiffusionModel(predict_epsilon=False) (predict_noise_from_start, diffusion_model_base.py:114-124, as
    ddim_sample uses it at :248).  Highways, agent 3 of the g5 / g6 case, a via state pinned at rows 17 and 40 besides start / goal:
    a guided DDPM chain (T = 25, B = 4, injected noise) with its sensitivity row, a guided DDIM chain (T = 100, B = 4) with the
    same four hard rows, and an unguided DDIM chain of an x0-predicting model (T = 50, B = 4, start / goal only)."""
    from ref_harness import GaussianDiffusionModel, TemporalUnet
    starts, goals, soft, hard = highways_case()
    hc = hard_conds_for(starts[3], goals[3])
    via = {17: np.array([-0.15, 0.3, 0.2, -0.1], np.float32), 40: np.array([0.2, -0.25, 0.0, 0.3], np.float32)}
    for row, v in via.items():
        hc[row] = torch.from_numpy(normalize(v).astype(np.float32))
    out = {"via_rows": np.array(sorted(via)), "via_states": np.stack([via[r] for r in sorted(via)])}
    sd = synth.synth_unet_state_dict(0)
    T, B = 25, 4
    xT = synth.synth_noise(43, (B, H, D))
    steps = synth.synth_noise(44, (T + 1, B, H, D))

    def run(perturb=0.0, seed=0):
        with quiet():
            model = make_model(sd, T)
            guide, robot, task, env = make_guide("EnvHighways2D", MINS, MAXS)
        if perturb:
            gen = torch.Generator().manual_seed(seed)
            model.model.register_forward_hook(
                lambda mod, inp, o: o * (1 + perturb * torch.empty(o.shape).normal_(generator=gen)))
        guide.add_extra_costs([make_cost_constraint(robot, *soft, True), make_cost_constraint(robot, *hard, False)], [2e-2, 2e-1])
        with quiet(), injected_noise([xT] + list(steps)) as q:
            chain = model.run_inference(None, hc, n_samples=B, horizon=H, return_chain=True, sample_fn=ddpm_sample_fn, guide=guide,
                                        n_guide_steps=20, t_start_guide=ceil(0.5 * T), noise_std_extra_schedule_fn=lambda x: 0.5,
                                        n_diffusion_steps_without_noise=1)
            assert len(q) == 0
        return chain.numpy()
    chain = run()
    sens = np.zeros(chain.shape[0])
    for ps in range(1, 9):
        pert = run(1e-6, ps)
        sens = np.maximum(sens, [rel_l2(pert[r], chain[r]) for r in range(chain.shape[0])])
    out["ddpm_chain"], out["ddpm_sens"], out["ddpm_meta"] = chain, sens, np.array([T, B, 43, 44])
    # DDIM, four hard rows, guided
    T2 = 100
    with quiet():
        model = make_model(sd, T2)
        guide, robot, task, env = make_guide("EnvHighways2D", MINS, MAXS)
    guide.add_extra_costs([make_cost_constraint(robot, *soft, True), make_cost_constraint(robot, *hard, False)], [2e-2, 2e-1])
    xT2 = synth.synth_noise(45, (B, H, D))
    with quiet(), injected_noise([xT2] + [np.zeros((B, H, D), np.float32)] * (T2 // 5 + 1)):
        x, ch = model.ddim_sample((B, H, D), hc, n_diffusion_steps=T2, return_chain=True, guide=guide, t_start_guide=ceil(0.5 * T2),
                                  n_guide_steps=20)
    out["ddim_chain"], out["ddim_meta"] = ch.transpose(0, 1).numpy(), np.array([T2, B, 45])
    # DDIM with an x0-predicting model, unguided
    T3 = 50
    unet = TemporalUnet(state_dim=4, n_support_points=64, unet_input_dim=32, dim_mults=(1, 2, 4))
    m0 = GaussianDiffusionModel(model=unet, variance_schedule="exponential", n_diffusion_steps=T3, predict_epsilon=False)
    m0.load_state_dict({"model." + k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    m0.eval()
    xT3 = synth.synth_noise(46, (B, H, D))
    with quiet(), injected_noise([xT3] + [np.zeros((B, H, D), np.float32)] * (T3 // 5 + 1)):
        x, ch0 = m0.ddim_sample((B, H, D), hard_conds_for(starts[3], goals[3]), n_diffusion_steps=T3, return_chain=True)
    out["ddim_x0_chain"], out["ddim_x0_meta"] = ch0.transpose(0, 1).numpy(), np.array([T3, B, 46])
    np.savez_compressed(os.path.join(OUT, "g18_hard_rows_ddim_x0.npz"), **out)
    print("   g18:", {k: v.shape for k, v in out.items()}, "ddpm final-row sens", f"{sens[-1]:.2e}")


def synth_training_trajectories(n, seed):
    """Synthetic collision-free demonstrations on the EMPTY map (there is no dataset offline): smooth-step motion from a
    random start to a random goal plus two low-frequency sine modes that vanish at both ends, velocities by central differences
    at dt = trajectory_duration / H (mpd.py:140), zero at the ends (the planners' hard conditions pin zero velocity,
    trajectories.py:216-239).  Un-normalised [n, H, 4]; bit-reproducible (numpy PCG64)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    u = np.linspace(0.0, 1.0, H)
    sm = 3 * u ** 2 - 2 * u ** 3
    start = rng.uniform(-0.85, 0.85, size=(n, 2))
    goal = rng.uniform(-0.85, 0.85, size=(n, 2))
    pos = start[:, None, :] + (goal - start)[:, None, :] * sm[None, :, None]
    for k in (1, 2):
        a = rng.normal(0.0, 0.12 / k, size=(n, 1, 2))
        pos = pos + a * np.sin(k * np.pi * u)[None, :, None]
    pos = np.clip(pos, -0.95, 0.95)
    vel = np.gradient(pos, 5.0 / H, axis=1)
    vel[:, 0] = 0.0
    vel[:, -1] = 0.0
    return np.concatenate([pos, np.clip(vel, -1.5, 1.5)], axis=-1).astype(np.float32)


def train_reference_unet(T=25, n_steps=2500, batch=64, lr=3e-4, n_data=4096):
    """The reference's own training objective on the reference's own modules: GaussianDiffusionModel.loss -> p_losses
    (diffusion_model_base.py:435-461: q_sample at a random t, hard conditioning, eps-prediction, l2) as
    GaussianDiffusionLoss.loss_fn calls it (mmd/losses/gaussian_diffusion_loss.py:9-28), Adam as in mmd/trainer/trainer.py:119, from
    the synthetic init (seed 0), fixed seeds.  No EMA (the trainer starts it at step 1000): a BRIEFLY trained network, enough to
    denoise.  Returns the UNet state dict as {key: float32 ndarray} in unet_param_spec order."""
    sd0 = synth.synth_unet_state_dict(0)
    with quiet():
        model = make_model(sd0, T)
    for p in model.parameters():
        p.requires_grad_(True)
    model.train()
    data = torch.from_numpy(normalize(synth_training_trajectories(n_data, 190)).astype(np.float32))
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    gen = np.random.Generator(np.random.PCG64(191))
    torch.manual_seed(192)
    for step in range(n_steps):
        idx = torch.from_numpy(gen.integers(0, n_data, size=batch))
        x = data[idx]
        hard = {0: x[:, 0].clone(), H - 1: x[:, -1].clone()}
        loss, _ = model.loss(x, None, hard)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if step % 250 == 0 or step == n_steps - 1:
            print(f"   g19 train step {step}: loss {float(loss.detach()):.4f}", flush=True)
    model.eval()
    out = {k[len("model."):]: v.detach().numpy().astype(np.float32).copy() for k, v in model.state_dict().items() if k.startswith("model.")}
    assert list(out.keys()) == list(sd0.keys())
    return out, float(loss.detach())


def g19():
    """A network that DENOISES: the reference's TemporalUnet trained for a fixed number of Adam steps with the
    reference's loss on synthetic collision-free trajectories, then the guided chains of the two constraint cases whose random-init
    versions are chaotic end to end -- the 32-robot north-star shape (Empty, robot 5, 31 x 63 soft points) and Highways with soft +
    hard constraints -- at T = 25 (the released checkpoints' step count): the full chain of noise seed 0 with its `sens` rows
    (the reference against itself under relative 1e-6 perturbations of the UNet output, N_SENS_DRAWS draws), and the final rows of
    8 noise seeds for the matched-within-1e-3 fraction.  Data only: the state dict, the chains."""
    T = 25
    sd, final_loss = train_reference_unet(T)
    np.savez_compressed(os.path.join(OUT, "g19_trained_unet.npz"), final_loss=np.float32(final_loss), **sd)
    # how well it denoises: eps-prediction error on held-out synthetic trajectories at t = 5, 12, 20 (a random-init net: ~1)
    with quiet():
        model = make_model(sd, T)
    held = torch.from_numpy(normalize(synth_training_trajectories(256, 193)).astype(np.float32))
    out = {"T": np.array(T), "final_loss": np.float32(final_loss)}
    mse = []
    for t in (5, 12, 20):
        z = torch.from_numpy(synth.synth_noise(194 + t, tuple(held.shape)))
        tt = torch.full((held.shape[0],), t, dtype=torch.long)
        with torch.no_grad():
            xn = model.q_sample(held, tt, z)
            mse.append(float(((model.model(xn, tt, None) - z) ** 2).mean()))
    out["heldout_eps_mse_t5_t12_t20"] = np.array(mse, np.float32)
    print("   g19 held-out eps MSE at t = 5 / 12 / 20:", mse, flush=True)
    n_seeds = 8
    starts32, goals32 = synth.start_goal_circle(32, 0.8)
    q, tr, r = soft_points(synth.straight_line_paths(starts32, goals32, H), 5)
    hs, hg, soft, hard = highways_case()
    cases = {"empty32": ("EnvEmpty2D", 4, starts32[5], goals32[5], [(q, tr, r, True)], 400),
             "highways": ("EnvHighways2D", 8, hs[3], hg[3], [(*soft, True), (*hard, False)], 440)}
    for name, (env_id, B, st, go, cons, seed0) in cases.items():
        chain = run_ref_inference(env_id, T, B, st, go, cons, seed0, seed0 + 1, sd=sd)
        sens = np.zeros(chain.shape[0])
        for ps in range(1, N_SENS_DRAWS + 1):
            pert = run_ref_inference(env_id, T, B, st, go, cons, seed0, seed0 + 1, sd=sd, perturb=1e-6, perturb_seed=ps)
            sens = np.maximum(sens, [rel_l2(pert[k], chain[k]) for k in range(chain.shape[0])])
        finals = [chain[-1]]
        for sdx in range(1, n_seeds):
            finals.append(run_ref_inference(env_id, T, B, st, go, cons, seed0 + 2 * sdx, seed0 + 2 * sdx + 1, sd=sd)[-1])
        out[f"{name}.chain"], out[f"{name}.sens"] = chain, sens
        out[f"{name}.finals"] = np.stack(finals)                       # [n_seeds, B, H, D]; noise seeds seed0 + 2 s, + 1
        out[f"{name}.meta"] = np.array([T, B, seed0, n_seeds])
        print(f"   g19 {name}: final-row sens {sens[-1]:.2e}, max over rows {sens.max():.2e}", flush=True)
    np.savez_compressed(os.path.join(OUT, "g19_trained_chains.npz"), **out)


def _state_dict_named(w):
    """tile weights of synth.ensemble3_case: a synth seed or "g19" (the trained state dict stored by g19 -- data)."""
    if w == "g19":
        from collections import OrderedDict
        from mmd_amd.unet_spec import unet_param_spec
        g = np.load(os.path.join(OUT, "g19_trained_unet.npz"))
        return OrderedDict((k, np.ascontiguousarray(g[k], dtype=np.float32)) for k in unet_param_spec())
    return synth.synth_unet_state_dict(w)


def g20():
    """3-tile, corner-turning, heterogeneous MPDEnsemble instance (synth.ensemble3_case) through the GENUINE classes:
      * PlanningTaskEnsemble over three genuine PlanningTasks (tasks_ensemble.py:39-49) with the tile transforms of
        inference_multi_agent.py:148-151;
      * MPDEnsemble.split_cost_constraints_to_tasks (mpd_ensemble.py:431-507) on one hard + one soft CostConstraint built as
        MPDEnsemble.__call__ builds them (:360-372), then the shift of run_constrained_inference (:515-522) into the tile guides;
      * DiffusionsEnsemble.run_inference (diffusion_ensemble.py:223-263: the einops repeat of the hard conditions, p_sample_loop
        with apply_cross_conditioning after every tile step, the chain rearrange) with injected noise;
      * the post-processing of MPDEnsemble.__call__ (:385-429): per tile get_traj_unnormalized + get_stats, combine_trajs
        (tasks_ensemble.py:162-225: a sample is free iff it is free in EVERY tile's own map), smooth_trajs.
    Stored per direction ("fwd" / "rev"): every chain row of every tile + `sens` (as g8), the per-tile constraint split after the
    shift, the combined PlannerOutput fields.  The MPDEnsemble object itself cannot be constructed offline (its __init__ reads the
    dataset / checkpoint files), so the three methods run on an attribute stand-in that carries exactly what they touch."""
    import types
    from mmd.common.trajectory_utils import smooth_trajs
    from mmd.models.diffusion_models.diffusion_ensemble import DiffusionsEnsemble
    from mmd.planners.single_agent.mpd_ensemble import MPDEnsemble
    from torch_robotics.tasks.tasks_ensemble import PlanningTaskEnsemble
    import ref_harness
    T, B = 25, 8
    out = {}
    for direction in ("fwd", "rev"):
        case = synth.ensemble3_case(direction)
        K = len(case["env_ids"])
        transforms = {j: torch.from_numpy(case["transforms"][j]) for j in range(K)}
        with quiet():
            models = {j: make_model(_state_dict_named(case["weights"][j]), T) for j in range(K)}
            guides, tasks, datasets = {}, {}, []
            for j in range(K):
                # fresh (uncached) env / task per tile: EnvEnsemble shifts the tile objects' positions in place (env_ensemble.py:42-45)
                ref_harness._ENV_CACHE.pop((case["env_ids"][j], 0.01), None)
                guides[j], robot_j, tasks[j], _ = make_guide(case["env_ids"][j], MINS, MAXS, cutoff_margin=0.01)   # mpd_ensemble.py:139
                ref_harness._ENV_CACHE.pop((case["env_ids"][j], 0.01), None)
                datasets.append(guides[j].dataset)
                if j == 0:
                    robot = robot_j
            task_ens = PlanningTaskEnsemble(tasks, transforms, tensor_args=TENSOR_ARGS)
        me = types.SimpleNamespace(task=task_ens, robot=robot, n_support_points=H, tensor_args=TENSOR_ARGS)
        # mpd_ensemble.py:286-296
        start = torch.from_numpy(case["start"])
        goal = torch.from_numpy(case["goal"])
        start_local = task_ens.inverse_transform_q(0, start)
        goal_local = task_ens.inverse_transform_q(K - 1, goal)
        hard_conds = {0: {0: hard_conds_for(start_local.numpy(), start_local.numpy())[0]}}
        hard_conds.setdefault(K - 1, {})[-1] = hard_conds_for(goal_local.numpy(), goal_local.numpy())[0]
        cross_conds = {(i, i + 1): (H - 1, 0) for i in range(K - 1)}                      # :301-303
        cost_constraints = [make_cost_constraint(robot, q, tr, r, soft) for (q, tr, r, soft) in case["constraints"]]
        with quiet():
            split = MPDEnsemble.split_cost_constraints_to_tasks(me, cost_constraints)
        out[f"{direction}.task_order"] = np.array(list(split.keys()), dtype=np.int64)
        for task_id, cl in split.items():
            out[f"{direction}.n_{task_id}"] = np.int64(len(cl))
            for k, c in enumerate(cl):
                c.traj_ranges -= task_id * H                                              # :517
                c.qs -= transforms[task_id]                                               # :518
                guides[task_id].add_extra_costs([c], [2e-1 if not c.is_soft else 2e-2])   # :519-522, mmd_params.py:42-43
                out[f"{direction}.qs_{task_id}_{k}"] = c.qs.numpy().astype(np.float32)
                out[f"{direction}.ranges_{task_id}_{k}"] = np.asarray(c.traj_ranges.numpy(), dtype=np.float32)
                out[f"{direction}.radii_{task_id}_{k}"] = c.radii.numpy().astype(np.float32)
                out[f"{direction}.soft_{task_id}_{k}"] = np.bool_(c.is_soft)
        x0 = [synth.synth_noise(case["seeds"]["x0"][j], (B, H, D)) for j in range(K)]
        steps = synth.synth_noise(case["seeds"]["steps"], (T + 1, K, B, H, D))
        draws = list(x0) + [steps[k, j] for k in range(T + 1) for j in range(K)]
        sample_kwargs = [dict(guide=guides[j], n_guide_steps=20, t_start_guide=ceil(0.5 * T),
                              noise_std_extra_schedule_fn=lambda x: 0.5) for j in range(K)]     # :236-243 (a list, indexed by tile)
        ens = DiffusionsEnsemble(models, transforms)

        def run(perturb=0.0, ps=0):
            handles = []
            if perturb:
                gen = torch.Generator().manual_seed(ps)
                for j in range(K):
                    handles.append(models[j].model.register_forward_hook(
                        lambda mod, inp, o: o * (1 + perturb * torch.empty(o.shape).normal_(generator=gen))))
            with quiet(), injected_noise(draws) as q:
                chains = ens.run_inference(None, hard_conds, cross_conds=cross_conds, n_samples=B, return_chain=True,
                                           sample_fn=ddpm_sample_fn, sample_kwargs=sample_kwargs,
                                           n_diffusion_steps_without_noise=1)                  # :527-535
                assert len(q) == 0
            for h in handles:
                h.remove()
            return {j: chains[j].clone() for j in range(K)}                                     # [T+2, B, H, D]

        chains = run()
        sens = {j: np.zeros(T + 2) for j in range(K)}
        for ps in range(1, N_SENS_DRAWS + 1):
            pert = run(1e-6, ps)
            for j in range(K):
                sens[j] = np.maximum(sens[j], [rel_l2(pert[j][r].numpy(), chains[j][r].numpy()) for r in range(T + 2)])
        # ---- MPDEnsemble.__call__ :385-429 ----
        with quiet():
            results_ensemble = {}
            for j in range(K):
                r6 = task_ens.get_traj_unnormalized(j, datasets, chains[j])
                results_ensemble[j] = task_ens.get_stats(j, *r6, 0.0, save_data=False)
                out[f"{direction}.tile{j}_coll_idxs"] = np.asarray(r6[3].numpy(), dtype=np.int64).reshape(-1)
            res = task_ens.combine_trajs(results_ensemble)
            smoothed = smooth_trajs(res["trajs_iters"][-1])
        for j in range(K):
            out[f"{direction}.chain{j}"] = chains[j].numpy()
            out[f"{direction}.sens{j}"] = sens[j]
            # how many (sample, constraint point, step) triples are inside a constraint radius on the final row (tile frame)
            pos = datasets[j].unnormalize_trajectories(chains[j][-1])[..., :2]
            act = 0
            for cst in split.get(j, []):
                for q, tr, rad in zip(cst.qs, cst.traj_ranges, cst.radii):
                    seg = pos[:, int(tr[0]):int(tr[1]) + 1]
                    act += int((torch.linalg.norm(seg - q, dim=-1) < rad).sum())
            print(f"   g20 {direction} tile {j} ({case['env_ids'][j]}): final-row sens {sens[j][-1]:.2e}, max {sens[j].max():.2e}, "
                  f"constraint hits on the final row {act}, tile coll idxs {out[f'{direction}.tile{j}_coll_idxs'].tolist()}", flush=True)
            guides[j].reset_extra_costs()
        out[f"{direction}.trajs_final_global"] = res["trajs_iters"][-1].numpy()
        out[f"{direction}.trajs_iters_mid_global"] = res["trajs_iters"][T // 2 + 1].numpy()
        out[f"{direction}.smoothed"] = smoothed.numpy()
        out[f"{direction}.free_idxs"] = res["trajs_final_free_idxs"].numpy().astype(np.int64).reshape(-1)
        out[f"{direction}.coll_idxs"] = res["trajs_final_coll_idxs"].numpy().astype(np.int64).reshape(-1)
        out[f"{direction}.fraction_free"] = np.float64(res["fraction_free_trajs"])
        if len(out[f"{direction}.free_idxs"]):
            out[f"{direction}.idx_best_traj"] = np.int64(int(res["idx_best_traj"]))
            out[f"{direction}.cost_smoothness"] = res["cost_smoothness_trajs_final_free"].numpy()
            out[f"{direction}.cost_path_length"] = res["cost_path_length_trajs_final_free"].numpy()
            out[f"{direction}.cost_best_free_traj"] = np.float32(float(res["cost_best_free_traj"]))
            out[f"{direction}.variance_waypoint"] = np.float32(float(res["variance_waypoint_trajs_final_free"]))
        out[f"{direction}.meta"] = np.array([T, B, K])
        print(f"   g20 {direction}: free {out[f'{direction}.free_idxs'].tolist()} coll {out[f'{direction}.coll_idxs'].tolist()}", flush=True)
    np.savez_compressed(os.path.join(OUT, "g20_ensemble3.npz"), **out)


def g21():
    """Outer-boundary contract of the multi-tile task (what CBS / PP read from an MPDEnsemble planner's `.task`,
    cbs.py:149-156, multi_agent_utils.py:47,84,89): the GENUINE PlanningTaskEnsemble.compute_collision
    (tasks_ensemble.py:227-269: tile inferred from the global position, tile-frame occupancy against that tile's own map, points
    outside every tile stay "in collision") on [n, 2] stacked global positions and on a single [2] position, and
    infer_task_id_from_q (:340-365), over the three tiles of synth.ensemble3_case("fwd")."""
    from torch_robotics.tasks.tasks_ensemble import PlanningTaskEnsemble
    import ref_harness
    case = synth.ensemble3_case("fwd")
    K = len(case["env_ids"])
    transforms = {j: torch.from_numpy(case["transforms"][j]) for j in range(K)}
    with quiet():
        tasks = {}
        for j in range(K):
            ref_harness._ENV_CACHE.pop((case["env_ids"][j], 0.01), None)
            tasks[j] = make_task(case["env_ids"][j], 0.01)[2]
            ref_harness._ENV_CACHE.pop((case["env_ids"][j], 0.01), None)
        task_ens = PlanningTaskEnsemble(tasks, transforms, tensor_args=TENSOR_ARGS)
    rng = np.random.Generator(np.random.PCG64(220))
    local = rng.uniform(-1.0, 1.0, size=(K, 40, 2)).astype(np.float32)
    pts = np.concatenate([local[j] + case["transforms"][j] for j in range(K)] +
                         [np.array([[-1.5, 0.0], [0.0, -2.0], [3.5, -2.0], [2.0, 1.2], [1.0, 0.3], [2.4, -1.0], [1.0, -1.0]], np.float32)])
    pts_t = torch.from_numpy(pts)
    out = {"points": pts, "task_ids": task_ens.infer_task_id_from_q(pts_t.unsqueeze(1)).numpy()}
    with quiet():
        out["collision"] = task_ens.compute_collision(pts_t).numpy()
        single = []
        for k in (0, 45, 90, 120, 121, 124):
            single.append(np.asarray(task_ens.compute_collision(pts_t[k]).numpy()).reshape(-1))
        out["single_idx"] = np.array([0, 45, 90, 120, 121, 124])
        out["single_collision"] = np.concatenate(single)
    print("   g21: task ids", np.unique(out["task_ids"], return_counts=True), "colliding", int(out["collision"].sum()), "of", len(pts),
          "shape", out["collision"].shape, out["collision"].dtype, "single", out["single_collision"].tolist())
    np.savez_compressed(os.path.join(OUT, "g21_ensemble_task.npz"), **out)


def g22():
    """DiffusionsEnsemble.run_local_inference on the 3-tile corner-turning instance (diffusion_ensemble.py:265-313: the re-plan path of
    MPDEnsemble with an experience, mpd_ensemble.py:571-600): the seed batch [B, K*64, D] in the GLOBAL frame is forward-noised as a whole
    by models[0].q_sample (ONE draw of the full shape), split per tile and moved to the tile frames (p_sample_loop :67-72, x and y tile
    offsets), hard / cross conditioned, then denoised for 3 + 1 steps with the tile guides and the routed constraints of g20.  Both
    directions; injected noise; every chain row of every tile + `sens`."""
    import types
    from mmd.models.diffusion_models.diffusion_ensemble import DiffusionsEnsemble
    from mmd.planners.single_agent.mpd_ensemble import MPDEnsemble
    from torch_robotics.tasks.tasks_ensemble import PlanningTaskEnsemble
    import ref_harness
    T, B, n_noise, n_denoise = 25, 4, 3, 3
    out = {}
    for direction in ("fwd", "rev"):
        case = synth.ensemble3_case(direction)
        K = len(case["env_ids"])
        transforms = {j: torch.from_numpy(case["transforms"][j]) for j in range(K)}
        with quiet():
            models = {j: make_model(_state_dict_named("g19"), T) for j in range(K)}
            guides, tasks = {}, {}
            for j in range(K):
                ref_harness._ENV_CACHE.pop((case["env_ids"][j], 0.01), None)
                guides[j], robot_j, tasks[j], _ = make_guide(case["env_ids"][j], MINS, MAXS, cutoff_margin=0.01)
                ref_harness._ENV_CACHE.pop((case["env_ids"][j], 0.01), None)
                if j == 0:
                    robot = robot_j
            task_ens = PlanningTaskEnsemble(tasks, transforms, tensor_args=TENSOR_ARGS)
        me = types.SimpleNamespace(task=task_ens, robot=robot, n_support_points=H, tensor_args=TENSOR_ARGS)
        start, goal = torch.from_numpy(case["start"]), torch.from_numpy(case["goal"])
        start_local, goal_local = task_ens.inverse_transform_q(0, start), task_ens.inverse_transform_q(K - 1, goal)
        hard_conds = {0: {0: hard_conds_for(start_local.numpy(), start_local.numpy())[0]}}
        hard_conds.setdefault(K - 1, {})[-1] = hard_conds_for(goal_local.numpy(), goal_local.numpy())[0]
        cross_conds = {(i, i + 1): (H - 1, 0) for i in range(K - 1)}
        with quiet():
            split = MPDEnsemble.split_cost_constraints_to_tasks(me, [make_cost_constraint(robot, q, tr, r, soft) for (q, tr, r, soft) in case["constraints"]])
        for task_id, cl in split.items():
            for c in cl:
                c.traj_ranges -= task_id * H
                c.qs -= transforms[task_id]
                guides[task_id].add_extra_costs([c], [2e-1 if not c.is_soft else 2e-2])
        # the seed: a piecewise-linear GLOBAL path start -> tile centres' joints -> goal with zero velocity, + small noise (the caller
        # passes the previous call's un-normalised, smoothed trajs_final: cbs.py:424 -> mpd_ensemble.py:571)
        way = [case["start"]] + [0.5 * (case["transforms"][j] + case["transforms"][j + 1]) for j in range(K - 1)] + [case["goal"]]
        segs = []
        for j in range(K):
            a = np.linspace(0.0, 1.0, H, dtype=np.float32)[:, None]
            segs.append(way[j][None] * (1 - a) + way[j + 1][None] * a)
        pos = np.concatenate(segs, 0)                                                   # [K*64, 2]
        seed = np.concatenate([np.repeat(pos[None], B, 0), np.zeros((B, K * H, 2), np.float32)], -1)
        seed = (seed + 0.02 * synth.synth_noise(case["seeds"]["steps"] + 20, (B, K * H, D))).astype(np.float32)
        qn = synth.synth_noise(case["seeds"]["steps"] + 21, (B, K * H, D))
        steps = synth.synth_noise(case["seeds"]["steps"] + 22, (n_denoise + 1, K, B, H, D))
        draws = [qn] + [steps[k, j] for k in range(n_denoise + 1) for j in range(K)]
        sample_kwargs = [dict(guide=guides[j], n_guide_steps=20, t_start_guide=ceil(0.5 * T), noise_std_extra_schedule_fn=lambda x: 0.5)
                         for j in range(K)]
        ens = DiffusionsEnsemble(models, transforms)

        def run(perturb=0.0, ps=0):
            handles = []
            if perturb:
                gen = torch.Generator().manual_seed(ps)
                for j in range(K):
                    handles.append(models[j].model.register_forward_hook(
                        lambda mod, inp, o: o * (1 + perturb * torch.empty(o.shape).normal_(generator=gen))))
            with quiet(), injected_noise(draws) as q:
                chains = ens.run_local_inference(torch.from_numpy(seed), n_noise, n_denoise, None, hard_conds, cross_conds=cross_conds,
                                                 n_samples=B, return_chain=True, sample_fn=ddpm_sample_fn, sample_kwargs=sample_kwargs,
                                                 n_diffusion_steps_without_noise=1)
                assert len(q) == 0
            for h in handles:
                h.remove()
            return {j: chains[j].clone() for j in range(K)}

        chains = run()
        sens = {j: np.zeros(n_denoise + 2) for j in range(K)}
        for ps in range(1, N_SENS_DRAWS + 1):
            pert = run(1e-6, ps)
            for j in range(K):
                sens[j] = np.maximum(sens[j], [rel_l2(pert[j][r].numpy(), chains[j][r].numpy()) for r in range(n_denoise + 2)])
        for j in range(K):
            out[f"{direction}.chain{j}"] = chains[j].numpy()
            out[f"{direction}.sens{j}"] = sens[j]
            guides[j].reset_extra_costs()
        out[f"{direction}.seed"] = seed
        out[f"{direction}.meta"] = np.array([T, B, K, n_noise, n_denoise])
        print(f"   g22 {direction}: chain shapes {[tuple(chains[j].shape) for j in range(K)]}, final-row sens {[float(sens[j][-1]) for j in range(K)]}", flush=True)
    np.savez_compressed(os.path.join(OUT, "g22_ensemble3_local.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    todo = sys.argv[1:] or ["g1", "g2", "g3", "g45", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15", "g16", "g17", "g18", "g19", "g20", "g21", "g22"]
    for name in todo:
        print("generating", name, flush=True)
        {"g1": g1, "g2": g2, "g3": g3, "g45": g4_g5, "g6": g6, "g7": g7, "g8": g8, "g9": g9, "g10": g10, "g11": g11, "g12": g12, "g13": g13, "g14": g14, "g6full": g6_full, "g15": g15, "g16": g16, "g17": g17, "g18": g18, "g19": g19, "g20": g20, "g21": g21, "g22": g22}[name]()
    print("done")
