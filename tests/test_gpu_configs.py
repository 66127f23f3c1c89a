"""-m gpu: BASELINE.json's five configs as concrete synthetic inputs, each run end to end through the
product classes on one GPU and checked through domain properties, plus teacher-forced guided steps against the oracle: at the
short schedule (T = 25) inside the sampled chain, and at BASELINE's full size (B = 64, T = 100: 384 / 640 / 512 trajectories for
configs 2, 3 and the per-GPU shard of config 5 -- both sides of the unet_kernel<2> / <4> launch threshold) on >= 4
trajectories each.  Bound everywhere: the north-star 1e-3, or 1.5 x the oracle's own response to a rounding-sized perturbation
of eps where that is larger (gpu_common.teacher_forced_guided_step)."""
from math import ceil

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mmd_amd import synth                # noqa: E402
from oracle import mmd_oracle as O       # noqa: E402
import cases                             # noqa: E402
from cases import H, D, rel_l2           # noqa: E402


def _sampler(n_robots, env_id, starts, goals, T=25, B=16, **kw):
    import gpu_common
    from mmd_amd.multi_robot import MultiRobotSampler
    return MultiRobotSampler(gpu_common.hip_model(T), starts, goals, env_id=env_id, n_samples=B, **kw)


def _check_hard_conds(s, trajs):
    B = s.n_samples
    assert torch.isfinite(trajs).all()
    assert torch.equal(trajs[:, 0], s.hard_conds[0].repeat_interleave(B, 0))
    assert torch.equal(trajs[:, -1], s.hard_conds[H - 1].repeat_interleave(B, 0))


def _chain_step_vs_oracle(tag, s, sd, tb, chain, k, sl, hc, i, gp, groups, st_k):
    """Chain row k -> k + 1 of the sampled batch slice `sl` (one robot of sampler `s`), per trajectory, against the oracle
    restarted from row k: the north-star 1e-3, a step beyond it must be a shown branch flip (gpu_common.GuidedStepJudge)."""
    import gpu_common
    jd = gpu_common.GuidedStepJudge(s.model, s.guide, chain[k].clone(), s.hard_conds, i, 13, s.n_local, st_k, chain[k + 1])
    for j in range(sl.start, sl.stop):
        jd.check("config_chain_step_teacher_forced", f"{tag}_traj{j}", j, sd, tb, gp, groups, hc, 2000 + j)


def _picks(n_robots, robot0, n_local, B, seed, n=4):
    """n (global robot, sample) pairs from >= min(n, n_local) different local robots."""
    rng = np.random.Generator(np.random.PCG64(seed))
    robots = rng.choice(n_local, size=min(n, n_local), replace=False)
    return [(robot0 + int(r), int(rng.integers(B))) for r in robots]


def test_config2_full_size_guided_step_vs_oracle():
    """configs[1] at BASELINE size: 6 robots x B = 64 = 384 trajectories (unet_kernel<2>), T = 100, Empty map, no inter-robot
    term; one guided step at i = 49 (the first guided one) and one at i = 10, 4 trajectories each, vs the oracle."""
    import gpu_common
    starts, goals = synth.start_goal_circle(6, 0.8)
    s = _sampler(6, "EnvEmpty2D", starts, goals, T=100, B=64)
    s.set_other_paths(None)
    for i, seeds in ((49, (110, 111)), (10, (112, 113))):
        gpu_common.teacher_forced_guided_step("config_fullsize_guided_step", f"config2_i{i}", s, 100, starts, goals, "EnvEmpty2D",
                                              _picks(6, 0, 6, 64, 31 + i), i, seeds)


def test_config3_full_size_guided_step_vs_oracle():
    """configs[2] at BASELINE size: 10 robots x 64 = 640 trajectories (unet_kernel<4>), T = 100, Highways map, every robot
    soft-constrained by the other nine (9 x 63 points); 4 trajectories of 4 robots vs the oracle."""
    import gpu_common
    starts, goals = synth.start_goal_circle(10, 0.45)
    paths_np = synth.straight_line_paths(starts, goals, H)
    s = _sampler(10, "EnvHighways2D", starts, goals, T=100, B=64)
    s.set_other_paths(torch.from_numpy(paths_np).cuda())
    gpu_common.teacher_forced_guided_step("config_fullsize_guided_step", "config3_i49", s, 100, starts, goals, "EnvHighways2D",
                                          _picks(10, 0, 10, 64, 41), 49, (114, 115), paths_np=paths_np)


def test_config5_full_size_shard_guided_step_vs_oracle():
    """configs[4] at BASELINE size, the per-GPU shard: rank 5 of 8 owns robots 40 .. 47 of the 64-robot Conveyor instance = 512
    trajectories (the largest unet_kernel<2> launch), each robot soft-constrained by the other 63 (63 x 63 points); 4 trajectories
    of 4 local robots vs the oracle, and the shard's sampled rows equal the unsharded run's bitwise in the production noise path
    at T = 100 (the 1024-trajectory unsharded launch runs unet_kernel<4>: the two kernels agree bit for bit)."""
    import gpu_common
    starts, goals = synth.start_goal_boundary(64)
    paths_np = synth.straight_line_paths(starts, goals, H)
    paths = torch.from_numpy(paths_np).cuda()
    shard = _sampler(64, "EnvConveyor2D", starts, goals, T=100, B=64, rank=5, world_size=8)
    shard.set_other_paths(paths)
    assert (shard.robot0, shard.n_local) == (40, 8)
    gpu_common.teacher_forced_guided_step("config_fullsize_guided_step", "config5_rank5_i49", shard, 100, starts, goals,
                                          "EnvConveyor2D", _picks(64, 40, 8, 64, 51), 49, (116, 117), paths_np=paths_np)
    b = shard.sample(seed=77)
    _check_hard_conds(shard, b)
    pair = _sampler(64, "EnvConveyor2D", starts, goals, T=100, B=64, rank=2, world_size=4)      # robots 32 .. 47: 1024 trajectories
    pair.set_other_paths(paths)
    a = pair.sample(seed=77)
    assert torch.equal(a[8 * 64:], b)


def test_config1_single_robot_mpd_b1_t50():
    """configs[0]: single-robot MPD, Empty2D, 1 sample, 50 denoise steps (the reference's CPU-runnable case)."""
    from mmd_amd.planners import MPD
    p = MPD(model_id="EnvEmpty2D-RobotPlanarDisk", planner_alg="mmd", start_state_pos=torch.tensor([-0.8, 0.0]),
            goal_state_pos=torch.tensor([0.8, 0.0]), n_samples=1, model_state_dict=synth.synth_unet_state_dict(0),
            model_args=dict(n_diffusion_steps=50), device="cuda")
    out = p(torch.tensor([-0.8, 0.0]), torch.tensor([0.8, 0.0]))
    assert out.trajs_iters.shape == (52, 1, H, D) and torch.isfinite(out.trajs_iters).all()
    assert torch.allclose(out.trajs_iters[-1][0, 0, :2].cpu(), torch.tensor([-0.8, 0.0]), atol=1e-6)


def test_config2_six_robots_empty_no_interrobot_term():
    """configs[1]: 6-robot Empty circle map, TemporalUnet + SDF/ws/GP guidance, no inter-robot term."""
    starts, goals = synth.start_goal_circle(6, 0.8)
    s = _sampler(6, "EnvEmpty2D", starts, goals)
    s.set_other_paths(None)
    trajs = s.sample(seed=3)
    _check_hard_conds(s, trajs)
    # teacher-forced: one guided step of robot 4 (Empty map: GP + workspace terms only) against the oracle
    xT = torch.from_numpy(synth.synth_noise(100, (96, H, D)))
    st = torch.from_numpy(synth.synth_noise(101, (26, 96, H, D)))
    chain = s.sample(x_init=xT.cuda(), step_noise=st.cuda(), return_chain=True).cpu()
    sd = O.state_dict_to_torch(synth.synth_unet_state_dict(0))
    tb = O.schedule_tables(25)
    gp = cases.guide_params("EnvEmpty2D")
    r, k = 4, 14                                             # chain row k -> k+1 is loop index i = 24 - k = 10 (guided)
    sl = slice(r * 16, (r + 1) * 16)
    hc = cases.hard_conds_for(starts[r], goals[r])
    _chain_step_vs_oracle("config2_T25", s, sd, tb, chain, k, sl, hc, 24 - k, gp, [], st[k])


def test_config3_ten_robots_highways_with_soft_constraints():
    """configs[2]: 10-robot Highways map with inter-robot soft-constraint guidance; robot 3's first guided step is
    checked teacher-forced against the oracle."""
    starts, goals = synth.start_goal_circle(10, 0.45)
    paths_np = synth.straight_line_paths(starts, goals, H)
    s = _sampler(10, "EnvHighways2D", starts, goals, T=25, B=8)
    paths = torch.from_numpy(paths_np).cuda()
    s.set_other_paths(paths)
    xT = torch.from_numpy(synth.synth_noise(102, (80, H, D)))
    st = torch.from_numpy(synth.synth_noise(103, (26, 80, H, D)))
    chain = s.sample(x_init=xT.cuda(), step_noise=st.cuda(), return_chain=True).cpu()      # [27, 80, H, D]
    _check_hard_conds(s, chain[-1].cuda())
    sd = O.state_dict_to_torch(synth.synth_unet_state_dict(0))
    tb = O.schedule_tables(25)
    gp = cases.guide_params("EnvHighways2D")
    r, k = 3, 12                                             # chain row k -> k+1 is loop index i = 24 - k = 12 (first guided)
    sl = slice(r * 8, (r + 1) * 8)
    hc = cases.hard_conds_for(starts[r], goals[r])
    grp = cases.soft_group(paths_np, r)
    _chain_step_vs_oracle("config3_T25", s, sd, tb, chain, k, sl, hc, 24 - k, gp, [grp], st[k])
    # the device-side pick + conflict mask run on the result
    from mmd_amd.multi_agent import check_rr_collisions
    best = s.best_paths(chain[-1].cuda(), paths)
    coll, _ = check_rr_collisions(best)
    assert coll.shape == (H, 10, 10) and not coll[:, torch.arange(10), torch.arange(10)].any()


def test_config4_ensemble_1x2_four_robots():
    """configs[3]: MPDEnsemble multi_tile 1x2 Empty grid, 4 robots, composed models across tiles."""
    from mmd_amd.planners import MPDEnsemble
    sd = synth.synth_unet_state_dict(0)
    tr = {0: torch.tensor([0.0, 0.0]), 1: torch.tensor([2.0, 0.0])}
    for r in range(4):
        start = torch.tensor([-0.7, -0.6 + 0.4 * r])
        goal = torch.tensor([2.7, 0.6 - 0.4 * r])
        p = MPDEnsemble(model_ids=("EnvEmptyNoWait2D-RobotPlanarDisk",) * 2, transforms=tr, planner_alg="mmd",
                        start_state_pos=start, goal_state_pos=goal, n_samples=8, model_state_dicts=[sd, sd],
                        model_args=dict(n_diffusion_steps=25), device="cuda", seed=18 + r)
        out = p(start, goal)
        tf = out.trajs_iters[-1]
        assert tf.shape == (8, 2 * H, D) and torch.isfinite(tf).all()
        assert torch.allclose(tf[:, 0, :2].cpu(), start.expand(8, 2), atol=1e-5)
        assert torch.allclose(tf[:, -1, :2].cpu(), goal.expand(8, 2), atol=1e-5)
        assert float((tf[:, H - 1, :2] - tf[:, H, :2]).abs().max()) < 1e-4


def test_config5_sixty_four_robots_conveyor_sharded_by_eight():
    """configs[4]: 64-robot Conveyor map, robots sharded 8 per GPU.  One GPU plays rank 5 of 8: its shard equals rows
    [40, 48) of the unsharded 64-robot run (injected noise), and the all-pairs table has 63 slots per robot."""
    starts, goals = synth.start_goal_boundary(64)
    paths = torch.from_numpy(synth.straight_line_paths(starts, goals, H)).cuda()
    full = _sampler(64, "EnvConveyor2D", starts, goals, T=25, B=16)
    shard = _sampler(64, "EnvConveyor2D", starts, goals, T=25, B=16, rank=5, world_size=8)
    assert (shard.robot0, shard.n_local) == (40, 8)
    full.set_other_paths(paths)
    shard.set_other_paths(paths)
    assert shard.guide._external_cons[0].shape == (8 * 63, H, 4)
    xT = torch.from_numpy(synth.synth_noise(104, (64 * 16, H, D))).cuda()
    st = torch.from_numpy(synth.synth_noise(105, (26, 64 * 16, H, D))).cuda()
    a = full.sample(x_init=xT, step_noise=st)
    sl = slice(40 * 16, 48 * 16)
    b = shard.sample(x_init=xT[sl].contiguous(), step_noise=st[:, sl].contiguous())
    _check_hard_conds(shard, b)
    assert torch.equal(a[sl], b)
