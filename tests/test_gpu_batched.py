"""-m gpu: planners.plan_batched -- R independent planner calls as ONE launch sequence, bitwise equal to the calls made
one after the other.  The reference's granularity is one planner call per agent with 64 samples (inference_multi_agent.py:225-237,
cbs.py:316-324, mmd_params.py:33); a UNet launch of 64 trajectories costs what one of 256 does."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mmd_amd import synth                # noqa: E402
import cases                             # noqa: E402
from cases import H, D                   # noqa: E402


def _same_output(a, b, ensemble=False):
    assert torch.equal(a.trajs_iters, b.trajs_iters)
    assert torch.equal(a.trajs_final, b.trajs_final)
    assert torch.equal(a.trajs_final_free_idxs, b.trajs_final_free_idxs) and torch.equal(a.trajs_final_coll_idxs, b.trajs_final_coll_idxs)
    assert a.trajs_final_free_idxs.shape == b.trajs_final_free_idxs.shape and a.trajs_final_free_idxs.dtype == b.trajs_final_free_idxs.dtype
    assert a.success_free_trajs == b.success_free_trajs and a.fraction_free_trajs == b.fraction_free_trajs
    for name in ("trajs_final_free", "trajs_final_coll", "cost_smoothness", "cost_path_length", "cost_all", "traj_final_free_best"):
        u, v = getattr(a, name), getattr(b, name)
        assert (u is None) == (v is None), name
        if u is not None:
            assert torch.equal(u, v), name
    assert (a.idx_best_traj is None) == (b.idx_best_traj is None)
    if a.idx_best_traj is not None:
        assert int(a.idx_best_traj) == int(b.idx_best_traj)
        assert float(a.cost_best_free_traj) == float(b.cost_best_free_traj)
        assert float(a.variance_waypoint_trajs_final_free) == float(b.variance_waypoint_trajs_final_free)
    assert b.t_total > 0 and a.constraints_l is b.constraints_l


def _mpd(env, start, goal, seed, B=16, T=25, **over):
    from mmd_amd.planners import MPD
    kw = dict(model_id=env + "-RobotPlanarDisk", planner_alg="mmd", start_state_pos=torch.as_tensor(start), goal_state_pos=torch.as_tensor(goal),
              device="cuda", seed=seed, n_samples=B, model_state_dict=synth.synth_unet_state_dict(0), model_args=dict(n_diffusion_steps=T),
              trained_models_dir="")
    kw.update(over)
    return MPD(**kw)


def test_plan_batched_equals_the_sequential_calls():
    """Five MPD calls on four different maps with their own start / goal and constraint sets (none, soft from other robots' paths, hard +
    soft) in one [5 * 16, H, D] launch sequence: every PlannerOutput field bitwise the sequential call's under the same seeds; the
    default seeds are the ones a sequential loop draws; a sixth call with different weights, a re-plan from an experience and a
    `diffusion_prior_then_guide` planner ride along in the same list (their own groups / on their own)."""
    from mmd_amd import diffusion_model as dm
    from mmd_amd.constraints import MultiPointConstraint
    from mmd_amd.planners import PathBatchExperience, plan_batched
    starts, goals = synth.start_goal_circle(10, 0.45)
    paths = synth.straight_line_paths(starts, goals, H)

    def soft_for(r):
        return MultiPointConstraint(q_l=[torch.from_numpy(paths[j, t]) for j in range(10) if j != r for t in range(1, H)],
                                    t_range_l=[(t, t + 1) for j in range(10) if j != r for t in range(1, H)], is_soft=True)
    hard = MultiPointConstraint(q_l=[torch.tensor([0.1, 0.2]), torch.tensor([-0.2, 0.1])], t_range_l=[(20, 27), (30, 33)])
    envs = ("EnvHighways2D", "EnvEmpty2D", "EnvConveyor2D", "EnvDropRegion2D", "EnvHighways2D")
    robots = (1, 4, 7, 2, 8)
    ps = [_mpd(e, starts[r], goals[r], 18 + r) for e, r in zip(envs, robots)]
    cons = [[soft_for(1)], None, [soft_for(7), hard], [hard], []]
    calls = [(p, torch.from_numpy(starts[r]), torch.from_numpy(goals[r]), c) for p, r, c in zip(ps, robots, cons)]
    draws = dm._GLOBAL_DRAWS
    seq = [c[0](*c[1:]) for c in calls]
    dm._GLOBAL_DRAWS = draws                                  # the same point of the global stream
    bat = plan_batched(calls)
    for a, b in zip(seq, bat):
        _same_output(a, b)
    assert all(p.guide.extra_cost_l == [[]] and p.recent_call_data is o for p, o in zip(ps, bat))
    assert not torch.equal(bat[0].trajs_iters[-1], bat[4].trajs_iters[-1])
    # explicit seeds + calls that cannot be packed with the others
    other = _mpd("EnvEmpty2D", starts[5], goals[5], 30, model_state_dict=synth.synth_unet_state_dict(3))       # other weights: own group
    ptg = _mpd("EnvHighways2D", starts[6], goals[6], 31, planner_alg="diffusion_prior_then_guide")
    prior = [_mpd("EnvEmpty2D", starts[r], goals[r], 40 + r, planner_alg="diffusion_prior") for r in (0, 3)]  # a prior-only group
    calls2 = calls + [(other, torch.from_numpy(starts[5]), torch.from_numpy(goals[5])),
                      (ptg, torch.from_numpy(starts[6]), torch.from_numpy(goals[6]), [hard])] + \
        [(p, torch.from_numpy(starts[r]), torch.from_numpy(goals[r])) for p, r in zip(prior, (0, 3))]
    calls2[1] = calls[1] + (PathBatchExperience(seq[1].trajs_final),)                                          # a re-plan: its own group
    seeds = [900 + j for j in range(len(calls2))]
    seq2 = [c[0](*c[1:], seed=s) for c, s in zip(calls2, seeds)]
    bat2 = plan_batched(calls2, seeds=seeds)
    for a, b in zip(seq2, bat2):
        _same_output(a, b)
    assert bat2[1].trajs_iters.shape[0] == 5 and bat2[6].trajs_iters.shape[0] > 27
    with pytest.raises(ValueError):
        plan_batched([calls[0], calls[0]])
    with pytest.raises(ValueError):
        plan_batched([(ps[0], torch.from_numpy(goals[1]), torch.from_numpy(goals[1]))] + calls[1:])


def test_plan_batched_replans_from_experiences():
    """The two children of a CBS expansion re-plan two different agents from their previous batches (cbs.py:397-432, xCBS / xECBS:
    experience = the agent's last trajs_final) -- three such re-plans on different maps, each with its own hard + soft constraints and
    its own experience, as ONE launch sequence: the forward noising under the call's own Philox stream, one denoising loop, every
    PlannerOutput field bitwise the sequential call's; an MPD whose local-inference step counts differ forms its own group."""
    from mmd_amd.constraints import MultiPointConstraint
    from mmd_amd.planners import PathBatchExperience, plan_batched
    starts, goals = synth.start_goal_circle(10, 0.45)
    paths = synth.straight_line_paths(starts, goals, H)
    envs, robots = ("EnvHighways2D", "EnvConveyor2D", "EnvEmpty2D", "EnvDropRegion2D"), (2, 5, 9, 0)
    ps = [_mpd(e, starts[r], goals[r], 50 + r) for e, r in zip(envs[:3], robots[:3])]
    ps.append(_mpd(envs[3], starts[0], goals[0], 50, n_local_inference_noising_steps=5, n_local_inference_denoising_steps=4))
    first = [p(torch.from_numpy(starts[r]), torch.from_numpy(goals[r]), seed=100 + r) for p, r in zip(ps, robots)]

    def cons(r, t0):
        soft = MultiPointConstraint(q_l=[torch.from_numpy(paths[j, t]) for j in range(10) if j != r for t in range(1, H, 3)],
                                    t_range_l=[(t, t + 1) for j in range(10) if j != r for t in range(1, H, 3)], is_soft=True)
        return [MultiPointConstraint(q_l=[torch.from_numpy(paths[r, t0 + 2])], t_range_l=[(t0, t0 + 5)]), soft]
    calls = [(p, torch.from_numpy(starts[r]), torch.from_numpy(goals[r]), cons(r, 10 + 7 * k), PathBatchExperience(o.trajs_final))
             for k, (p, r, o) in enumerate(zip(ps, robots, first))]
    seeds = [3100, 3101, 3102, 3103]
    seq = [c[0](*c[1:], seed=s) for c, s in zip(calls, seeds)]
    bat = plan_batched(calls, seeds=seeds)
    for a, b in zip(seq, bat):
        _same_output(a, b)
    assert [o.trajs_iters.shape[0] for o in bat] == [5, 5, 5, 6]            # n_denoising + 1 no-noise step + the seed row
    assert not torch.equal(bat[0].trajs_iters[0], first[0].trajs_final)     # (row 0 is the forward-noised seed batch)
    assert all(p.guide.extra_cost_l == [[]] for p in ps)


def _config4_planners(B=8, T=25):
    """The reference's multi_tile example (inference_multi_agent.py:418-431): 4 agents on the 1 x 2 EnvEmptyNoWait2D grid, skeletons
    alternating [[0,0],[0,1]] / [[0,1],[0,0]] -- the relative tile transform is +2 for agents 0 and 2 and -2 for agents 1 and 3."""
    from mmd_amd.planners import MPDEnsemble
    sd = synth.synth_unet_state_dict(0)
    starts = torch.tensor([[0, 0.8], [0, 0.3], [0, -0.3], [0, -0.8]])
    goals = torch.tensor([[0, -0.8], [0, -0.3], [0, 0.3], [0, 0.8]])
    calls = []
    for r in range(4):
        sk = [[0, 0], [0, 1]] if r % 2 == 0 else [[0, 1], [0, 0]]
        tr = {j: torch.tensor([c * 2.0, -row * 2.0]) for j, (row, c) in enumerate(sk)}
        start, goal = starts[r] + tr[0], goals[r] + tr[1]
        calls.append((MPDEnsemble(model_ids=("EnvEmptyNoWait2D-RobotPlanarDisk",) * 2, transforms=tr, planner_alg="mmd",
                                  start_state_pos=start, goal_state_pos=goal, n_samples=B, model_state_dicts=[sd, sd],
                                  model_args=dict(n_diffusion_steps=T), device="cuda", seed=18 + r), start, goal))
    return calls


def test_plan_batched_ensembles_equal_the_sequential_calls():
    """(a) config 4's four MPDEnsemble calls (per-robot tile transforms: the cross conditioning's relative transform is +2 for two of
    them and -2 for the others) with a constraint on one of them; (b) the two 3-tile corner-turning instances of golden g20 (different
    maps per tile and per call, different weights per tile would split the group: both use the g19 weights here) with their
    constraints routed per call -- each as one launch sequence, bitwise the sequential calls."""
    from mmd_amd.constraints import MultiPointConstraint
    from mmd_amd.planners import MPDEnsemble, plan_batched
    calls = _config4_planners()
    c1 = MultiPointConstraint(q_l=[torch.tensor([1.7, 0.1]), torch.tensor([0.3, 0.2])], t_range_l=[(10, 14), (64 + 20, 64 + 26)])
    calls[1] = calls[1] + ([c1],)
    seeds = [7001, 7002, 7003, 7004]
    seq = [c[0](*c[1:], seed=s) for c, s in zip(calls, seeds)]
    bat = plan_batched(calls, seeds=seeds)
    for a, b in zip(seq, bat):
        assert a.trajs_iters.shape[-2] == 2 * H
        _same_output(a, b, ensemble=True)
    assert all(g.extra_cost_l == [[]] for c in calls for g in c[0].guides.values())
    # (b)
    calls3 = []
    for direction in cases.ENSEMBLE3_DIRECTIONS:
        case = synth.ensemble3_case(direction)
        K = len(case["env_ids"])
        p = MPDEnsemble(model_ids=tuple(e + "-RobotPlanarDisk" for e in case["env_ids"]),
                        transforms={m: torch.from_numpy(case["transforms"][m]) for m in range(K)}, planner_alg="mmd",
                        start_state_pos=torch.from_numpy(case["start"]), goal_state_pos=torch.from_numpy(case["goal"]),
                        model_state_dicts=[cases.named_state_dict("g19")] * K, model_args=dict(n_diffusion_steps=25), n_samples=8,
                        device="cuda", trained_models_dir="")
        cl = [MultiPointConstraint(q_l=[torch.from_numpy(q) for q in qs], t_range_l=[tuple(int(v) for v in t) for t in tr],
                                   radius_l=[float(r) for r in rad], is_soft=soft) for (qs, tr, rad, soft) in case["constraints"]]
        calls3.append((p, torch.from_numpy(case["start"]), torch.from_numpy(case["goal"]), cl))
    seq3 = [c[0](*c[1:], seed=s) for c, s in zip(calls3, (811, 812))]
    bat3 = plan_batched(calls3, seeds=(811, 812))
    for a, b in zip(seq3, bat3):
        assert a.trajs_iters.shape[-2] == 3 * H
        _same_output(a, b, ensemble=True)
    assert any(len(o.trajs_final_coll_idxs) for o in bat3)        # (the per-tile collision split bites on these maps)
    # (c) re-plans from experiences (the previous trajs_final, global frame, K * 64 points): config 4's four calls with alternating
    # tile transforms -- each seed batch goes into ITS call's tile frames -- and the two 3-tile calls
    from mmd_amd.planners import PathBatchExperience
    for cs, prev, sds in ((calls, seq, (7101, 7102, 7103, 7104)), (calls3, seq3, (821, 822))):
        cs = [c[:3] + ((c[3] if len(c) > 3 else None), PathBatchExperience(o.trajs_final)) for c, o in zip(cs, prev)]
        seq_x = [c[0](*c[1:], seed=s) for c, s in zip(cs, sds)]
        bat_x = plan_batched(cs, seeds=sds)
        for a, b in zip(seq_x, bat_x):
            assert a.trajs_iters.shape[0] == 5
            _same_output(a, b, ensemble=True)


def test_robot_seeds_reproduce_separate_calls_through_every_step_kernel():
    """mmd_sampler_desc.robot_seeds_dev at the C-ABI level, on the launch shapes the host layer above does not reach: a batch of 3 robots x
    200 samples (the four-trajectories-per-workgroup UNet kernel with its fused unguided tail, the eight-trajectory guide kernel) and
    T = 100 (the persistent run of unguided steps), against three separate one-robot calls."""
    import gpu_common
    from mmd_amd import _lib
    from mmd_amd.diffusion_model import ddpm_sample_fn
    T, R, B = 100, 3, 200
    model = gpu_common.hip_model(T)
    starts, goals = synth.start_goal_circle(6, 0.8)
    paths = synth.straight_line_paths(starts, goals, H)
    hc = {0: torch.stack([cases.hard_conds_for(starts[r], goals[r])[0] for r in range(R)]),
          H - 1: torch.stack([cases.hard_conds_for(starts[r], goals[r])[H - 1] for r in range(R)])}
    seeds = [55, 2 ** 40 + 3, 77]
    kw = dict(horizon=H, return_chain=False, sample_fn=ddpm_sample_fn, n_guide_steps=20, t_start_guide=50,
              noise_std_extra_schedule_fn=lambda t: 0.5, n_diffusion_steps_without_noise=1)
    try:
        for flags in (0, _lib.SAMPLER_PERSIST, _lib.SAMPLER_NO_FUSED_STEP):
            model.sampler_flags = flags
            guide = gpu_common.hip_guide("EnvHighways2D", [[cases.soft_group(paths, r)] for r in range(R)], n_robots=R)
            got = model.run_inference(None, hc, n_samples=B, n_robots=R, guide=guide, robot_seeds=seeds, **kw)
            for r in range(R):
                g1 = gpu_common.hip_guide("EnvHighways2D", [[cases.soft_group(paths, r)]])
                one = model.run_inference(None, {k: v[r] for k, v in hc.items()}, n_samples=B, n_robots=1, guide=g1, seed=seeds[r], **kw)
                assert torch.equal(got[r * B:(r + 1) * B], one), (flags, r)
    finally:
        model.sampler_flags = 0


def test_robot_seeds_across_stream_chunks():
    """32 robots x 64 samples = 2048 trajectories: mmd_p_sample_loop splits the robots into two stream chunks (and uses the
    four-trajectories-per-workgroup UNet kernel); with per-robot Philox streams every robot's rows must still be exactly what its own
    one-robot call with that seed draws (robots 0, 15, 16 -- the chunk boundary -- and 31 are checked)."""
    import gpu_common
    from mmd_amd import _lib
    from mmd_amd.diffusion_model import ddpm_sample_fn
    T, R, B = 25, 32, 64
    model = gpu_common.hip_model(T)
    assert _lib.load().mmd_sampler_stream_chunks(0, R, B) == 2
    starts, goals = synth.start_goal_circle(R, 0.8)
    paths = synth.straight_line_paths(starts, goals, H)
    hc = {0: torch.stack([cases.hard_conds_for(starts[r], goals[r])[0] for r in range(R)]),
          H - 1: torch.stack([cases.hard_conds_for(starts[r], goals[r])[H - 1] for r in range(R)])}
    seeds = [9000 + 7 * r for r in range(R)]
    kw = dict(horizon=H, return_chain=False, sample_fn=ddpm_sample_fn, n_guide_steps=20, t_start_guide=13,
              noise_std_extra_schedule_fn=lambda t: 0.5, n_diffusion_steps_without_noise=1)
    guide = gpu_common.hip_guide("EnvEmpty2D", [[cases.soft_group(paths, r)] for r in range(R)], n_robots=R)
    got = model.run_inference(None, hc, n_samples=B, n_robots=R, guide=guide, robot_seeds=seeds, **kw)
    for r in (0, 15, 16, 31):
        g1 = gpu_common.hip_guide("EnvEmpty2D", [[cases.soft_group(paths, r)]])
        one = model.run_inference(None, {k: v[r] for k, v in hc.items()}, n_samples=B, n_robots=1, guide=g1, seed=seeds[r], **kw)
        assert torch.equal(got[r * B:(r + 1) * B], one), r
    with pytest.raises(ValueError):
        model.run_inference(None, hc, n_samples=B, n_robots=R, guide=guide, robot_seeds=seeds[:5], **kw)


def test_soft_paths_tensor_equals_the_constraint_list():
    """MPD.__call__(..., soft_paths=(paths_all, agent)) -- the other agents' best paths as ONE device tensor, the soft group built on the
    device and placed after the hard groups -- against the reference's form of the same call: a MultiPointConstraint of 9 x 63 tiny
    tensors appended to constraints_l as cbs.py:407-413 / :468-508 build it.  Bitwise, with and without hard constraints, for a fresh
    plan and for a re-plan from an experience; and the guide is left clean."""
    from mmd_amd.constraints import MultiPointConstraint
    from mmd_amd.planners import PathBatchExperience
    starts, goals = synth.start_goal_circle(10, 0.45)
    paths = synth.straight_line_paths(starts, goals, H)
    r = 3
    p = _mpd("EnvHighways2D", starts[r], goals[r], 21)
    s, g = torch.from_numpy(starts[r]), torch.from_numpy(goals[r])
    paths_dev = torch.from_numpy(paths).cuda()
    soft = MultiPointConstraint(q_l=[paths_dev[j, t] for j in range(10) if j != r for t in range(1, H)],
                                t_range_l=[(t, t + 1) for j in range(10) if j != r for t in range(1, H)], is_soft=True)
    hard = [MultiPointConstraint(q_l=[torch.tensor([0.1, 0.2])], t_range_l=[(20, 27)]),
            MultiPointConstraint(q_l=[torch.tensor([-0.2, 0.1]), torch.tensor([0.0, 0.05])], t_range_l=[(30, 33), (40, 47)])]
    first = p(s, g, seed=400)
    exp = PathBatchExperience(first.trajs_final)
    for k, (h, e) in enumerate(((hard, None), ([], None), (hard[:1], exp), (None, exp))):
        a = p(s, g, list(h or []) + [soft], e, seed=410 + k)
        b = p(s, g, h, e, soft_paths=(paths_dev, r), seed=410 + k)
        for name in ("trajs_iters", "trajs_final", "trajs_final_free_idxs", "trajs_final_coll_idxs"):
            assert torch.equal(getattr(a, name), getattr(b, name)), (k, name)
        assert (a.idx_best_traj is None) == (b.idx_best_traj is None) and (a.idx_best_traj is None or int(a.idx_best_traj) == int(b.idx_best_traj))
        assert p.guide.extra_cost_l == [[]] and p.guide._soft_paths is None
    # the soft group does something: without it the call differs
    assert not torch.equal(p(s, g, hard, seed=410).trajs_final, p(s, g, hard, soft_paths=(paths_dev, r), seed=410).trajs_final)
    with pytest.raises(ValueError):
        p(s, g, hard, soft_paths=(paths_dev[:, :32], r))
    assert p.guide.extra_cost_l == [[]] and p.guide._soft_paths is None
