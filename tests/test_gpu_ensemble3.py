"""-m gpu: the 3-tile, corner-turning, heterogeneous MPDEnsemble instance of golden g20 -- the shape the
reference's multi-tile experiments actually run (mmd_experiment_configs.py:180-221, inference_multi_agent.py:133-175): tile hops in
x AND y, positive ("fwd": +x then -y) and negative-clamp ("rev": +y then -x) relative transforms in apply_cross_conditioning
(sample_functions.py:17-31), a map (EnvHighways2D / EnvDropRegion2D / EnvConveyor2D) and a weight set per tile, one hard + one soft
constraint routed to the tiles by split_cost_constraints_to_tasks (mpd_ensemble.py:431-522), the per-tile collision / free split of
MPDEnsemble.__call__ (tasks_ensemble.py:79-88, 162-225)."""
import os
from math import ceil

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mmd_amd import synth                # noqa: E402
from oracle import mmd_oracle as O       # noqa: E402
import cases                             # noqa: E402
import parity_log                        # noqa: E402
from cases import GOLDEN, H, D, rel_l2   # noqa: E402


def _setup(direction):
    import gpu_common as gc
    g = np.load(os.path.join(GOLDEN, "g20_ensemble3.npz"))
    T, B, K = (int(v) for v in g[f"{direction}.meta"])
    case = synth.ensemble3_case(direction)
    models = {m: gc.hip_model(T, case["weights"][m]) for m in range(K)}
    cons = cases.ensemble3_tile_groups(g, direction, K)
    guides = {m: gc.hip_guide(case["env_ids"][m], [cons[m]], cutoff=0.01) for m in range(K)}
    transforms = {m: torch.from_numpy(case["transforms"][m]) for m in range(K)}
    hard = cases.ensemble3_hard_conds(case)
    cross = {(m, m + 1): (H - 1, 0) for m in range(K - 1)}
    skw = {m: dict(guide=guides[m], n_guide_steps=20, t_start_guide=ceil(0.5 * T), noise_std_extra_schedule_fn=lambda x: 0.5)
           for m in range(K)}
    x0, steps = cases.ensemble3_inputs(case, T, B)
    return g, T, B, K, case, models, cons, guides, transforms, hard, cross, skw, x0, steps


@pytest.mark.parametrize("direction", cases.ENSEMBLE3_DIRECTIONS)
def test_ensemble3_chain_golden_g20(direction):
    """DiffusionsEnsemble.p_sample_loop (ONE C-ABI call for the K = 3 tiles) with injected noise against every chain row of every
    tile of the reference, under the chaos bound used for g6 / g8 (rows before guidance calibrate the per-step deviation)."""
    from mmd_amd.diffusion_ensemble import DiffusionsEnsemble
    from mmd_amd.diffusion_model import ddpm_sample_fn
    g, T, B, K, case, models, cons, guides, transforms, hard, cross, skw, x0, steps = _setup(direction)
    ens = DiffusionsEnsemble(models, transforms)
    x, chains = ens.p_sample_loop((B, H, D), {m: dict(hard[m]) for m in range(K)}, cross, n_diffusion_steps=T, return_chain=True,
                                  sample_fn=ddpm_sample_fn, n_diffusion_steps_without_noise=1, x_init=x0, step_noise=steps.cuda(),
                                  sample_kwargs=skw)
    # the stitched boundaries of the final state: the second assignment of apply_cross_conditioning (sample_functions.py:30) holds
    # exactly for every pair -- start of tile m + 1 = max(end of tile m - rel, -boundary) -- and the coordinate that is not clamped
    # coincides in the global frame
    for m in range(K - 1):
        rel, boundary = (torch.tensor(v) for v in __import__("mmd_amd.diffusion_ensemble", fromlist=["x"])._rel_boundary(transforms, m, m + 1, D))
        a, b = x[m][:, H - 1].cpu(), x[m + 1][:, 0].cpu()
        assert torch.equal(b, torch.max(a - rel, -boundary)), m
        free_axis = int(torch.argmin(rel[:2].abs()))
        assert torch.equal(a[:, free_axis], b[:, free_axis]), (m, free_axis)
    failures = []
    for m in range(K):
        ref, sens = g[f"{direction}.chain{m}"], g[f"{direction}.sens{m}"]
        got = chains[m].transpose(0, 1).cpu()
        assert got.shape == ref.shape
        assert np.array_equal(got[0].numpy(), ref[0]), "row 0: noise + hard conditioning + stitching is exact"
        errs = [rel_l2(got[r], ref[r]) for r in range(ref.shape[0])]
        lin, bounds = cases.chaos_bounds(errs, [float(v) for v in sens], T - ceil(0.5 * T) + 1)
        assert lin < cases.LIN, (m, lin)
        for r in range(ref.shape[0]):
            parity_log.record(f"ensemble3_{direction}_chain_golden", f"tile{m}", r, errs[r], sens=float(sens[r]), bound=bounds[r],
                              note=f"lin = {lin:.2f}")
            if not errs[r] < bounds[r]:
                failures.append((m, r, errs[r], float(sens[r])))
        assert torch.equal(got[-1], x[m].cpu())
    assert not failures, failures


@pytest.mark.parametrize("direction", cases.ENSEMBLE3_DIRECTIONS)
def test_ensemble3_teacher_forced_steps_g20(direction):
    """EVERY outer step of EVERY tile, restarted from the reference's rows: tile step (UNet + posterior + 20 guide iterations + noise
    + hard conditioning) per trajectory against the oracle -- unguided 2e-5, guided the north-star 1e-3 or an ATTRIBUTED branch flip /
    fp32-rounding verdict (GuidedStepJudge) -- then the stitching kernel bit-exact against the oracle's apply_cross_conditioning; an
    outer step without a flip must land within 1e-3 of the reference's next rows."""
    import gpu_common as gc
    from mmd_amd.diffusion_ensemble import apply_cross_conditioning
    g, T, B, K, case, models, cons, guides, transforms, hard, cross, skw, x0, steps = _setup(direction)
    tsg = ceil(0.5 * T)
    sds = [O.state_dict_to_torch(cases.named_state_dict(w)) for w in case["weights"]]
    tb = O.schedule_tables(T)
    gps = [cases.guide_params(e, cutoff=0.01) for e in case["env_ids"]]
    verdicts = {}
    for k, i in enumerate(reversed(range(-1, T))):
        xs = {m: torch.from_numpy(g[f"{direction}.chain{m}"][k]).cuda() for m in range(K)}
        flipped = False
        for m in range(K):
            x_in, nz = xs[m].cpu().clone(), steps[k, m]
            models[m].sample_step(xs[m], hard[m], i, noise=nz.cuda(), **skw[m])
            y = xs[m].cpu().clone()
            if i < tsg:
                judge = gc.GuidedStepJudge(models[m], guides[m], x_in, hard[m], i, tsg, 1, nz, y)
                for b in range(B):
                    v, err = judge.check(f"ensemble3_{direction}_teacher_forced", f"tile{m}_sample{b}", b, sds[m], tb, gps[m],
                                         cons[m], hard[m], 2000 + 100 * m + b)
                    verdicts[v.split("@")[0]] = verdicts.get(v.split("@")[0], 0) + 1
                    flipped |= v != "within"
            else:
                ref = O.apply_hard_conditioning(O.ddpm_sample_step(sds[m], tb, x_in.clone(), hard[m], i, noise=nz,
                                                                   noise_std_extra=0.5), hard[m])
                # (an unguided step multiplies the forward's ~1e-6 by sqrt_recip_alphas_cumprod[t] x posterior_mean_coef1[t] wherever the
                # x0 clamp does not bite -- with the trained weights of "rev" it rarely does at t = T - 1: the reference's own response to
                # a relative 1e-6 perturbation, times LIN, is the yardstick there, as in tests/test_gpu_trained.py)
                err = rel_l2(y, ref)
                bound = max(2e-5, 1.5 * cases.LIN * float(g[f"{direction}.sens{m}"][k + 1]))
                parity_log.record(f"ensemble3_{direction}_teacher_forced", f"tile{m}", i, err, bound=bound)
                assert err < bound, (m, k, i, err, bound)
            xo = O.apply_cross_conditioning({j: xs[j].cpu().clone() for j in range(K)}, cross, transforms)
            xs = apply_cross_conditioning(xs, cross, transforms)
            for j in range(K):
                assert torch.equal(xs[j].cpu(), xo[j]), ("stitch", k, m, j)
        for m in range(K):
            # (the reference's rows of tiles m >= 1 before the last also carry the boundary rows stitched in the NEXT outer step --
            # include/mmd_amd.h: mmd_ensemble_tile.chain_dev; the one-call chain test compares those rows whole)
            rows = slice(0, H) if (m == 0 or k == T) else slice(1, H - 1)
            err = rel_l2(xs[m].cpu()[:, rows], g[f"{direction}.chain{m}"][k + 1][:, rows])
            bound = 1e-3 if i < tsg else max(2e-5, 1.5 * cases.LIN * float(g[f"{direction}.sens{m}"][k + 1]))
            parity_log.record(f"ensemble3_{direction}_outer_step_vs_reference", f"tile{m}", i, err, bound=bound,
                              note="a flip was attributed in this outer step" if flipped else "")
            if not flipped:
                assert err < bound, (m, k, i, err, bound)
    print("ensemble3", direction, "guided tile-step verdicts:", verdicts)
    assert verdicts.get("within", 0) >= 0.97 * sum(verdicts.values()), verdicts


@pytest.mark.parametrize("direction", cases.ENSEMBLE3_DIRECTIONS)
def test_mpd_ensemble_call_on_the_g20_instance(direction):
    """MPDEnsemble.__call__ on the same instance: the planner routes the two MultiPointConstraints itself
    (split_cost_constraints_to_tasks + the tile shift); its chains are BITWISE those of DiffusionsEnsemble.p_sample_loop fed with
    the reference's own per-tile tables (same injected noise) -- the routing produced the same guide tables; the PlannerOutput is
    the oracle's restatement of the reference's post-processing (pinned on the reference's chains by the CPU test of g20) applied
    to the HIP chains: per-tile collision split in the tile frame, free = free in every tile, global concatenation, costs, best
    sample, SavGol; and on the reference's OWN final rows the split is the reference's."""
    from mmd_amd.constraints import MultiPointConstraint
    from mmd_amd.diffusion_ensemble import DiffusionsEnsemble
    from mmd_amd.diffusion_model import ddpm_sample_fn
    from mmd_amd.planners import MPDEnsemble, _fill_output_ensemble, PlannerOutput
    g, T, B, K, case, models, cons, guides, transforms, hard, cross, skw, x0, steps = _setup(direction)
    ens = DiffusionsEnsemble(models, transforms)
    _, chains = ens.p_sample_loop((B, H, D), {m: dict(hard[m]) for m in range(K)}, cross, n_diffusion_steps=T, return_chain=True,
                                  sample_fn=ddpm_sample_fn, n_diffusion_steps_without_noise=1, x_init=x0, step_noise=steps.cuda(),
                                  sample_kwargs=skw)
    p = MPDEnsemble(model_ids=tuple(e + "-RobotPlanarDisk" for e in case["env_ids"]), transforms=transforms, planner_alg="mmd",
                    start_state_pos=torch.from_numpy(case["start"]), goal_state_pos=torch.from_numpy(case["goal"]),
                    model_state_dicts=[cases.named_state_dict(w) for w in case["weights"]], model_args=dict(n_diffusion_steps=T),
                    n_samples=B, device="cuda", trained_models_dir="")
    constraints_l = [MultiPointConstraint(q_l=[torch.from_numpy(q) for q in qs], t_range_l=[tuple(int(v) for v in t) for t in tr],
                                          radius_l=[float(r) for r in rad], is_soft=soft) for (qs, tr, rad, soft) in case["constraints"]]
    out = p(torch.from_numpy(case["start"]), torch.from_numpy(case["goal"]), constraints_l=constraints_l, x_init=x0,
            step_noise=steps.cuda())
    assert out.trajs_iters.shape == (T + 2, B, K * H, D)
    mins, maxs = cases.MINS, cases.MAXS
    for m in range(K):
        hip_chain = chains[m].transpose(0, 1).cpu()                       # [T+2, B, H, D] normalised, tile frame
        tile = out.trajs_iters[:, :, m * H:(m + 1) * H].cpu().clone()
        tile[..., :2] -= transforms[m]
        assert torch.equal(O.normalize(tile, mins, maxs).float()[-1].isfinite(), torch.ones_like(tile[-1], dtype=torch.bool))
        assert rel_l2(tile, O.unnormalize(hip_chain, mins, maxs, clip_mode="always")) < 1e-6, m
    gps = {m: cases.guide_params(case["env_ids"][m], cutoff=0.01) for m in range(K)}
    tr_np = {m: case["transforms"][m] for m in range(K)}
    ref_out = O.ensemble_planner_output({m: chains[m].transpose(0, 1).cpu() for m in range(K)}, gps, tr_np, mins, maxs)

    def same(out, ref_out):
        assert out.trajs_final_free_idxs.cpu().tolist() == ref_out["trajs_final_free_idxs"].tolist()
        assert out.trajs_final_coll_idxs.cpu().tolist() == ref_out["trajs_final_coll_idxs"].tolist()
        assert out.trajs_final_free_idxs.dtype == torch.int64 and out.trajs_final_free_idxs.dim() == 1
        assert out.success_free_trajs == ref_out["success_free_trajs"] and out.fraction_free_trajs == ref_out["fraction_free_trajs"]
        assert tuple(out.trajs_final_coll.shape) == tuple(ref_out["trajs_final_coll"].shape)
        assert tuple(out.trajs_final_free.shape) == tuple(ref_out["trajs_final_free"].shape)
        assert np.allclose(out.trajs_final.cpu().numpy(), ref_out["trajs_final"].numpy(), atol=2e-6)
        if ref_out["success_free_trajs"]:
            assert int(out.idx_best_traj) == int(ref_out["idx_best_traj"])
            assert np.allclose(out.cost_smoothness.cpu().numpy(), ref_out["cost_smoothness"].numpy(), rtol=1e-5)
            assert np.allclose(out.cost_path_length.cpu().numpy(), ref_out["cost_path_length"].numpy(), rtol=1e-5)
            assert abs(float(out.cost_best_free_traj) - float(ref_out["cost_best_free_traj"])) < 1e-4
            assert abs(float(out.variance_waypoint_trajs_final_free) - float(ref_out["variance_waypoint_trajs_final_free"])) < 1e-4
        else:
            assert out.idx_best_traj is None and out.cost_all is None and out.trajs_final_free.numel() == 0
    same(out, ref_out)
    # the post-processing on the REFERENCE's chains against the reference's own PlannerOutput fields (golden g20)
    ref_chains = {m: torch.from_numpy(g[f"{direction}.chain{m}"]) for m in range(K)}
    parts, tile_final = [], {}
    for m in range(K):
        tr = p.datasets[m].unnormalize_trajectories(ref_chains[m].cuda()).clone()
        tile_final[m] = tr[-1].clone()
        tr[..., :2] += transforms[m].cuda()
        parts.append(tr)
    o2 = _fill_output_ensemble(PlannerOutput(), p.task, tile_final, torch.cat(parts, dim=-2))
    assert o2.trajs_final_free_idxs.cpu().tolist() == g[f"{direction}.free_idxs"].tolist()
    assert o2.trajs_final_coll_idxs.cpu().tolist() == g[f"{direction}.coll_idxs"].tolist()
    assert np.allclose(o2.trajs_iters[-1].cpu().numpy(), g[f"{direction}.trajs_final_global"], atol=1e-6)
    assert np.allclose(o2.trajs_final.cpu().numpy(), g[f"{direction}.smoothed"], atol=2e-6)
    if len(g[f"{direction}.free_idxs"]):
        assert int(o2.idx_best_traj) == int(g[f"{direction}.idx_best_traj"])
        assert np.allclose(o2.cost_smoothness.cpu().numpy(), g[f"{direction}.cost_smoothness"], rtol=1e-5)
        assert np.allclose(o2.cost_path_length.cpu().numpy(), g[f"{direction}.cost_path_length"], rtol=1e-5)
        assert abs(float(o2.cost_best_free_traj) - float(g[f"{direction}.cost_best_free_traj"])) < 1e-4
        assert abs(float(o2.variance_waypoint_trajs_final_free) - float(g[f"{direction}.variance_waypoint"])) < 1e-4
    with pytest.raises(ValueError):
        p(torch.from_numpy(case["goal"]), torch.from_numpy(case["goal"]))


def test_ensemble_task_boundary_contract_g21():
    """What CBS / PP read from an MPDEnsemble planner's `.task` (cbs.py:149-156, multi_agent_utils.py:47,84,89): compute_collision on
    stacked GLOBAL positions [n, 2] and on one position [2], infer_task_id_from_q -- against the genuine PlanningTaskEnsemble
    (golden g21: 40 random points per tile, points outside every tile, points on shared tile edges)."""
    from mmd_amd.planners import MPDEnsemble
    g = np.load(os.path.join(GOLDEN, "g21_ensemble_task.npz"))
    case = synth.ensemble3_case("fwd")
    K = len(case["env_ids"])
    p = MPDEnsemble(model_ids=tuple(e + "-RobotPlanarDisk" for e in case["env_ids"]),
                    transforms={m: torch.from_numpy(case["transforms"][m]) for m in range(K)}, planner_alg="mmd",
                    start_state_pos=torch.from_numpy(case["start"]), goal_state_pos=torch.from_numpy(case["goal"]),
                    model_state_dicts=[synth.synth_unet_state_dict(0)] * K, model_args=dict(n_diffusion_steps=25), n_samples=4,
                    device="cuda", trained_models_dir="")
    pts = torch.from_numpy(g["points"])
    assert p.task.infer_task_id_from_q(pts.cuda().unsqueeze(1)).cpu().tolist() == g["task_ids"].tolist()
    coll = p.task.compute_collision(pts)
    assert coll.dtype == torch.int64 and tuple(coll.shape) == tuple(g["collision"].shape)
    assert coll.tolist() == g["collision"].tolist()
    for k, want in zip(g["single_idx"].tolist(), g["single_collision"].tolist()):
        one = p.task.compute_collision(pts[k])
        assert one.dim() == 0 and int(one) == want, (k, int(one), want)
    assert torch.equal(p.task.transform_q(1, torch.zeros(4)), torch.tensor([2.0, 0.0, 0.0, 0.0]))
    assert torch.equal(p.task.inverse_transform_q(2, torch.zeros(2)), torch.tensor([-2.0, 2.0]))
    task_id, task = p.task.infer_task_id_from_q_idx(130)
    assert task_id == 2 and task is p.task.tasks[2]


def test_mpd_ensemble_prior_then_guide_post_steps():
    """planner_alg 'diffusion_prior_then_guide' on an MPDEnsemble (mpd_ensemble.py:540-564): after the UNGUIDED ensemble sample every
    tile runs (t_start_guide + 1) * n_guide_steps guide steps on its own -- its guide with the routed constraints, its own hard
    conditions, no stitching -- appended to its chain.  K = 2 (Highways | DropRegion) with a constraint per tile: the appended rows
    against the oracle's guide steps restarted from the HIP prior sample (every row within 1e-3 -- no noise, the iteration contracts),
    the prior rows bitwise those of planner_alg 'diffusion_prior' under the same seed; K = 3 raises KeyError as the reference does
    (its middle tile has no entry in hard_conds)."""
    from mmd_amd.constraints import MultiPointConstraint
    from mmd_amd.planners import MPDEnsemble
    T, B, n_gs = 25, 4, 3
    tr = {0: torch.tensor([0.0, 0.0]), 1: torch.tensor([2.0, 0.0])}
    envs = ("EnvHighways2D", "EnvDropRegion2D")
    start, goal = torch.tensor([-0.6, 0.5]), torch.tensor([2.6, -0.3])
    kw = dict(model_ids=tuple(e + "-RobotPlanarDisk" for e in envs), transforms=tr, start_state_pos=start, goal_state_pos=goal,
              model_state_dicts=[cases.named_state_dict("g19")] * 2, model_args=dict(n_diffusion_steps=T), n_samples=B, device="cuda",
              trained_models_dir="", n_guide_steps=n_gs)
    cons = [MultiPointConstraint(q_l=[torch.tensor([0.3, 0.2]), torch.tensor([2.1, 0.1])], t_range_l=[(20, 30), (64 + 10, 64 + 20)]),
            MultiPointConstraint(q_l=[torch.tensor([0.6, 0.0])], t_range_l=[(40, 44)], is_soft=True)]
    p = MPDEnsemble(planner_alg="diffusion_prior_then_guide", **kw)
    out = p(start, goal, constraints_l=cons, seed=321)
    n_post = (ceil(0.5 * T) + 1) * n_gs
    assert out.trajs_iters.shape == (T + 2 + n_post, B, 2 * H, D) and torch.isfinite(out.trajs_iters).all()
    prior = MPDEnsemble(planner_alg="diffusion_prior", **kw)(start, goal, constraints_l=cons, seed=321)
    assert torch.equal(out.trajs_iters[:T + 2], prior.trajs_iters)
    assert all(g.extra_cost_l == [[]] for g in p.guides.values())
    # the appended rows per tile against the oracle (normalised tile frame)
    mins, maxs = cases.MINS, cases.MAXS
    groups = {0: [cases.hard_group([[0.3, 0.2]], [[20, 30]]), O.ConstraintGroup(q=torch.tensor([[0.6, 0.0]]), t_range=torch.tensor([[40.0, 44.0]]),
                                                                                 radius=torch.tensor([cases.RADIUS_SOFT]), weight=2e-2)],
              1: [cases.hard_group([[0.1, 0.1]], [[10, 20]])]}
    hard = {0: {0: O.normalize(torch.tensor([-0.6, 0.5, 0.0, 0.0]), mins, maxs)}, 1: {H - 1: O.normalize(torch.tensor([0.6, -0.3, 0.0, 0.0]), mins, maxs)}}
    worst = 0.0
    for m in (0, 1):
        gp = cases.guide_params(envs[m], cutoff=0.01)
        tile = out.trajs_iters[:, :, m * H:(m + 1) * H].cpu().clone()
        tile[..., :2] -= tr[m]
        x = O.normalize(tile[T + 1], mins, maxs)                         # the HIP prior sample of this tile, normalised
        for k in range(n_post):
            x = O.apply_hard_conditioning(x + O.guide_grad(x, gp, groups[m], clip_mode="always"), hard[m])
            err = rel_l2(O.unnormalize(x, mins, maxs, clip_mode="always"), tile[T + 2 + k])
            worst = max(worst, err)
            assert err < 1e-3, (m, k, err)
    parity_log.record("ensemble_prior_then_guide_post_steps", "K2", None, worst, bound=1e-3)
    # K = 3: the middle tile has no hard conditions -> KeyError, as mpd_ensemble.py:552 `self.hard_conds[task_id]`
    case = synth.ensemble3_case("fwd")
    p3 = MPDEnsemble(model_ids=tuple(e + "-RobotPlanarDisk" for e in case["env_ids"]),
                     transforms={m: torch.from_numpy(case["transforms"][m]) for m in range(3)}, planner_alg="diffusion_prior_then_guide",
                     start_state_pos=torch.from_numpy(case["start"]), goal_state_pos=torch.from_numpy(case["goal"]),
                     model_state_dicts=[cases.named_state_dict("g19")] * 3, model_args=dict(n_diffusion_steps=T), n_samples=B,
                     device="cuda", trained_models_dir="", n_guide_steps=2)
    with pytest.raises(KeyError):
        p3(torch.from_numpy(case["start"]), torch.from_numpy(case["goal"]))
    assert all(g.extra_cost_l == [[]] for g in p3.guides.values())


@pytest.mark.parametrize("direction", cases.ENSEMBLE3_DIRECTIONS)
def test_ensemble3_local_inference_golden_g22(direction):
    """DiffusionsEnsemble.run_local_inference (the re-plan path: MPDEnsemble with an experience) on the 3-tile instance against the
    reference (golden g22): row 0 = the whole-seed q_sample + the split into the tile frames (x and y offsets) + conditioning within 1e-6,
    every later row under the chaos bound; each of the 3 + 1 outer steps teacher-forced from the reference's rows within 1e-3 per tile."""
    import gpu_common as gc
    from mmd_amd.diffusion_ensemble import DiffusionsEnsemble, apply_cross_conditioning
    from mmd_amd.diffusion_model import ddpm_sample_fn
    g = np.load(os.path.join(GOLDEN, "g22_ensemble3_local.npz"))
    g20 = np.load(os.path.join(GOLDEN, "g20_ensemble3.npz"))
    T, B, K, n_noise, n_denoise = (int(v) for v in g[f"{direction}.meta"])
    case = synth.ensemble3_case(direction)
    models = {m: gc.hip_model(T, "g19") for m in range(K)}
    cons = cases.ensemble3_tile_groups(g20, direction, K)
    guides = {m: gc.hip_guide(case["env_ids"][m], [cons[m]], cutoff=0.01) for m in range(K)}
    transforms = {m: torch.from_numpy(case["transforms"][m]) for m in range(K)}
    hard = cases.ensemble3_hard_conds(case)
    cross = {(m, m + 1): (H - 1, 0) for m in range(K - 1)}
    tsg = ceil(0.5 * T)
    skw = {m: dict(guide=guides[m], n_guide_steps=20, t_start_guide=tsg, noise_std_extra_schedule_fn=lambda x: 0.5) for m in range(K)}
    seed, qn, steps = cases.ensemble3_local_inputs(case, g, direction, B, K, n_denoise)
    ens = DiffusionsEnsemble(models, transforms)
    chains = ens.run_local_inference(seed.cuda(), n_noise, n_denoise, None, {m: dict(hard[m]) for m in range(K)}, cross_conds=cross,
                                     n_samples=B, return_chain=True, q_noise=qn.cuda(), sample_fn=ddpm_sample_fn, sample_kwargs=skw,
                                     n_diffusion_steps_without_noise=1, step_noise=steps.cuda())
    for m in range(K):
        ref, sens = g[f"{direction}.chain{m}"], g[f"{direction}.sens{m}"]
        got = chains[m].cpu()
        assert got.shape == ref.shape
        assert rel_l2(got[0], ref[0]) < 1e-6, (m, rel_l2(got[0], ref[0]))
        for r in range(ref.shape[0]):
            err, bound = rel_l2(got[r], ref[r]), max(1e-3, 1.5 * cases.LIN * float(sens[r]))
            parity_log.record(f"ensemble3_{direction}_local_inference_golden", f"tile{m}", r, err, sens=float(sens[r]), bound=bound)
            assert err < bound, (m, r, err, bound)
    # teacher-forced outer steps (all guided: i = 2, 1, 0, -1 < t_start_guide)
    for k, i in enumerate(reversed(range(-1, n_denoise))):
        xs = {m: torch.from_numpy(g[f"{direction}.chain{m}"][k]).cuda() for m in range(K)}
        for m in range(K):
            models[m].sample_step(xs[m], hard[m], i, noise=steps[k, m].cuda(), **skw[m])
            xs = apply_cross_conditioning(xs, cross, transforms)
        for m in range(K):
            rows = slice(0, H) if (m == 0 or k == n_denoise) else slice(1, H - 1)
            err = rel_l2(xs[m].cpu()[:, rows], g[f"{direction}.chain{m}"][k + 1][:, rows])
            parity_log.record(f"ensemble3_{direction}_local_outer_step_vs_reference", f"tile{m}", i, err, bound=1e-3)
            assert err < 1e-3, (m, k, i, err)
