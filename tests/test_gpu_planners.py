"""-m gpu: the outer boundary (MPD / MPDEnsemble / DiffusionsEnsemble) over the HIP kernels."""
import os
from math import ceil

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mmd_amd import _lib, synth          # noqa: E402

SG = _lib.signed64(_lib.HARD_ROWS_START_GOAL)   # the start / goal row mask as a torch custom-op int
import cases                             # noqa: E402
from oracle import mmd_oracle as O       # noqa: E402
import parity_log                        # noqa: E402
from cases import GOLDEN, H, D, rel_l2   # noqa: E402


def _mpd_kwargs(**over):
    kw = dict(model_id="EnvHighways2D-RobotPlanarDisk", planner_alg="mmd", use_guide_on_extra_objects_only=False,
              start_guide_steps_fraction=0.5, n_guide_steps=20, n_diffusion_steps_without_noise=1,
              weight_grad_cost_collision=2e-2, weight_grad_cost_smoothness=8e-2, weight_grad_cost_constraints=2e-1,
              weight_grad_cost_soft_constraints=2e-2, factor_num_interpolated_points_for_collision=1.5,
              trajectory_duration=5.0, device="cuda", debug=False, seed=18, results_dir="logs", trained_models_dir="",
              n_samples=16, n_local_inference_noising_steps=3, n_local_inference_denoising_steps=3,
              model_state_dict=synth.synth_unet_state_dict(0), model_args=dict(n_diffusion_steps=25))
    kw.update(over)
    return kw


def test_ensemble_two_tiles_golden():
    """DiffusionsEnsemble.p_sample_loop with injected noise vs the reference (g8), sensitivity-free check on the
    mid-chain row plus teacher-free final rows under the chaos bound used elsewhere."""
    import gpu_common
    from mmd_amd.diffusion_ensemble import DiffusionsEnsemble
    from mmd_amd.diffusion_model import ddpm_sample_fn
    g = np.load(os.path.join(GOLDEN, "g8_ensemble.npz"))
    T, B = 25, 4
    models = {0: gpu_common.hip_model(T), 1: gpu_common.hip_model(T)}
    transforms = {0: torch.tensor([0.0, 0.0]), 1: torch.tensor([2.0, 0.0])}
    ens = DiffusionsEnsemble(models, transforms)
    s = cases.hard_conds_for([-0.7, 0.3], [0, 0])[0]
    gl = cases.hard_conds_for([0.6, -0.4], [0, 0])[0]
    hard = {0: {0: s}, 1: {H - 1: gl}}
    guides = {0: gpu_common.hip_guide("EnvEmptyNoWait2D", [[cases.hard_group([[0.2, 0.1]], [[30, 36]])]], cutoff=0.01),
              1: gpu_common.hip_guide("EnvEmptyNoWait2D", [[cases.hard_group([[-0.3, -0.1]], [[10, 14]])]], cutoff=0.01)}
    x0 = {m: torch.from_numpy(synth.synth_noise(26 + m, (B, H, D))) for m in (0, 1)}
    steps = torch.from_numpy(synth.synth_noise(28, (T + 1, 2, B, H, D))).cuda()
    skw = {m: dict(guide=guides[m], n_guide_steps=20, t_start_guide=ceil(0.5 * T),
                   noise_std_extra_schedule_fn=lambda x: 0.5) for m in (0, 1)}
    x, chains = ens.p_sample_loop((B, H, D), hard, {(0, 1): (H - 1, 0)}, n_diffusion_steps=T, return_chain=True,
                                  sample_fn=ddpm_sample_fn, n_diffusion_steps_without_noise=1, x_init=x0,
                                  step_noise=steps, sample_kwargs=skw)
    assert chains[0].shape == (B, T + 2, H, D)
    # the stitched boundary: end of tile 0 == start of tile 1 shifted by the tile offset (where not clamped)
    rel = torch.tensor([2.0, 0, 0, 0], device="cuda")
    assert torch.allclose(x[0][:, H - 1] - rel, x[1][:, 0], atol=1e-5)
    # EVERY chain row of both tiles against the reference; bound per row max(1e-3, 1.5 * sens) with the reference's own
    # sensitivity to a relative 1e-6 UNet perturbation (stored in g8 by tools/make_golden.py, like g6)
    failures = []
    for m in (0, 1):
        ref, sens = g[f"chain{m}"], g[f"sens{m}"]
        got = chains[m].transpose(0, 1).cpu()
        assert got.shape == ref.shape
        from cases import chaos_bounds
        errs = [rel_l2(got[r], ref[r]) for r in range(ref.shape[0])]
        # rows 0 .. T - t_start_guide are unguided (well conditioned): they calibrate the kernel's per-step deviation `lin`
        lin, bounds = chaos_bounds(errs, [float(v) for v in sens], T - ceil(0.5 * T) + 1)
        assert lin < cases.LIN, (m, lin)
        for r in range(ref.shape[0]):
            err, bound = errs[r], bounds[r]
            parity_log.record("ensemble_two_tiles_golden", f"tile{m}", r, err, sens=float(sens[r]), bound=bound, note=f"lin = {lin:.2f}")
            if not err < bound:
                failures.append((m, r, err, float(sens[r])))
        assert torch.equal(got[-1], x[m].cpu())
    assert not failures, failures
    # teacher-forced per outer step: restart both tiles from the reference's rows k, run the product loop for ONE outer step
    # (tile 0 step, stitch, tile 1 step, stitch) and compare with the reference's rows k + 1
    worst = 0.0
    for k in range(T + 1):
        xk = {m: torch.from_numpy(g[f"chain{m}"][k]) for m in (0, 1)}
        i = T - 1 - k
        # drive the two tiles through sample_step + apply_cross_conditioning exactly as the C loop orders them
        from mmd_amd.diffusion_ensemble import apply_cross_conditioning
        xs = {m: xk[m].clone().cuda() for m in (0, 1)}
        for j, m in enumerate((0, 1)):
            models[m].sample_step(xs[m], hard[m] if m in hard else {}, i, noise=steps[k, j], **skw[m])
            xs = apply_cross_conditioning(xs, {(0, 1): (H - 1, 0)}, transforms)
        for m in (0, 1):
            err = rel_l2(xs[m].cpu(), g[f"chain{m}"][k + 1])
            worst = max(worst, err)
            guided = i < ceil(0.5 * T)
            parity_log.record("ensemble_teacher_forced", f"tile{m}", k, err, bound=1e-3 if guided else 2e-5)
            assert err < (1e-3 if guided else 2e-5), (m, k, i, err)


def test_ensemble_local_inference_warm_start():
    """MPDEnsemble with an experience (XCBS re-plan): DiffusionsEnsemble.run_local_inference forward-noises the whole
    [B, K*64, 4] seed with models[0].q_sample and splits it per tile (diffusion_ensemble.py:279-300).  Row 0 of every
    tile's chain is checked against the oracle's q_sample + tile split + hard / cross conditioning (the
    second tile used to start from uninitialised memory)."""
    import gpu_common
    from mmd_amd.diffusion_ensemble import DiffusionsEnsemble
    from mmd_amd.diffusion_model import ddpm_sample_fn
    from oracle import mmd_oracle as O
    T, B, K = 25, 4, 2
    models = {0: gpu_common.hip_model(T), 1: gpu_common.hip_model(T)}
    transforms = {0: torch.tensor([0.0, 0.0]), 1: torch.tensor([2.0, 0.0])}
    ens = DiffusionsEnsemble(models, transforms)
    s = cases.hard_conds_for([-0.7, 0.3], [0, 0])[0]
    gl = cases.hard_conds_for([0.6, -0.4], [0, 0])[0]
    hard = {0: {0: s}, 1: {H - 1: gl}}
    a = torch.linspace(0, 1, K * H)[None, :, None]
    seed = torch.cat((torch.tensor([-0.7, 0.3]) * (1 - a) + torch.tensor([2.6, -0.4]) * a, torch.zeros(1, K * H, 2)), -1)
    seed = (seed.repeat(B, 1, 1) + 0.01 * torch.from_numpy(synth.synth_noise(58, (B, K * H, D)))).float()
    skw = {m: dict(guide=None, n_guide_steps=20, t_start_guide=13, noise_std_extra_schedule_fn=lambda x: 0.5) for m in (0, 1)}
    # injected q_sample noise: patch through the model's q_sample
    qn = torch.from_numpy(synth.synth_noise(59, (B, K * H, D)))
    noised = models[0].q_sample(seed.cuda(), 3, noise=qn.cuda())
    ref_noised = O.q_sample(O.schedule_tables(T), seed, 3, qn)
    assert rel_l2(noised.cpu(), ref_noised) < 1e-6
    chains = ens.run_local_inference(seed.cuda(), None, 3, None, hard, cross_conds={(0, 1): (H - 1, 0)}, n_samples=B,
                                     return_chain=True, sample_fn=ddpm_sample_fn, sample_kwargs=skw,
                                     n_diffusion_steps_without_noise=1)
    assert chains[0].shape == (5, B, H, D) and all(torch.isfinite(c).all() for c in chains.values())
    chains = ens.run_local_inference(seed.cuda(), 3, 3, None, hard, cross_conds={(0, 1): (H - 1, 0)}, n_samples=B,
                                     return_chain=True, sample_fn=ddpm_sample_fn, sample_kwargs=skw,
                                     n_diffusion_steps_without_noise=1)
    assert chains[0].shape == (5, B, H, D) and all(torch.isfinite(c).all() for c in chains.values())
    # the warm start of both tiles is the noised seed (in-kernel noise here: compare its statistics with the seed)
    for m in (0, 1):
        x0 = chains[m][0].cpu()
        tile = seed[:, m * H:(m + 1) * H].clone()
        tile[..., :2] -= transforms[m]
        dev = (x0[:, 1:-1] - float(models[0].sqrt_alphas_cumprod[3]) * tile[:, 1:-1]).std()
        assert abs(float(dev) - float(models[0].sqrt_one_minus_alphas_cumprod[3])) < 0.02, (m, float(dev))


def test_mpd_call_contract():
    from mmd_amd.planners import MPD, PlannerOutput
    from mmd_amd.constraints import MultiPointConstraint
    starts, goals = synth.start_goal_circle(10, 0.45)
    with pytest.raises(NotImplementedError):
        MPD(start_state_pos=starts[3], goal_state_pos=goals[3], **_mpd_kwargs(planner_alg="bogus"))
    p = MPD(start_state_pos=torch.from_numpy(starts[3]), goal_state_pos=torch.from_numpy(goals[3]), **_mpd_kwargs())
    with pytest.raises(ValueError):
        p(torch.from_numpy(starts[2]), torch.from_numpy(goals[3]))
    paths = synth.straight_line_paths(starts, goals, H)
    soft = MultiPointConstraint(q_l=[torch.from_numpy(paths[j, t]) for j in range(10) if j != 3 for t in range(1, H)],
                                t_range_l=[(t, t + 1) for j in range(10) if j != 3 for t in range(1, H)])
    soft.is_soft = True
    hard = MultiPointConstraint(q_l=[torch.tensor([0.1, 0.2])], t_range_l=[(20, 27)])
    out = p(torch.from_numpy(starts[3]), torch.from_numpy(goals[3]), constraints_l=[soft, hard])
    assert isinstance(out, PlannerOutput)
    assert out.trajs_iters.shape == (27, 16, H, D) and out.trajs_final.shape == (16, H, D)
    assert torch.isfinite(out.trajs_iters).all() and out.t_total > 0
    # hard conditioning: every sample starts / ends at the requested states (un-normalised, zero velocity)
    tf = out.trajs_iters[-1]
    assert torch.allclose(tf[:, 0, :2].cpu(), torch.from_numpy(starts[3]).expand(16, 2), atol=1e-5)
    assert torch.allclose(tf[:, -1, :2].cpu(), torch.from_numpy(goals[3]).expand(16, 2), atol=1e-5)
    assert float(tf[:, 0, 2:].abs().max()) < 1e-6
    n_free = 0 if out.trajs_final_free is None else out.trajs_final_free.shape[0]
    n_coll = 0 if out.trajs_final_coll is None else out.trajs_final_coll.shape[0]
    assert n_free + n_coll == 16
    assert p.guide.extra_cost_l == [[]], "extra costs must be reset after the call (mpd.py:456)"
    # local inference from the previous batch (XCBS re-plan path)
    from mmd_amd.planners import PathBatchExperience
    out2 = p(torch.from_numpy(starts[3]), torch.from_numpy(goals[3]), constraints_l=[hard],
             experience=PathBatchExperience(out.trajs_final))
    assert out2.trajs_iters.shape == (5, 16, H, D) and torch.isfinite(out2.trajs_iters).all()
    # prior only / prior then guide
    p2 = MPD(start_state_pos=torch.from_numpy(starts[3]), goal_state_pos=torch.from_numpy(goals[3]),
             **_mpd_kwargs(planner_alg="diffusion_prior_then_guide", n_samples=4))
    out3 = p2(torch.from_numpy(starts[3]), torch.from_numpy(goals[3]))
    assert out3.trajs_iters.shape[0] == 27 + (13 + 1) * 20


def test_mpd_guide_on_extra_objects_only():
    """use_guide_on_extra_objects_only (mpd.py:216-219): the guide's only collision field is the env's extra ObjectField,
    empty in every shipped map, so the guide is GP prior + constraints (no fixed-object grid, no workspace walls); the
    planner's own collision checks still see the full map."""
    import gpu_common
    from mmd_amd.planners import MPD
    from oracle import mmd_oracle as O
    starts, goals = synth.start_goal_circle(10, 0.45)
    p = MPD(start_state_pos=torch.from_numpy(starts[3]), goal_state_pos=torch.from_numpy(goals[3]),
            **_mpd_kwargs(use_guide_on_extra_objects_only=True, n_samples=8))
    x = torch.from_numpy(synth.synth_noise(130, (8, H, D))) * 0.9
    gp = cases.guide_params("EnvHighways2D")
    gp.sdf_grids = []
    gp.ws_min, gp.ws_max = torch.tensor([-1e6, -1e6]), torch.tensor([1e6, 1e6])
    ref = O.guide_grad(x, gp, [], clip_mode="always")
    assert float((p.guide(x.cuda()).cpu() - ref).abs().max()) < 2e-6
    full = gpu_common.hip_guide("EnvHighways2D", [[]])
    assert float((p.guide(x.cuda()) - full(x.cuda())).abs().max()) > 1e-3          # the map terms really are off
    out = p(torch.from_numpy(starts[3]), torch.from_numpy(goals[3]))
    assert torch.isfinite(out.trajs_iters).all()
    pts = torch.tensor([[0.0, 0.0], [0.45, 0.45], [0.9, -0.2]])
    expect = O.compute_collision(pts, cases.guide_params("EnvHighways2D"))
    assert p.task.compute_collision(pts.cuda()).view(-1).cpu().tolist() == expect.tolist() and expect.any()


def test_mpd_guidance_reduces_constraint_violations():
    """Domain property at full B=64: with the inter-robot term on, samples keep further from the other robots' paths
    than the unguided prior does."""
    from mmd_amd.planners import MPD
    from mmd_amd.constraints import MultiPointConstraint
    starts, goals = synth.start_goal_circle(6, 0.8)
    paths = synth.straight_line_paths(starts, goals, H)
    others = torch.from_numpy(paths[1:]).cuda()                                           # [5,H,2]
    soft = MultiPointConstraint(q_l=[torch.from_numpy(paths[j, t]) for j in range(1, 6) for t in range(1, H)],
                                t_range_l=[(t, t + 1) for j in range(1, 6) for t in range(1, H)])
    soft.is_soft = True

    def violations(alg, cons):
        p = MPD(start_state_pos=torch.from_numpy(starts[0]), goal_state_pos=torch.from_numpy(goals[0]),
                **_mpd_kwargs(model_id="EnvEmpty2D-RobotPlanarDisk", planner_alg=alg, n_samples=64))
        tf = p(torch.from_numpy(starts[0]), torch.from_numpy(goals[0]), constraints_l=cons, seed=5).trajs_iters[-1]
        d = torch.linalg.norm(tf[:, None, 1:-1, :2] - others[None, :, 1:-1], dim=-1)       # [B,5,H-2]
        return float((d < 0.12).float().mean())

    v_prior = violations("diffusion_prior", None)
    v_guided = violations("mmd", [soft])
    assert v_guided < v_prior or v_prior == 0.0, (v_guided, v_prior)


def test_mpd_ensemble_call_contract():
    from mmd_amd.planners import MPDEnsemble
    from mmd_amd.constraints import MultiPointConstraint
    kw = _mpd_kwargs(n_samples=8)
    for k in ("model_id", "model_state_dict"):
        kw.pop(k)
    start, goal = torch.tensor([-0.7, 0.3]), torch.tensor([2.6, -0.4])
    p = MPDEnsemble(model_ids=("EnvEmptyNoWait2D-RobotPlanarDisk", "EnvEmptyNoWait2D-RobotPlanarDisk"),
                    transforms={0: torch.tensor([0.0, 0.0]), 1: torch.tensor([2.0, 0.0])}, start_state_pos=start,
                    goal_state_pos=goal, model_state_dicts=[synth.synth_unet_state_dict(0)] * 2, **kw)
    c0 = MultiPointConstraint(q_l=[torch.tensor([0.2, 0.1])], t_range_l=[(30, 36)])
    c1 = MultiPointConstraint(q_l=[torch.tensor([1.7, -0.1])], t_range_l=[(64 + 10, 64 + 14)])
    from mmd_amd.constraints import CostConstraint
    split = p.split_cost_constraints_to_tasks([
        CostConstraint(None, H, q_l=c.get_q_l(), traj_range_l=c.get_t_range_l(), radius_l=c.radius_l, is_soft=c.is_soft)
        for c in (c0, c1)])
    assert sorted(split) == [0, 1]
    out = p(start, goal, constraints_l=[c0, c1])
    assert out.trajs_iters.shape == (27, 8, 2 * H, D) and torch.isfinite(out.trajs_iters).all()
    tf = out.trajs_iters[-1]
    assert torch.allclose(tf[:, 0, :2].cpu(), start.expand(8, 2), atol=1e-5)
    assert torch.allclose(tf[:, -1, :2].cpu(), goal.expand(8, 2), atol=1e-5)
    # tiles are stitched: consecutive support points across the tile boundary coincide in the global frame
    assert float((tf[:, H - 1, :2] - tf[:, H, :2]).abs().max()) < 1e-4
    assert out.fraction_free_trajs == 1.0
    with pytest.raises(ValueError):
        p(goal, goal)


def test_multi_agent_layer_vs_reference_g10():
    """SURVEY §8f-1 on device: robot-robot collisions (bit-exact mask, midpoints) and the least_collisions scan against
    the reference robot's check_rr_collisions / the conflict totals CBS would count (tests/golden/g10_multi_agent.npz)."""
    from mmd_amd.multi_agent import check_rr_collisions, count_collisions, least_collision_samples
    from oracle import mmd_oracle as O
    g = np.load(os.path.join(GOLDEN, "g10_multi_agent.npz"))
    paths = torch.from_numpy(g["paths"]).cuda()
    coll, mid = check_rr_collisions(paths)
    assert np.array_equal(coll.cpu().numpy(), g["collisions"])
    assert np.array_equal(np.isnan(mid.cpu().numpy()), np.isnan(g["midpoints"]))
    assert np.array_equal(np.nan_to_num(mid.cpu().numpy()), np.nan_to_num(g["midpoints"]))
    samples = torch.from_numpy(g["samples"]).cuda()
    cnt = count_collisions(samples, paths, 0, 1)[0].cpu()
    base = int(O.check_rr_collisions(paths[1:].cpu().permute(1, 0, 2))[0].sum())
    assert (base + 2 * cnt).tolist() == g["conflict_totals"].tolist()
    assert int(least_collision_samples(samples, paths, 0, 1)[0]) == int(np.argmin(g["conflict_totals"]))
    # several local robots at once == one at a time, and == the oracle
    rng_paths = paths + 0.02 * torch.from_numpy(synth.synth_noise(90, tuple(paths.shape))).cuda()
    batch = torch.from_numpy(synth.synth_noise(91, (3 * 8, H, D))).cuda() * 0.5
    c3 = count_collisions(batch, rng_paths, 2, 3).cpu()
    for r in range(3):
        ref = O.count_collisions_with_others(batch[r * 8:(r + 1) * 8, :, :2].cpu(), rng_paths.cpu(), 2 + r)
        assert c3[r].tolist() == ref.tolist()


def test_mpd_loads_checkpoint_and_dataset_from_disk(tmp_path):
    """the released on-disk formats -- <trained_models_dir>/<model_id>/{args.yaml,
    checkpoints/ema_model_current_state_dict.pth} (state-dict keys with the `model.` prefix + the 12 schedule buffers,
    mpd.py:120,167-171) and the dataset's trajs-free.pt for the normaliser limits (trajectories.py:84-112)."""
    import yaml
    from mmd_amd.planners import MPD, normalizer_limits_from_dataset
    from mmd_amd.schedules import diffusion_buffers
    model_id = "EnvEmpty2D-RobotPlanarDisk"
    mdir = tmp_path / "models" / model_id
    (mdir / "checkpoints").mkdir(parents=True)
    yaml.safe_dump(dict(variance_schedule="exponential", n_diffusion_steps=25, predict_epsilon=True, unet_input_dim=32,
                        unet_dim_mults_option=0, use_ema=True, dataset_subdir=model_id, include_velocity=True),
                   open(mdir / "args.yaml", "w"))
    sd = {"model." + k: torch.from_numpy(v) for k, v in synth.synth_unet_state_dict(0).items()}
    sd.update(diffusion_buffers(25))
    torch.save(sd, mdir / "checkpoints" / "ema_model_current_state_dict.pth")
    ddir = tmp_path / "data" / model_id / "0"
    ddir.mkdir(parents=True)
    trajs = torch.from_numpy(synth.synth_noise(110, (20, H, D))) * torch.tensor([0.9, 0.9, 1.2, 1.2])
    torch.save(trajs, ddir / "trajs-free.pt")
    mins, maxs = normalizer_limits_from_dataset(tmp_path / "data" / model_id)
    assert np.allclose(mins, trajs.reshape(-1, D).min(0).values.numpy()) and mins.shape == (D,)
    start, goal = torch.tensor([-0.5, 0.1]), torch.tensor([0.5, -0.1])
    p = MPD(model_id=model_id, planner_alg="mmd", start_state_pos=start, goal_state_pos=goal, n_samples=8,
            trained_models_dir=str(tmp_path / "models"), dataset_dir=str(tmp_path / "data" / model_id), device="cuda")
    assert p.model.n_diffusion_steps == 25 and p.env_id == "EnvEmpty2D"
    assert torch.allclose(p.dataset.normalizer.mins, torch.from_numpy(mins))
    out = p(start, goal)
    assert out.trajs_iters.shape == (27, 8, H, D) and torch.isfinite(out.trajs_iters).all()
    assert torch.allclose(out.trajs_iters[-1][:, 0, :2].cpu(), start.expand(8, 2), atol=1e-5)
    # same weights given in memory -> same samples for the same seed
    p2 = MPD(model_id=model_id, planner_alg="mmd", start_state_pos=start, goal_state_pos=goal, n_samples=8,
             model_state_dict=synth.synth_unet_state_dict(0), model_args=dict(n_diffusion_steps=25),
             normalizer_limits=(mins, maxs), device="cuda")
    assert torch.equal(p2(start, goal, seed=9).trajs_iters, p(start, goal, seed=9).trajs_iters)


def test_postprocess_vs_reference_g9():
    """SURVEY §8f-2 on the device: mmd_postprocess_trajs (collision / free split bit-exact incl. every interpolated
    waypoint, path length + smoothness, SavGol) and mmd_variance_waypoints against the reference's
    PlanningTask.get_trajs_collision_and_free / metrics / smooth_trajs on a Highways batch (g9, g12)."""
    import gpu_common
    from mmd_amd import postprocess as post
    g = np.load(os.path.join(GOLDEN, "g9_post.npz"))
    g12 = np.load(os.path.join(GOLDEN, "g12_boundary.npz"))
    guide = gpu_common.hip_guide("EnvHighways2D", [[]])
    trajs = torch.from_numpy(g["trajs"]).cuda()
    coll, coll_idxs, free, free_idxs, wp = post.get_trajs_collision_and_free(trajs, guide)
    assert free_idxs.reshape(-1).tolist() == g["free_idxs"].tolist() and free_idxs.ndim == 2
    assert sorted(coll_idxs.reshape(-1).tolist()) == sorted(g["coll_idxs"].tolist())
    assert np.array_equal(wp.cpu().numpy(), g["waypoint_collisions"])
    assert free.shape[0] == len(g["free_idxs"]) and coll.shape[0] == len(g["coll_idxs"])
    r = post.postprocess_batch(guide, trajs, smooth=True)
    for name, got, ref, tol in (("smoothness", r.smoothness, g["smoothness"], 1e-6),
                                ("path_length", r.path_length, g["path_length"], 1e-6),
                                ("savgol", r.smoothed, g["smoothed"], 1e-6)):
        err = float(np.max(np.abs(got.cpu().numpy() - ref) / np.maximum(np.abs(ref), 1.0)))
        parity_log.record("postprocess_g9", name, None, err, bound=tol, note="max rel (abs below 1)")
        assert err < tol, (name, err)
    var = float(post.compute_variance_waypoints(trajs))
    parity_log.record("postprocess_g9", "variance_waypoints", None, abs(var - float(g12["variance_waypoints"])) / float(g12["variance_waypoints"]), bound=1e-5)
    assert abs(var - float(g12["variance_waypoints"])) < 1e-5 * float(g12["variance_waypoints"])
    # per-batch pick: argmin of path length + smoothness over the free samples == torch.argmin on the reference's costs
    idx, n_free = post.select_best(r.free_mask, 1, cost_a=r.path_length, cost_b=r.smoothness)
    cost = (g["path_length"] + g["smoothness"])[g["free_idxs"]]
    assert int(n_free) == len(g["free_idxs"]) and int(idx) == int(g["free_idxs"][int(np.argmin(cost))])
    # several robots in one launch == one robot at a time; least-collisions pick = first free sample with the fewest
    rep = torch.cat([trajs, trajs.flip(0)])
    r2 = post.postprocess_batch(guide, rep, n_robots=2, smooth=False)
    assert torch.equal(r2.free_mask[:trajs.shape[0]], r.free_mask) and torch.equal(r2.free_mask[trajs.shape[0]:], r.free_mask.flip(0))
    counts = torch.arange(rep.shape[0], dtype=torch.int32, device="cuda") % 5
    idx2, nf2 = post.select_best(r2.free_mask, 2, counts=counts)
    for rb in range(2):
        fm = r2.free_mask.view(2, -1)[rb].bool().cpu().numpy()
        c = counts.view(2, -1)[rb].cpu().numpy()
        cand = np.where(fm)[0]
        assert int(idx2[rb]) == int(cand[np.argmin(c[cand])]) and int(nf2[rb]) == int(fm.sum())


def test_outer_boundary_contract_cbs_pp_g12():
    """What the reference's CBS / PrioritizedPlanning constructors and loops call on `planner.robot` / `planner.task`
    (cbs.py:142-156,178-192,316-336,427-458; prioritized_planning.py:141-180,260-276;
    mmd/common/multi_agent_utils.py:30-95), executed against MPD and MPDEnsemble with the reference's own outputs (g12)."""
    from mmd_amd.planners import MPD, MPDEnsemble
    g = np.load(os.path.join(GOLDEN, "g12_boundary.npz"))
    starts, goals = synth.start_goal_circle(10, 0.45)
    planners = [MPD(start_state_pos=torch.from_numpy(starts[i]), goal_state_pos=torch.from_numpy(goals[i]),
                    **_mpd_kwargs(n_samples=8)) for i in range(3)]
    p0 = planners[0]
    robot, task = p0.robot, p0.task                      # cbs.py:144,149
    assert p0.tensor_args["device"].type == "cuda" and isinstance(p0.results_dir, str)      # cbs.py:152-153
    dev = p0.tensor_args["device"]
    # --- is_multi_agent_start_goal_states_valid (multi_agent_utils.py:47-95): [N,2] stacks
    for key in ("starts", "close"):
        q = torch.from_numpy(g[key]).to(dev)
        coll, pts = robot.check_rr_collisions(q)
        assert coll.dtype == torch.bool and coll.shape == (q.shape[0], q.shape[0]) and pts.shape == (q.shape[0], q.shape[0], 2)
        assert np.array_equal(coll.cpu().numpy(), g[f"rr_{key}"])
    _, pts = robot.check_rr_collisions(torch.from_numpy(g["close"]).to(dev))
    assert np.array_equal(np.isnan(pts.cpu().numpy()), np.isnan(g["mid_close"]))
    assert np.array_equal(np.nan_to_num(pts.cpu().numpy()), np.nan_to_num(g["mid_close"]))
    wc = task.compute_collision(torch.from_numpy(g["points"]).to(dev))
    assert wc.dtype == torch.bool and tuple(wc.shape) == tuple(g["coll_points"].shape)
    assert np.array_equal(wc.cpu().numpy(), g["coll_points"])
    assert tuple(task.compute_collision(torch.from_numpy(g["points"][5]).to(dev)).shape) == tuple(g["coll_one"].shape)
    # CPU tensors are accepted too and come back on the CPU
    assert not task.compute_collision(torch.from_numpy(g["points"])).is_cuda
    # --- get_conflicts (cbs.py:178-192): (H, n_robots, q_dim) -> (H, n, n), (H, n, n, 2)
    paths = torch.from_numpy(g["paths"]).to(dev)
    coll, pts = robot.check_rr_collisions(paths)
    assert np.array_equal(coll.cpu().numpy(), g["rr_paths"])
    assert np.array_equal(np.nan_to_num(pts.cpu().numpy()), np.nan_to_num(g["mid_paths"]))
    assert torch.nonzero(coll.int()).shape[1] == 3
    # trajectories: [B,H,4] -> [B,H]
    g9 = np.load(os.path.join(GOLDEN, "g9_post.npz"))
    ct = task.compute_collision(torch.from_numpy(g9["trajs"]).to(dev))
    assert np.array_equal(ct.cpu().numpy(), g["coll_trajs"])
    # positions / velocities (cbs.py:178,486)
    x = torch.from_numpy(g9["trajs"]).to(dev)
    assert torch.equal(robot.get_position(x), x[..., :2]) and torch.equal(robot.get_velocity(x), x[..., 2:4])
    assert robot.radius == 0.05 and robot.q_dim == 2
    # --- the planner call surface of the CBS / PP root loops (cbs.py:316-336, prioritized_planning.py:141-180)
    path_bl, ix_best = [], []
    for i, p in enumerate(planners):
        out = p(torch.from_numpy(starts[i]), torch.from_numpy(goals[i]), constraints_l=[])
        assert out.trajs_final_free_idxs.ndim == 2 and out.trajs_final_free_idxs.shape[0] > 0
        ix_best.append(out.idx_best_traj)
        path_bl.append(out.trajs_final)
        for ix_traj in out.trajs_final_free_idxs:                     # cbs.py:448 iterates the free indices
            assert out.trajs_final[ix_traj].shape == (1, H, D)
    best = [path_bl[i][ix].squeeze(0) for i, ix in enumerate(ix_best)]                      # cbs.py:171-172
    pos_b = torch.stack([robot.get_position(b) for b in best]).permute(1, 0, 2)             # cbs.py:189-191
    coll, pts = robot.check_rr_collisions(pos_b)
    assert coll.shape == (H, 3, 3) and pts.shape == (H, 3, 3, 2)
    # --- MPDEnsemble exposes the same surface
    kw = _mpd_kwargs(n_samples=4)
    for k in ("model_id", "model_state_dict"):
        kw.pop(k)
    pe = MPDEnsemble(model_ids=("EnvEmptyNoWait2D-RobotPlanarDisk",) * 2,
                     transforms={0: torch.tensor([0.0, 0.0]), 1: torch.tensor([2.0, 0.0])},
                     start_state_pos=torch.tensor([-0.7, 0.3]), goal_state_pos=torch.tensor([2.6, -0.4]),
                     model_state_dicts=[synth.synth_unet_state_dict(0)] * 2, **kw)
    c, _ = pe.robot.check_rr_collisions(torch.tensor([[-0.7, 0.3], [-0.62, 0.3]], device=dev))
    assert c.tolist() == [[False, True], [True, False]]
    assert not pe.task.compute_collision(torch.tensor([[-0.7, 0.3], [2.6, -0.4]], device=dev)).any()


def test_device_side_sharing_of_models_and_maps():
    """N+1 planner objects (the reference builds one per agent + a reference one,
    inference_multi_agent.py:186-237) hold ONE packed weight blob + ONE time table (sized to the schedule) and ONE SDF
    texture per map on the device."""
    import time
    from mmd_amd import guides, temporal_unet
    from mmd_amd.planners import MPD
    sd = synth.synth_unet_state_dict(3)                               # weights no other test uses
    starts, goals = synth.start_goal_circle(8, 0.8)
    created0, tex0 = temporal_unet.N_DEVICE_MODELS_CREATED, guides.N_TEXTURE_UPLOADS
    t0 = time.perf_counter()
    planners = [MPD(start_state_pos=torch.from_numpy(starts[i]), goal_state_pos=torch.from_numpy(goals[i]),
                    **_mpd_kwargs(model_id="EnvDropRegion2D-RobotPlanarDisk", model_state_dict=sd, n_samples=4))
                for i in range(8)]
    outs = [p(torch.from_numpy(starts[i]), torch.from_numpy(goals[i])) for i, p in enumerate(planners)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert all(torch.isfinite(o.trajs_iters).all() for o in outs)
    assert temporal_unet.N_DEVICE_MODELS_CREATED - created0 == 1, "8 planners must share one mmd_unet_create"
    assert guides.N_TEXTURE_UPLOADS - tex0 <= 1, "8 planners must share one SDF texture upload"
    handles = {p.model.model.handle(25).value for p in planners}
    assert len(handles) == 1
    assert len({p.guide._grids.data_ptr() for p in planners}) == 1
    print(f"8 MPD constructors + first calls: {dt * 1e3:.0f} ms")
    parity_log.record("device_side_sharing", "8_planners_construct_and_call_ms", None, dt * 1e3)
    del planners, outs


def test_torch_ops_match_the_ctypes_path():
    """torch.ops.mmd_amd.{unet_forward, guide_steps, p_sample_loop, ddim_sample} (torch.library custom ops over the same
    C ABI) are bit-identical to the ctypes host mirror, run on the current stream, and capture into a HIP graph."""
    import gpu_common
    from mmd_amd import ops
    from mmd_amd.diffusion_model import ddpm_sample_fn
    T, B, R = 25, 8, 2
    model = gpu_common.hip_model(T)
    starts, goals = synth.start_goal_circle(6, 0.8)
    paths = synth.straight_line_paths(starts, goals, H)
    guide = gpu_common.hip_guide("EnvHighways2D", [[cases.soft_group(paths, r)] for r in range(R)], n_robots=R)
    hc = {0: torch.stack([cases.hard_conds_for(starts[r], goals[r])[0] for r in range(R)]),
          H - 1: torch.stack([cases.hard_conds_for(starts[r], goals[r])[H - 1] for r in range(R)])}
    hard = torch.stack([hc[0], hc[H - 1]], dim=1).cuda().contiguous()
    tm, tu, tg = ops.register(model), ops.register(model.model), ops.register(guide)
    x = torch.from_numpy(synth.synth_noise(120, (R * B, H, D))).cuda()
    assert torch.equal(torch.ops.mmd_amd.unet_forward(x, 7, T, tu), model.model(x, 7))
    y1, y2 = x.clone(), x.clone()
    torch.ops.mmd_amd.guide_steps(y1, hard, SG, 5, tg)
    guide.guide_steps(y2, hard, _lib.HARD_ROWS_START_GOAL, 5)
    assert torch.equal(y1, y2)
    ref = model.run_inference(None, hc, n_samples=B, n_robots=R, horizon=H, return_chain=True, sample_fn=ddpm_sample_fn,
                              guide=guide, n_guide_steps=20, t_start_guide=13, noise_std_extra_schedule_fn=lambda t: 0.5,
                              n_diffusion_steps_without_noise=1, seed=77)
    xo = torch.empty((R * B, H, D), device="cuda")
    chain = torch.ops.mmd_amd.p_sample_loop(xo, hard, SG, tm, tg, R, T, 1, True, None, 77, 20, 13, 0.5, 0, True)
    assert torch.equal(chain, ref) and torch.equal(xo, ref[-1])
    refd, chd = model.ddim_sample((R * B, H, D), hc, n_diffusion_steps=T, return_chain=True, guide=guide,
                                  t_start_guide=13, n_robots=R, seed=78)
    xd = torch.empty((R * B, H, D), device="cuda")
    chain_d = torch.ops.mmd_amd.ddim_sample(xd, hard, SG, tm, tg, R, T, True, 78, 13, 0, True)
    assert torch.equal(chain_d, chd.transpose(0, 1)) and torch.equal(xd, refd)
    # a side stream: the op runs on torch's current stream
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        xs = torch.empty((R * B, H, D), device="cuda")
        cs = torch.ops.mmd_amd.p_sample_loop(xs, hard, SG, tm, tg, R, T, 1, True, None, 77, 20, 13, 0.5, 0, True)
    torch.cuda.current_stream().wait_stream(side)
    assert torch.equal(cs, ref)
    # stream capture: the whole 26-step guided loop as one hipGraph, replayed
    xg = torch.empty((R * B, H, D), device="cuda")
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        cg = torch.ops.mmd_amd.p_sample_loop(xg, hard, SG, tm, tg, R, T, 1, True, None, 77, 20, 13, 0.5, 0, True)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(cg, ref) and torch.equal(xg, ref[-1])


def test_plan_round_selection_vs_oracle_two_rounds():
    """MultiRobotSampler.plan_round as a composition (VERDICT r2 #5b): 6 robots on Highways, two planning rounds.  In each
    round the device-built soft-constraint table equals the oracle's (cbs.py:468-508), and the best-path pick of every
    robot -- free-first, then fewest robot-robot collisions against the gathered paths, first minimum; all samples if none
    is free (tasks.py:236-311, cbs.py:446-458) -- equals the oracle's selection on the same sampled batch."""
    import gpu_common
    from mmd_amd.multi_robot import MultiRobotSampler
    N, B, T = 6, 16, 25
    model = gpu_common.hip_model(T)
    starts, goals = synth.start_goal_circle(N, 0.45)
    s = MultiRobotSampler(model, starts, goals, env_id="EnvHighways2D", n_samples=B)
    gp = cases.guide_params("EnvHighways2D")
    paths = torch.from_numpy(synth.straight_line_paths(starts, goals, H)).cuda()
    n_nonfree = 0
    for rnd in range(2):
        paths_in = paths.clone()
        trajs, paths = s.plan_round(paths, seed=300 + rnd)
        # (a) the constraint table the round sampled with
        ell = s.guide._constraints()[0].cpu()
        for r in range(N):
            grp = O.soft_constraints_from_paths(paths_in.cpu(), r, 0.05 * 2.4, 2e-2)
            q = ell[r * (N - 1):(r + 1) * (N - 1), 1:, :2].reshape(-1, 2)          # slots x t >= 1
            assert torch.equal(q, grp.q.view(N - 1, H - 1, 2).reshape(-1, 2)), (rnd, r)
        # (b) the selection, on the sampled batch
        tu = s.unnormalize(trajs).cpu()
        idx = s.last_idx.cpu().tolist()
        for r in range(N):
            tr = tu[r * B:(r + 1) * B]
            _, _, _, free_idxs, _ = O.get_trajs_collision_and_free(tr, gp)
            free = free_idxs.reshape(-1).tolist()
            counts = O.count_collisions_with_others(tr[..., :2], paths_in.cpu(), r).tolist()
            cand = free if free else list(range(B))
            n_nonfree += B - len(free)
            want = min(cand, key=lambda b: (counts[b], b))
            assert idx[r] == want, (rnd, r, idx[r], want, free, counts)
            assert int(s.last_n_free[r]) == len(free)
            assert torch.equal(paths[r].cpu(), tr[want, :, :2])
    assert n_nonfree > 0, "the case must exercise the free / colliding split"


def test_extra_objects_guide_and_occupancy_g14():
    """A map WITH extra objects (env_base.py:76-89, mpd.py:215-233): the HIP guide evaluates the env's extra
    spheres / boxes analytically next to the fixed-object grid -- full guide and use_guide_on_extra_objects_only against the
    reference (g14, max-abs 2e-6), the task facade's compute_collision of random points exactly."""
    import gpu_common
    from mmd_amd.guides import GuideManagerTrajectoriesWithVelocity
    from mmd_amd.planners import PlanningTaskFacade, RobotPlanarDiskFacade
    g = np.load(os.path.join(GOLDEN, "g14_extra_objects.npz"))
    xo = {"spheres": g["spheres"].tolist(), "boxes": g["boxes"].tolist()}
    x = (torch.from_numpy(synth.synth_noise(95, (8, H, D))) * 0.6).cuda()
    full = GuideManagerTrajectoriesWithVelocity(gpu_common.dataset(), env_id="EnvHighways2D", extra_objects=xo, device="cuda")
    e1 = float((full(x).cpu() - torch.from_numpy(g["guide_full"])).abs().max())
    only = GuideManagerTrajectoriesWithVelocity(gpu_common.dataset(), env_id="EnvHighways2D", extra_objects=xo,
                                                extra_objects_only=True, device="cuda")
    e2 = float((only(x).cpu() - torch.from_numpy(g["guide_extra_only"])).abs().max())
    parity_log.record("extra_objects_g14", "guide_full", None, e1, bound=2e-6)
    parity_log.record("extra_objects_g14", "guide_extra_only", None, e2, bound=2e-6)
    assert e1 < 2e-6 and e2 < 2e-6, (e1, e2)
    # without the extra objects the same batch gets a different gradient (the objects matter)
    plain = GuideManagerTrajectoriesWithVelocity(gpu_common.dataset(), env_id="EnvHighways2D", device="cuda")
    assert float((plain(x).cpu() - torch.from_numpy(g["guide_full"])).abs().max()) > 1e-3
    task = PlanningTaskFacade(full, RobotPlanarDiskFacade(torch.device("cuda")))
    coll = task.compute_collision(torch.from_numpy(g["points"]))
    assert np.array_equal(coll.cpu().numpy().reshape(-1), g["coll_random"].reshape(-1))



def test_plan_concurrently_equals_the_sequential_calls():
    """planners.plan_concurrently: independent planner calls on one host thread + one stream each.  (a) three MPD planners of the
    Highways case with constraints: the default seeds are the ones the sequential loop draws, so the outputs are bitwise those of the
    loop; (b) the four MPDEnsemble planners of BASELINE's config 4 under explicit seeds; (c) re-plans from an experience (the forward
    noising draws under the call's seed too); a planner listed twice is refused."""
    from mmd_amd import diffusion_model as dm
    from mmd_amd.constraints import MultiPointConstraint
    from mmd_amd.planners import MPD, MPDEnsemble, PathBatchExperience, plan_concurrently
    starts, goals = synth.start_goal_circle(10, 0.45)
    paths = synth.straight_line_paths(starts, goals, H)

    def soft_for(r):
        c = MultiPointConstraint(q_l=[torch.from_numpy(paths[j, t]) for j in range(10) if j != r for t in range(1, H)],
                                 t_range_l=[(t, t + 1) for j in range(10) if j != r for t in range(1, H)])
        c.is_soft = True
        return c
    robots = (1, 4, 7)
    ps = [MPD(start_state_pos=torch.from_numpy(starts[r]), goal_state_pos=torch.from_numpy(goals[r]), **_mpd_kwargs(seed=18 + r))
          for r in robots]
    calls = [(p, torch.from_numpy(starts[r]), torch.from_numpy(goals[r]), [soft_for(r)]) for p, r in zip(ps, robots)]
    draws = dm._GLOBAL_DRAWS
    seq = [c[0](*c[1:]) for c in calls]
    dm._GLOBAL_DRAWS = draws                                  # the same point of the global stream for the concurrent run
    con = plan_concurrently(calls)
    for a, b in zip(seq, con):
        assert torch.equal(a.trajs_iters, b.trajs_iters) and torch.equal(a.trajs_final, b.trajs_final)
        assert a.idx_best_traj == b.idx_best_traj and b.t_total > 0
    assert all(p.guide.extra_cost_l == [[]] for p in ps)
    assert not torch.equal(con[0].trajs_iters[-1], con[1].trajs_iters[-1])
    # (c) re-plans seeded by the previous batches
    seeds = [901, 902, 903]
    calls2 = [c + (PathBatchExperience(o.trajs_final),) for c, o in zip(calls, seq)]
    seq2 = [c[0](*c[1:], seed=s) for c, s in zip(calls2, seeds)]
    con2 = plan_concurrently(calls2, seeds=seeds)
    for a, b in zip(seq2, con2):
        assert a.trajs_iters.shape[0] == 5 and torch.equal(a.trajs_iters, b.trajs_iters)
    with pytest.raises(ValueError):
        plan_concurrently([calls[0], calls[0]])
    # (b) config 4's ensemble planners
    sd = synth.synth_unet_state_dict(0)
    tr = {0: torch.tensor([0.0, 0.0]), 1: torch.tensor([2.0, 0.0])}
    ecalls = []
    for r in range(4):
        start, goal = torch.tensor([-0.7, -0.6 + 0.4 * r]), torch.tensor([2.7, 0.6 - 0.4 * r])
        ecalls.append((MPDEnsemble(model_ids=("EnvEmptyNoWait2D-RobotPlanarDisk",) * 2, transforms=tr, planner_alg="mmd",
                                   start_state_pos=start, goal_state_pos=goal, n_samples=8, model_state_dicts=[sd, sd],
                                   model_args=dict(n_diffusion_steps=25), device="cuda", seed=18 + r), start, goal))
    eseeds = [7001, 7002, 7003, 7004]
    eseq = [c[0](*c[1:], seed=s) for c, s in zip(ecalls, eseeds)]
    econ = plan_concurrently(ecalls, seeds=eseeds)
    for a, b in zip(eseq, econ):
        assert a.trajs_iters.shape[-2] == 2 * H and torch.equal(a.trajs_iters, b.trajs_iters)


def test_unnormalize_on_device_equals_the_torch_form_bitwise():
    """mmd_unnormalize_trajs (TrajectoryDatasetFacade.unnormalize_trajectories on CUDA tensors) against LimitsNormalizer.unnormalize's torch
    form (normalization.py:157-168) on the CPU: values inside [-1, 1], values in (1, 1 + 1e-4] (NOT clipped: they stay above the limit),
    one element beyond 1 + 1e-4 anywhere in the tensor (the WHOLE tensor is clipped), and the lower side; a [T+2, B, H, 4] chain shape."""
    from mmd_amd.normalization import TrajectoryDatasetFacade
    ds = TrajectoryDatasetFacade(np.array([-1.0, -1.1, -1.5, -1.7], np.float32), np.array([1.0, 0.9, 1.5, 1.3], np.float32))
    base = torch.from_numpy(synth.synth_noise(140, (27, 8, H, D))).clamp(-3, 3) / 3.0
    cases_ = {"inside": base.clone()}
    v = base.clone(); v[3, 2, 10, 1] = 1.00005; cases_["within_eps"] = v
    v = base.clone(); v[26, 7, 63, 3] = 1.0002; v[0, 0, 0, 0] = 0.99999; cases_["one_above"] = v
    v = base.clone(); v[5, 5, 5, 2] = -1.0002; cases_["one_below"] = v
    v = base.clone() * 2.5; cases_["many_outside"] = v
    for name, x in cases_.items():
        ref = ds.unnormalize_trajectories(x.clone())                      # the torch form on the CPU
        got = ds.unnormalize_trajectories(x.cuda()).cpu()
        assert torch.equal(got, ref), name
    assert float(ds.unnormalize_trajectories(cases_["within_eps"].cuda())[3, 2, 10, 1]) > 0.9       # above the limit: not clipped
    assert float(ds.unnormalize_trajectories(cases_["one_above"].cuda()).max()) <= 1.5
    # the chains of several planner calls batched robot-major (plan_batched): every call keeps its OWN clip decision
    x = torch.cat([cases_["inside"], cases_["one_above"], cases_["within_eps"], cases_["one_below"]], dim=1)       # [27, 4*8, H, D]
    ref = torch.cat([ds.unnormalize_trajectories(c.clone()) for c in x.chunk(4, dim=1)], dim=1)
    assert torch.equal(ds.unnormalize_trajectories(x.cuda(), n_tensors=4).cpu(), ref)
    assert torch.equal(ds.unnormalize_trajectories(x.clone(), n_tensors=4), ref)                              # the CPU form of the same
    assert not torch.equal(ds.unnormalize_trajectories(x.cuda()).cpu(), ref)            # one joint decision would clip call 2's 1.00005
