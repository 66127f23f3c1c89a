"""-m gpu: the outer boundary (MPD / MPDEnsemble / DiffusionsEnsemble) over the HIP kernels."""
import os
from math import ceil

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mmd_amd import synth                # noqa: E402
import cases                             # noqa: E402
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
    # row T//2+1 is the state after the FIRST guided step of both tiles: tight; finals: chaotic regime
    assert rel_l2(chains[0][:, T // 2 + 1].cpu(), g["chain0_mid"]) < 2e-3
    assert rel_l2(chains[1][:, T // 2 + 1].cpu(), g["chain1_mid"]) < 2e-3
    assert rel_l2(x[0].cpu(), g["final0"]) < 0.3 and rel_l2(x[1].cpu(), g["final1"]) < 0.3


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
