"""Helpers for the -m gpu parity tests: build the HIP-side objects for a golden case."""
from math import ceil

import numpy as np
import torch

from mmd_amd import synth
from mmd_amd.constraints import CostConstraint
from mmd_amd.diffusion_model import GaussianDiffusionModel, ddpm_sample_fn
from mmd_amd.guides import GuideManagerTrajectoriesWithVelocity
from mmd_amd.normalization import TrajectoryDatasetFacade
from mmd_amd.temporal_unet import TemporalUnet

_MODELS = {}


def hip_model(T, weights_seed=0):
    key = (T, weights_seed)
    if key not in _MODELS:
        unet = TemporalUnet(state_dim=4, n_support_points=64, unet_input_dim=32, dim_mults=(1, 2, 4))
        unet.load_state_dict(synth.synth_unet_state_dict(weights_seed))
        _MODELS[key] = GaussianDiffusionModel(model=unet, variance_schedule="exponential", n_diffusion_steps=T,
                                              predict_epsilon=True)
    return _MODELS[key]


def dataset():
    return TrajectoryDatasetFacade(synth.NORM_MINS, synth.NORM_MAXS)


def to_cost_constraint(grp):
    """oracle ConstraintGroup -> product CostConstraint holder + weight."""
    return CostConstraint(None, 64, q_l=[q for q in grp.q], traj_range_l=grp.t_range.numpy().tolist(),
                          radius_l=grp.radius.numpy().tolist(), is_soft=grp.weight < 0.1), grp.weight


def hip_guide(map_name, cons_per_robot, cutoff=0.05, n_robots=1, robot_env_ids=None):
    g = GuideManagerTrajectoriesWithVelocity(dataset(), env_id=map_name, obstacle_cutoff_margin=cutoff,
                                             n_robots=n_robots, robot_env_ids=robot_env_ids, device="cuda")
    for r, groups in enumerate(cons_per_robot):
        pairs = [to_cost_constraint(grp) for grp in groups]
        g.add_extra_costs([p[0] for p in pairs], [p[1] for p in pairs], robot=r)
    return g


def hip_run_inference(case, xT, steps, weights_seed=0):
    import cases
    model = hip_model(case["T"], weights_seed)
    guide = hip_guide(case["map"], [case["cons"]], case.get("cutoff", 0.05)) if case.get("use_guide", True) else None
    hc = cases.hard_conds_for(case["start"], case["goal"])
    return model.run_inference(None, hc, n_samples=case["B"], horizon=64, return_chain=True, sample_fn=ddpm_sample_fn,
                               guide=guide, n_guide_steps=20, t_start_guide=ceil(0.5 * case["T"]),
                               noise_std_extra_schedule_fn=lambda x: 0.5, n_diffusion_steps_without_noise=1,
                               warm_start_path_b=xT.cuda(), step_noise=steps.cuda())


def teacher_forced_guided_step(test, tag, s, T, starts, goals, map_name, picks, i, seeds, paths_np=None):
    """ONE guided ddpm_sample_fn step (sample_functions.py:40-107: UNet + posterior mean + 20 guide iterations + noise + hard
    conditioning) of MultiRobotSampler `s` on ALL its local trajectories, started from a mid-chain looking state, against the
    oracle on the trajectories `picks` = [(global robot, sample)].  Bound per trajectory: the north-star 1e-3, or 1.5 x the
    oracle's OWN response to a rounding-sized (relative 2e-6) perturbation of eps where that is larger -- 20 norm-clipped
    iterations over hinge constraints are not continuous in eps, so a trajectory on a switching surface moves by more than 1e-3
    whatever computes it (tests/test_gpu_fullsize.py uses the same yardstick).  paths_np: all robots' paths for the inter-robot
    soft constraints (None: no inter-robot term).  Returns the worst error / bound ratio."""
    import cases
    import parity_log
    from oracle import mmd_oracle as O
    H, D, B = 64, 4, s.n_samples
    n = s.n_local * B
    model = s.model
    x = torch.from_numpy(synth.synth_noise(seeds[0], (n, H, D))) * 0.5
    x[:, 0] = s.hard_conds[0].cpu().repeat_interleave(B, 0)
    x[:, -1] = s.hard_conds[H - 1].cpu().repeat_interleave(B, 0)
    noise = torch.from_numpy(synth.synth_noise(seeds[1], (n, H, D)))
    ys = x.clone().cuda()
    model.sample_step(ys, s.hard_conds, i, guide=s.guide, n_guide_steps=20, t_start_guide=ceil(0.5 * T),
                      noise_std_extra_schedule_fn=lambda t: 0.5, n_robots=s.n_local, noise=noise.cuda())
    ys = ys.cpu()
    assert torch.isfinite(ys).all()
    sd = O.state_dict_to_torch(synth.synth_unet_state_dict(0))
    tb = O.schedule_tables(T)
    gp = cases.guide_params(map_name)
    worst = 0.0
    for r, b in picks:
        idx = (r - s.robot0) * B + b
        groups = [cases.soft_group(paths_np, r)] if paths_np is not None else []
        hc = cases.hard_conds_for(starts[r], goals[r])
        guide = lambda z, groups=groups: O.guide_grad(z, gp, groups, clip_mode="always")      # noqa: E731
        xi = x[idx:idx + 1]
        step = lambda pert=None: O.apply_hard_conditioning(                                   # noqa: E731
            O.ddpm_sample_step(sd, tb, xi.clone(), hc, i, guide=guide, n_guide_steps=20, t_start_guide=ceil(0.5 * T),
                               noise=noise[idx:idx + 1], noise_std_extra=0.5, eps_rel_perturb=pert), hc)
        ref = step()
        err = cases.rel_l2(ys[idx:idx + 1], ref)
        gen = torch.Generator().manual_seed(1000 + idx)
        sens = max(cases.rel_l2(step(2e-6 * torch.randn(xi.shape, generator=gen)), ref) for _ in range(8))
        bound = max(1e-3, 1.5 * sens)
        parity_log.record(test, f"{tag}_robot{r}_sample{b}", i, err, sens=sens, bound=bound)
        assert err < bound, (tag, r, b, err, sens)
        worst = max(worst, err / bound)
    return worst
