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
