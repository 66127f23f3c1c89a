"""Input construction for the golden cases (mirrors tools/make_golden.py WITHOUT importing the reference).

Inputs are regenerated from seeds (mmd_amd.synth); the expected outputs are in tests/golden/*.npz.
"""
import os
from math import ceil

import numpy as np
import torch

from mmd_amd import synth
from oracle import mmd_oracle as O

H, D = 64, 4
RADIUS_SOFT = 0.05 * 2.4
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MINS, MAXS = torch.from_numpy(synth.NORM_MINS), torch.from_numpy(synth.NORM_MAXS)

_GRIDS = {}


def sdf_grid(map_name):
    if map_name not in _GRIDS:
        _GRIDS[map_name] = O.build_sdf_grid(map_name)
    return _GRIDS[map_name]


def guide_params(map_name, cutoff=0.05):
    return O.GuideParams(norm_mins=MINS, norm_maxs=MAXS, sdf_grids=[sdf_grid(map_name)], cutoff_margin=cutoff)


def hard_conds_for(start, goal):
    s = O.normalize(torch.tensor([start[0], start[1], 0.0, 0.0]), MINS, MAXS)
    g = O.normalize(torch.tensor([goal[0], goal[1], 0.0, 0.0]), MINS, MAXS)
    return {0: s, H - 1: g}


def soft_group(paths, agent, weight=2e-2):
    return O.soft_constraints_from_paths(torch.from_numpy(paths), agent, RADIUS_SOFT, weight)


def hard_group(q, tr, weight=2e-1):
    q = torch.tensor(q, dtype=torch.float32).reshape(-1, 2)
    return O.ConstraintGroup(q=q, t_range=torch.tensor(tr, dtype=torch.float32).reshape(-1, 2),
                             radius=torch.full((q.shape[0],), RADIUS_SOFT), weight=weight)


def highways_case(agent=3, n_agents=10):
    starts, goals = synth.start_goal_circle(n_agents, 0.45)
    paths = synth.straight_line_paths(starts, goals, H)
    return starts, goals, soft_group(paths, agent), hard_group([[0.1, 0.2]], [[20, 27]])


# name -> (map, T, B, starts/goals fn, agent, constraint builder, seed_xT, seed_steps)
def sample_case(name):
    if name == "empty_T50":
        starts, goals = synth.start_goal_circle(6, 0.8)
        paths = synth.straight_line_paths(starts, goals, H)
        return dict(map="EnvEmpty2D", T=50, B=8, start=starts[0], goal=goals[0], cons=[soft_group(paths, 0)],
                    seeds=(11, 12))
    if name == "highways_T100":
        starts, goals, soft, hard = highways_case()
        return dict(map="EnvHighways2D", T=100, B=8, start=starts[3], goal=goals[3], cons=[soft, hard], seeds=(13, 14))
    if name == "empty_T25_nocons":
        starts, goals = synth.start_goal_circle(10, 0.45)
        return dict(map="EnvEmpty2D", T=25, B=4, start=starts[0], goal=goals[0], cons=[], seeds=(15, 16))
    if name == "cfg0_T50_B1":
        return dict(map="EnvEmpty2D", T=50, B=1, start=np.array([-0.8, 0], np.float32),
                    goal=np.array([0.8, 0], np.float32), cons=[], seeds=(17, 18))
    if name == "empty32_T25":
        starts, goals = synth.start_goal_circle(32, 0.8)
        paths = synth.straight_line_paths(starts, goals, H)
        return dict(map="EnvEmpty2D", T=25, B=4, start=starts[5], goal=goals[5], cons=[soft_group(paths, 5)],
                    seeds=(19, 20))
    if name == "conveyor_T50":
        starts, goals = synth.start_goal_boundary(8)
        paths = synth.straight_line_paths(starts, goals, H)
        return dict(map="EnvConveyor2D", T=50, B=4, start=starts[2], goal=goals[2], cons=[soft_group(paths, 2)],
                    seeds=(21, 22))
    if name == "prior_T100":
        starts, goals = synth.start_goal_circle(6, 0.8)
        return dict(map="EnvEmpty2D", T=100, B=8, start=starts[1], goal=goals[1], cons=[], seeds=(29, 30),
                    use_guide=False)
    raise KeyError(name)


def ddim_case(name):
    """DDIM golden cases (g11): name -> dict like sample_case plus the x_T seed."""
    if name == "empty_T50":
        starts, goals = synth.start_goal_circle(6, 0.8)
        return dict(map="EnvEmpty2D", T=50, B=8, start=starts[0], goal=goals[0], cons=[], seed=31, use_guide=False)
    if name == "highways_T100":
        starts, goals, soft, hard = highways_case()
        return dict(map="EnvHighways2D", T=100, B=8, start=starts[3], goal=goals[3], cons=[soft, hard], seed=32)
    raise KeyError(name)


DDIM_CASES = ("empty_T50", "highways_T100")


def oracle_ddim(case, weights_seed=0, clip_mode="reference"):
    sd = O.state_dict_to_torch(synth.synth_unet_state_dict(weights_seed))
    tb = O.schedule_tables(case["T"])
    gp = guide_params(case["map"])
    xT = torch.from_numpy(synth.synth_noise(case["seed"], (case["B"], H, D)))
    guide = (lambda x: O.guide_grad(x, gp, case["cons"], clip_mode=clip_mode)) if case.get("use_guide", True) else None
    return O.ddim_sample(sd, tb, xT, hard_conds_for(case["start"], case["goal"]), case["T"], guide=guide,
                         t_start_guide=ceil(0.5 * case["T"]))


SAMPLE_CASES = ("empty_T50", "highways_T100", "empty_T25_nocons", "cfg0_T50_B1", "empty32_T25", "conveyor_T50",
                "prior_T100")


def sample_inputs(case):
    T, B = case["T"], case["B"]
    xT = torch.from_numpy(synth.synth_noise(case["seeds"][0], (B, H, D)))
    steps = torch.from_numpy(synth.synth_noise(case["seeds"][1], (T + 1, B, H, D)))
    return xT, steps


def rel_l2(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def oracle_run_inference(case, weights_seed=0, clip_mode="reference"):
    sd = O.state_dict_to_torch(synth.synth_unet_state_dict(weights_seed))
    tb = O.schedule_tables(case["T"])
    gp = guide_params(case["map"], case.get("cutoff", 0.05))
    xT, steps = sample_inputs(case)
    guide = lambda x: O.guide_grad(x, gp, case["cons"], clip_mode=clip_mode)   # noqa: E731
    if not case.get("use_guide", True):
        guide = None
    return O.p_sample_loop(sd, tb, xT, hard_conds_for(case["start"], case["goal"]), case["T"], steps, guide=guide,
                           n_guide_steps=20, t_start_guide=ceil(0.5 * case["T"]), noise_std_extra=0.5,
                           n_diffusion_steps_without_noise=1)


def chaos_bounds(errs, sens, n_unguided_rows):
    """Per-row bounds of an end-to-end guided chain.  `sens` is the reference's own response to a relative 1e-6 perturbation
    of its UNet output (max over 24 draws, stored with the golden rows).  The kernel's deviation from the reference is not
    exactly that size: on the rows BEFORE guidance starts the chain is well conditioned (errors ~1e-6, linear in the
    perturbation), so lin = max(1, max err / sens over those rows) measures the kernel's per-step deviation in units of the
    calibration perturbation, and a chaotic row may be lin times further away than the reference is from its perturbed
    self -- times SENS_FACTOR for the heavy tail of the amplification -- or within the north-star 1e-3."""
    lin = 1.0
    for r in range(min(n_unguided_rows, len(errs))):
        if sens[r] > 0:
            lin = max(lin, errs[r] / sens[r])
    return lin, [max(1e-3, 1.5 * lin * float(v)) for v in sens]
