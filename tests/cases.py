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


LIN = 3.0     # the kernel's per-step deviation from the fp32 reference in units of the calibration perturbation (1e-6): FIXED


def chaos_bounds(errs, sens, n_unguided_rows):
    """Per-row bounds of an end-to-end guided chain.  `sens` is the reference's own response to a relative 1e-6 perturbation
    of its UNet output (max over 24 draws, stored with the golden rows).  The kernel's deviation from the reference is not
    exactly that size: its forward differs from the fp32 reference by <= 2.4e-6 rel-L2 (bounded against fp64 by
    test_unet_forward_accuracy_against_fp64), i.e. by at most LIN = 3 calibration units, so a chaotic row may be LIN times
    further away than the reference is from its perturbed self -- times 1.5 for the heavy tail of the amplification -- or within
    the north-star 1e-3.  The bound does NOT depend on the measured errors; the measured per-step deviation `lin` (max err / sens
    over the well-conditioned rows before guidance starts) is returned for the caller to assert lin < LIN (measured 1.0 .. 2.3)."""
    lin = 1.0
    for r in range(min(n_unguided_rows, len(errs))):
        if sens[r] > 0:
            lin = max(lin, errs[r] / sens[r])
    return lin, [max(1e-3, 1.5 * LIN * float(v)) for v in sens]


# ---- distribution-level parity (g15): the reference's guided sampler over many noise seeds ------------------------------------
DIST_CASES = ("empty32_T25", "highways_T100")
Z_MAX = 4.0            # every z-test below: |difference| <= Z_MAX standard errors (two independent sample sets assumed: conservative,
                       # the two samplers see the SAME noise and agree far better than independent draws would)


def distribution_inputs(T, B, n_seeds, base, first=0):
    """x_T [n B, H, D] and step noise [T + 1, n B, H, D] of seeds first .. first + n_seeds - 1 (tools/make_golden.py::g15)."""
    xT = torch.cat([torch.from_numpy(synth.synth_noise(base + 2 * j, (B, H, D))) for j in range(first, first + n_seeds)], 0)
    steps = torch.cat([torch.from_numpy(synth.synth_noise(base + 2 * j + 1, (T + 1, B, H, D))) for j in range(first, first + n_seeds)], 1)
    return xT, steps


def unnormalize(x):
    return (torch.clip(x, -1, 1) + 1) / 2 * (MAXS - MINS) + MINS


def violation_counts(pos, groups):
    """pos [n, H, 2] un-normalised positions; per trajectory the number of (constraint point, support point) pairs inside the
    radius at an active time t0 <= t < t1 (cost_functions.py:305)."""
    pos = np.asarray(pos, np.float64)
    out = np.zeros(pos.shape[0], np.int64)
    for g in groups:
        q, tr, rad = g.q.numpy().astype(np.float64), g.t_range.numpy().astype(np.int64), g.radius.numpy()
        for k in range(q.shape[0]):
            d = np.linalg.norm(pos[:, tr[k, 0]:tr[k, 1]] - q[k][None, None], axis=-1)
            out += (d < float(rad[k])).sum(1)
    return out


def position_stats(pos):
    pos = np.asarray(pos, np.float64)
    mean = pos.mean(0)
    dev = pos - mean[None]
    return mean, np.einsum("nhi,nhj->hij", dev, dev) / (pos.shape[0] - 1)


def distribution_z(pos_a, pos_b):
    """Largest z over the per-support-point position means and covariance entries of two sample sets [n, H, 2] (rows 0 / H-1 are
    hard-conditioned: zero variance, skipped).  SE(mean) = sqrt((va + vb) / n), SE(cov_ij) ~ sqrt((v_i v_j + c_ij^2) / (n - 1))
    per set (Gaussian approximation), combined in quadrature."""
    n = pos_a.shape[0]
    ma, ca = position_stats(pos_a)
    mb, cb = position_stats(pos_b)
    inner = slice(1, pos_a.shape[1] - 1)
    va, vb = np.diagonal(ca, axis1=1, axis2=2), np.diagonal(cb, axis1=1, axis2=2)
    z_mean = np.abs(ma - mb)[inner] / np.sqrt((va + vb)[inner] / n + 1e-30)

    def se_cov(c, v):
        return np.sqrt((v[:, :, None] * v[:, None, :] + c * c) / (n - 1))
    z_cov = np.abs(ca - cb)[inner] / np.sqrt(se_cov(ca, va)[inner] ** 2 + se_cov(cb, vb)[inner] ** 2 + 1e-30)
    return float(z_mean.max()), float(z_cov.max())


def proportion_z(ka, kb, n):
    """two-proportion z of ka / n against kb / n (0 when both are 0 or both n)."""
    p = (ka + kb) / (2.0 * n)
    if p <= 0.0 or p >= 1.0:
        return 0.0
    return abs(ka - kb) / n / np.sqrt(p * (1 - p) * 2.0 / n)


def mean_z(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    se = np.sqrt(a.var(ddof=1) / len(a) + b.var(ddof=1) / len(b))
    return 0.0 if se == 0 else float(abs(a.mean() - b.mean()) / se)


# ---- g19: a network that denoises (the reference's TemporalUnet after a fixed number of Adam steps with the reference's loss) -----
TRAINED_CASES = ("empty32", "highways")


def trained_state_dict():
    """key -> float32 ndarray in unet_param_spec order (tests/golden/g19_trained_unet.npz: data only)."""
    from collections import OrderedDict
    from mmd_amd.unet_spec import unet_param_spec
    g = np.load(os.path.join(GOLDEN, "g19_trained_unet.npz"))
    return OrderedDict((k, np.ascontiguousarray(g[k], dtype=np.float32)) for k in unet_param_spec())


def trained_case(name):
    """The two constraint cases of g19 (tools/make_golden.py::g19), T = 25: the 32-robot north-star shape (Empty, robot 5, 31 x 63
    soft points, B = 4) and Highways robot 3 with soft + hard constraints (B = 8); noise seeds seed0 + 2 s (x_T), + 1 (steps)."""
    if name == "empty32":
        starts, goals = synth.start_goal_circle(32, 0.8)
        paths = synth.straight_line_paths(starts, goals, H)
        return dict(map="EnvEmpty2D", T=25, B=4, start=starts[5], goal=goals[5], cons=[soft_group(paths, 5)], seed0=400)
    starts, goals, soft, hard = highways_case()
    return dict(map="EnvHighways2D", T=25, B=8, start=starts[3], goal=goals[3], cons=[soft, hard], seed0=440)


# ---- g20: 3-tile corner-turning heterogeneous ensemble (synth.ensemble3_case) -----------------------------------------------------
ENSEMBLE3_DIRECTIONS = ("fwd", "rev")


def named_state_dict(w):
    """tile weights of synth.ensemble3_case: a synth seed, or "g19" = the trained state dict stored as data by golden g19."""
    return trained_state_dict() if w == "g19" else synth.synth_unet_state_dict(w)


def ensemble3_hard_conds(case):
    """mpd_ensemble.py:286-296: start pinned on row 0 of the first tile, goal on the last row of the last tile, both in the tile
    frame (tasks_ensemble.inverse_transform_q) and normalised."""
    K = len(case["env_ids"])
    s = hard_conds_for(case["start"] - case["transforms"][0], [0, 0])[0]
    g = hard_conds_for(case["goal"] - case["transforms"][K - 1], [0, 0])[0]
    hard = {m: {} for m in range(K)}
    hard[0][0] = s
    hard[K - 1][H - 1] = g
    return hard


def ensemble3_tile_groups(g, direction, K=3):
    """The reference's own per-tile constraint tables of golden g20 (after split_cost_constraints_to_tasks + the tile shift) as
    oracle ConstraintGroups, in the order the tile guides received them; weights mmd_params.py:42-43."""
    out = {m: [] for m in range(K)}
    for m in g[f"{direction}.task_order"].tolist():
        for k in range(int(g[f"{direction}.n_{m}"])):
            soft = bool(g[f"{direction}.soft_{m}_{k}"])
            out[m].append(O.ConstraintGroup(q=torch.from_numpy(g[f"{direction}.qs_{m}_{k}"]),
                                            t_range=torch.from_numpy(g[f"{direction}.ranges_{m}_{k}"]),
                                            radius=torch.from_numpy(g[f"{direction}.radii_{m}_{k}"]),
                                            weight=2e-2 if soft else 2e-1))
    return out


def ensemble3_inputs(case, T, B):
    K = len(case["env_ids"])
    x0 = {m: torch.from_numpy(synth.synth_noise(case["seeds"]["x0"][m], (B, H, D))) for m in range(K)}
    steps = torch.from_numpy(synth.synth_noise(case["seeds"]["steps"], (T + 1, K, B, H, D)))
    return x0, steps


def ensemble3_local_inputs(case, g, direction, B, K, n_denoise):
    """golden g22's inputs: the stored seed batch (data), the q_sample draw and the step noise regenerated from their seeds."""
    base = case["seeds"]["steps"]
    seed = torch.from_numpy(g[f"{direction}.seed"])
    qn = torch.from_numpy(synth.synth_noise(base + 21, (B, K * H, D)))
    steps = torch.from_numpy(synth.synth_noise(base + 22, (n_denoise + 1, K, B, H, D)))
    return seed, qn, steps
