"""Synthetic, bit-reproducible inputs (numpy PCG64) shared by the golden generator, the tests and bench.py.

There are no datasets / checkpoints offline, so UNet weights are random-init-like, noise is
injected, and normaliser limits are the builder's choice (pos [-1,1]^2, vel [-1.5,1.5]^2).  Everything is
drawn from `numpy.random.Generator(PCG64(seed))`, which is identical on every machine.
"""
import math
import numpy as np

from .unet_spec import unet_param_spec

NORM_MINS = np.array([-1.0, -1.0, -1.5, -1.5], dtype=np.float32)
NORM_MAXS = np.array([1.0, 1.0, 1.5, 1.5], dtype=np.float32)


def synth_unet_state_dict(seed=0, state_dim=4, unet_input_dim=32, dim_mults=(1, 2, 4)):
    """dict key -> float32 ndarray.  torch-default-init-like scales; GroupNorm affine perturbed so it is exercised."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for key, shape in unet_param_spec(state_dim, unet_input_dim, dim_mults).items():
        is_gn = ".block.2." in key
        if is_gn:
            if key.endswith("weight"):
                v = 1.0 + 0.1 * rng.standard_normal(shape)
            else:
                v = 0.1 * rng.standard_normal(shape)
        else:
            wkey = key.rsplit(".", 1)[0] + ".weight"
            wshape = unet_param_spec(state_dim, unet_input_dim, dim_mults)[wkey]
            if key.startswith("ups.") and ".4.conv." in key:
                fan_in = wshape[1] * wshape[2]     # ConvTranspose1d: fan_in computed on dim 1 by torch
            else:
                fan_in = int(np.prod(wshape[1:]))
            b = 1.0 / math.sqrt(fan_in)
            v = rng.uniform(-b, b, size=shape)
        sd[key] = np.ascontiguousarray(v, dtype=np.float32)
    return sd


def synth_noise(seed, shape):
    """Standard normal float32 of `shape` (drawn in float64, rounded once)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.standard_normal(shape).astype(np.float32)


def start_goal_circle(num_agents, radius=0.8):
    """Circle formation with antipodal goals (reference mmd/common/multi_agent_utils.py:146-154), float32 [N,2] each."""
    starts = np.array([[radius * np.cos(2 * np.pi * i / num_agents), radius * np.sin(2 * np.pi * i / num_agents)]
                       for i in range(num_agents)], dtype=np.float32)
    goals = np.array([[radius * np.cos(2 * np.pi * i / num_agents + np.pi),
                       radius * np.sin(2 * np.pi * i / num_agents + np.pi)]
                      for i in range(num_agents)], dtype=np.float32)
    return starts, goals


def start_goal_boundary(num_agents, dist=0.87):
    """Boundary formation (reference mmd/common/multi_agent_utils.py:157-173)."""
    starts = np.array([[0.8 * np.cos(2 * np.pi * i / num_agents), 0.8 * np.sin(2 * np.pi * i / num_agents)]
                       for i in range(num_agents)], dtype=np.float32)
    for i in range(num_agents):
        if abs(starts[i, 0]) > abs(starts[i, 1]):
            starts[i, 0] = np.sign(starts[i, 0]) * dist
        else:
            starts[i, 1] = np.sign(starts[i, 1]) * dist
    goals = np.array([[starts[i, 0] if abs(starts[i, 0]) < abs(starts[i, 1]) else -starts[i, 0],
                       starts[i, 1] if abs(starts[i, 1]) < abs(starts[i, 0]) else -starts[i, 1]]
                      for i in range(num_agents)], dtype=np.float32)
    return starts, goals


def straight_line_paths(starts, goals, horizon=64):
    """[N,H,2] float32 straight-line position paths: the synthetic stand-in for 'previous best paths' (SURVEY §8d-3)."""
    a = np.linspace(0.0, 1.0, horizon, dtype=np.float32)[None, :, None]
    return (starts[:, None, :] * (1 - a) + goals[:, None, :] * a).astype(np.float32)


# ---- 3-tile corner-turning MPDEnsemble instance (golden g20) -------------------------------------------------------------
ENSEMBLE3_GRID = (("EnvHighways2D", "EnvDropRegion2D"), ("EnvEmptyNoWait2D", "EnvConveyor2D"))   # global_model_ids[row][col]
ENSEMBLE3_RADIUS = 0.05 * 2.4                                                                    # mmd_params.py:52


def ensemble3_case(direction="fwd", horizon=64):
    """A multi-tile planning problem shaped like the reference's multi-tile experiments (mmd_experiment_configs.py:180-221: 3-step
    skeletons that turn a corner over a grid of heterogeneous tile models): skeleton [[0,0],[0,1],[1,1]] ("fwd": one +x hop, then
    one -y hop) or the same tiles traversed the other way ("rev": +y, then -x, so both relative transforms of
    apply_cross_conditioning are negative in the clamped coordinate), tile transforms [col * 2, -row * 2] as
    scripts/inference/inference_multi_agent.py:148-151 builds them, every tile with its own map (an SDF gradient in each).
    Returns a dict: skeleton, env_ids [K], transforms [K,2], start / goal (GLOBAL frame, inference_multi_agent.py:196-199) and
    `constraints`: MultiPointConstraint-shaped tuples (q [n,2] global frame, t_range [n,2] over the K*H support points, radius
    [n], is_soft) -- one hard and one soft constraint whose points fall on different tiles (one soft range straddles a tile
    boundary and stays with the tile of its first index), for MPDEnsemble.split_cost_constraints_to_tasks
    (mpd_ensemble.py:431-522) to route."""
    skeleton = [[0, 0], [0, 1], [1, 1]]
    # (skeleton step, tile-local point, tile-local time range)
    hard = [(1, (-0.45, 0.0), (6, 12)), (2, (0.0, 0.45), (12, 18))]
    soft = [(0, (0.45, 0.0), (20, 21)), (1, (0.0, 0.45), (36, 37)), (2, (-0.45, 0.0), (42, 43)), (0, (0.9, 0.0), (62, 66))]
    start_local, goal_local = np.array([-0.6, 0.5], np.float32), np.array([0.5, -0.6], np.float32)
    if direction == "rev":
        skeleton = skeleton[::-1]
        start_local, goal_local = goal_local, start_local
        hard = [(2 - s, q, t) for (s, q, t) in hard]
        soft = [(2 - s, q, t) for (s, q, t) in soft[:3]] + [(0, (0.0, 0.9), (62, 66))]
    elif direction != "fwd":
        raise KeyError(direction)
    transforms = np.array([[c * 2.0, -r * 2.0] for r, c in skeleton], np.float32)
    env_ids = [ENSEMBLE3_GRID[r][c] for r, c in skeleton]

    def to_global(items, is_soft):
        q = np.array([np.asarray(p, np.float32) + transforms[s] for s, p, _ in items], np.float32)
        tr = np.array([(s * horizon + t0, s * horizon + t1) for s, _, (t0, t1) in items], np.int64)
        return q, tr, np.full(len(items), ENSEMBLE3_RADIUS, np.float32), is_soft
    return dict(skeleton=skeleton, env_ids=env_ids, transforms=transforms, start=start_local + transforms[0],
                goal=goal_local + transforms[-1], constraints=[to_global(hard, False), to_global(soft, True)],
                # tile UNets: three different random-init weight sets / the briefly trained one of g19 (it denoises, so the
                # trajectories are smooth and actually meet the constraint points) on every tile
                weights=(0, 1, 2) if direction == "fwd" else ("g19", "g19", "g19"),
                seeds=dict(x0=(200, 201, 202), steps=203) if direction == "fwd" else dict(x0=(210, 211, 212), steps=213))
