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
