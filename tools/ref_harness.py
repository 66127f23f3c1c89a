"""Builds the GENUINE reference objects (imported from /root/reference) the way `MPD.__init__` does
(reference mmd/planners/single_agent/mpd.py:124-265), without the dataset / checkpoint files that are not
available offline (SURVEY.md §8c recipe step 3).  Build-container only; never shipped to the GPU box.
"""
import contextlib
import sys
from math import ceil

import numpy as np
import torch

from ref_bootstrap import bootstrap

bootstrap()

from mmd.datasets.normalization import DatasetNormalizer                                    # noqa: E402
from mmd.models.diffusion_models.diffusion_model_base import GaussianDiffusionModel          # noqa: E402
from mmd.models.diffusion_models.guides import GuideManagerTrajectoriesWithVelocity          # noqa: E402
from mmd.models.diffusion_models.sample_functions import ddpm_sample_fn                      # noqa: E402
from mmd.models.diffusion_models.temporal_unet import TemporalUnet                           # noqa: E402
from mp_baselines.planners.costs.cost_functions import (CostCollision, CostComposite,       # noqa: E402
                                                        CostConstraint, CostGPTrajectory)
from torch_robotics import environments                                                     # noqa: E402
from torch_robotics.robots import RobotPlanarDisk                                           # noqa: E402
from torch_robotics.tasks.tasks import PlanningTask                                         # noqa: E402

TENSOR_ARGS = {"device": "cpu", "dtype": torch.float32}


class DatasetLike:
    """What GuideManagerTrajectoriesWithVelocity needs from TrajectoryDataset: (un)normalize_trajectories backed by
    the reference's own DatasetNormalizer/LimitsNormalizer (mmd/datasets/normalization.py:13-168)."""

    def __init__(self, mins, maxs):
        X = torch.stack([torch.as_tensor(mins), torch.as_tensor(maxs)]).float()[None]     # [1,2,D] 'b h d'
        self.normalizer = DatasetNormalizer({"traj": X}, "LimitsNormalizer")

    def unnormalize_trajectories(self, x):
        return self.normalizer.unnormalize(x, "traj")

    def normalize_trajectories(self, x):
        return self.normalizer.normalize(x, "traj")


_ENV_CACHE = {}


def make_task(env_id, cutoff_margin=0.05):
    key = (env_id, cutoff_margin)
    if key not in _ENV_CACHE:
        env = getattr(environments, env_id + "ExtraObjects")(tensor_args=TENSOR_ARGS)     # use_extra_objects=True, mpd.py:126
        robot = RobotPlanarDisk(tensor_args=TENSOR_ARGS)
        task = PlanningTask(env=env, robot=robot, obstacle_cutoff_margin=cutoff_margin, tensor_args=TENSOR_ARGS)
        _ENV_CACHE[key] = (env, robot, task)
    return _ENV_CACHE[key]


def make_model(sd_np, T, dim_mults=(1, 2, 4), unet_input_dim=32):
    unet = TemporalUnet(state_dim=4, n_support_points=64, unet_input_dim=unet_input_dim, dim_mults=dim_mults)
    model = GaussianDiffusionModel(model=unet, variance_schedule="exponential", n_diffusion_steps=T,
                                   predict_epsilon=True)
    ref_keys = [k[len("model."):] for k in model.state_dict().keys() if k.startswith("model.")]
    assert ref_keys == list(sd_np.keys()), "unet_param_spec order differs from the reference state_dict"
    missing = model.load_state_dict({"model." + k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=False)
    assert not missing.unexpected_keys and all(not k.startswith("model.") for k in missing.missing_keys)
    model.eval()
    for p in model.parameters():
        p.requires_grad_(False)
    return model


def make_guide(env_id, mins, maxs, cutoff_margin=0.05, which=("coll", "gp"), n_support_points=64,
               trajectory_duration=5.0, w_coll=2e-2, w_smooth=8e-2):
    """cost/guide assembly of mpd.py:209-265.  `which` lets the golden generator isolate cost terms."""
    env, robot, task = make_task(env_id, cutoff_margin)
    dt = trajectory_duration / n_support_points
    robot.dt = dt
    cost_l, w_l = [], []
    if "coll" in which or "obj" in which or "ws" in which:
        for fi, fld in enumerate(task.get_collision_fields()):
            if "coll" not in which and ((fi == 1 and "obj" not in which) or (fi == 2 and "ws" not in which) or fi == 0):
                continue
            cost_l.append(CostCollision(robot, n_support_points, field=fld, sigma_coll=1.0, tensor_args=TENSOR_ARGS))
            w_l.append(w_coll)
    if "gp" in which:
        cost_l.append(CostGPTrajectory(robot, n_support_points, dt, sigma_gp=1.0, tensor_args=TENSOR_ARGS))
        w_l.append(w_smooth)
    comp = CostComposite(robot, n_support_points, cost_l, weights_cost_l=w_l, tensor_args=TENSOR_ARGS)
    guide = GuideManagerTrajectoriesWithVelocity(
        DatasetLike(mins, maxs), comp, clip_grad=True, interpolate_trajectories_for_collision=True,
        num_interpolated_points=ceil(n_support_points * 1.5), tensor_args=TENSOR_ARGS)
    return guide, robot, task, env


def make_cost_constraint(robot, q, t_range, radius, is_soft, n_support_points=64):
    """CostConstraint exactly as MPD.__call__ builds it (mpd.py:329-342)."""
    return CostConstraint(robot, n_support_points, q_l=[torch.as_tensor(v) for v in q],
                          traj_range_l=[tuple(int(a) for a in tr) for tr in t_range],
                          radius_l=[float(r) for r in radius], is_soft=is_soft, tensor_args=TENSOR_ARGS)


@contextlib.contextmanager
def injected_noise(draws):
    """Replace torch.randn / torch.randn_like by a FIFO of pre-drawn tensors (call order = reference order)."""
    q = [torch.as_tensor(d) for d in draws]
    orig_randn, orig_like = torch.randn, torch.randn_like

    def _pop(shape):
        assert q, "reference drew more noise tensors than injected"
        z = q.pop(0)
        assert tuple(z.shape) == tuple(shape), (z.shape, shape)
        return z.clone()

    def randn(*size, **kw):
        shape = size[0] if len(size) == 1 and not isinstance(size[0], int) else size
        return _pop(tuple(shape))

    def randn_like(x, **kw):
        return _pop(tuple(x.shape))

    torch.randn, torch.randn_like = randn, randn_like
    try:
        yield q
    finally:
        torch.randn, torch.randn_like = orig_randn, orig_like


@contextlib.contextmanager
def quiet():
    class _N:
        def write(self, *_):
            pass

        def flush(self):
            pass
    old = sys.stdout
    sys.stdout = _N()
    try:
        yield
    finally:
        sys.stdout = old
