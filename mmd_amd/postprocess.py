"""Post-sampling selection (the step right after the hot path): collision / free split, path length +
smoothness costs, SavGol smoothing.  torch ops on the device tensors (+ scipy for the SavGol filter, as the reference
does in mmd/common/trajectory_utils.py:31-51).  Restated from deps/torch_robotics/torch_robotics/tasks/tasks.py:236-311,
trajectory/metrics.py:7-39, trajectory/utils.py:73-86."""
import numpy as np
import torch

from .environments import LIMITS, sdf_grid_texture

ROBOT_RADIUS = 0.05


def interpolate_traj_via_points(trajs, num_interpolation=10):
    H, D = trajs.shape[-2:]
    if num_interpolation <= 0:
        return trajs
    alpha = torch.linspace(0, 1, num_interpolation + 2).type_as(trajs)[1:num_interpolation + 1]
    alpha = alpha.view((1,) * len(trajs.shape[:-1]) + (-1, 1))
    out = trajs[..., 0:H - 1, None, :] * alpha + trajs[..., 1:H, None, :] * (1 - alpha)
    return out.view(trajs.shape[:-2] + (-1, D))


def compute_collision(pos, env_id, margin=ROBOT_RADIUS):
    """occupancy-type check of PlanningTask._compute_collision_or_cost (tasks.py:141-234): a point collides iff the
    fixed-object SDF (nearest grid cell) or any workspace-boundary distance is below `margin`."""
    tex = torch.from_numpy(sdf_grid_texture(env_id)).to(pos.device)
    lo = torch.tensor(LIMITS[0], device=pos.device)
    hi = torch.tensor(LIMITS[1], device=pos.device)
    n = torch.tensor(tex.shape[:2], device=pos.device)
    idx = ((pos - lo) / (hi - lo).abs() * n).floor().long()
    idx = torch.minimum(torch.maximum(idx, torch.zeros_like(idx)), n - 1)
    sdf = tex[idx[..., 0], idx[..., 1], 0]
    ws = torch.cat((pos - lo * 1.08, hi * 1.08 - pos), dim=-1)
    return (sdf < margin) | (ws < margin).any(dim=-1)


def get_trajs_collision_and_free(trajs, env_id, num_interpolation=5, all_free=False):
    """tasks.py:236-311 for [B,H,D] batches.  Returns (coll, coll_idxs, free, free_idxs, waypoint_collisions)."""
    B = trajs.shape[0]
    if all_free:                                             # PlanningTaskEnsemble, tasks_ensemble.py:271-277
        coll_pts = torch.zeros(B, 1, dtype=torch.bool, device=trajs.device)
    else:
        coll_pts = compute_collision(interpolate_traj_via_points(trajs, num_interpolation)[..., :2], env_id)
    in_coll = coll_pts.any(dim=-1)
    pos = trajs[..., :2]
    inside = ((pos >= torch.tensor(LIMITS[0], device=pos.device)) & (pos <= torch.tensor(LIMITS[1], device=pos.device)))
    free_mask = ~in_coll & inside.all(dim=-1).all(dim=-1)
    if all_free:
        free_mask = torch.ones_like(in_coll)
    free_idxs = torch.argwhere(free_mask)
    coll_idxs = torch.argwhere(~free_mask)
    free = trajs[free_mask] if free_mask.any() else None
    coll = trajs[~free_mask] if (~free_mask).any() else None
    return coll, coll_idxs, free, free_idxs, coll_pts


def compute_path_length(trajs):
    return torch.linalg.norm(torch.diff(trajs[..., :2], dim=-2), dim=-1).sum(-1)


def compute_smoothness(trajs):
    return torch.linalg.norm(torch.diff(trajs[..., 2:4], dim=-2), dim=-1).sum(-1)


def compute_variance_waypoints(trajs):
    pos = trajs[..., :2]
    total = 0.0
    for via in pos.permute(1, 0, 2):
        d = torch.cdist(via, via, p=2)
        total = total + (torch.var(torch.triu(d, diagonal=1).view(-1)) if d.numel() > 1 else 0.0)
    return total


_SAVGOL = {}


def savgol_matrix(n, window_size=10, poly_order=2):
    """The Savitzky-Golay filter (scipy mode='interp') is linear in the signal: column j of the [n,n] operator is the
    filter applied to the j-th unit vector.  Built once on the host, applied on the device."""
    key = (n, window_size, poly_order)
    if key not in _SAVGOL:
        from scipy.signal import savgol_filter
        _SAVGOL[key] = torch.from_numpy(savgol_filter(np.eye(n), window_size, poly_order, axis=0)).float()
    return _SAVGOL[key]


def smooth_trajs(trajs, window_size=10, poly_order=2):
    """mmd/common/trajectory_utils.py:31-40 without its GPU -> CPU scipy -> GPU round trip: out = S @ trajs along the
    horizon with the precomputed SavGol operator S (same result up to fp32 rounding, pinned by golden g9)."""
    S = savgol_matrix(trajs.shape[1], window_size, poly_order).to(trajs.device)
    return torch.einsum("ij,bjd->bid", S, trajs.float())
