"""Post-sampling selection (the step right after the hot path): collision / free split, path length +
smoothness costs, SavGol smoothing, waypoint variance and the per-robot pick -- gfx950 kernels of
mmd_amd/csrc/postprocess.hip behind the C ABI (mmd_postprocess_trajs, mmd_select_best, mmd_points_collision,
mmd_variance_waypoints).  Mirrors deps/torch_robotics/torch_robotics/tasks/tasks.py:141-311, trajectory/metrics.py:7-39,
trajectory/utils.py:73-86 and mmd/common/trajectory_utils.py:31-51.  The SDF texture is the guide's resident one; there
is no CPU path."""
import ctypes as C

import numpy as np
import torch

from . import _lib

ROBOT_RADIUS = 0.05                 # mmd/config/mmd_params.py:30
H = 64
Q_MIN, Q_MAX = (-1.0, -1.0), (1.0, 1.0)     # RobotPlanarDisk q_limits (robot_planar_disk.py:31-33)

_SAVGOL_HOST, _SAVGOL_DEV = {}, {}


def savgol_matrix(n, window_size=10, poly_order=2):
    """The Savitzky-Golay filter (scipy mode='interp') is linear in the signal: column j of the [n,n] operator is the
    filter applied to the j-th unit vector.  Built once on the host, applied on the device."""
    key = (n, window_size, poly_order)
    if key not in _SAVGOL_HOST:
        from scipy.signal import savgol_filter
        _SAVGOL_HOST[key] = torch.from_numpy(savgol_filter(np.eye(n), window_size, poly_order, axis=0)).float().contiguous()
    return _SAVGOL_HOST[key]


def _savgol_dev(n, device, window_size=10, poly_order=2):
    key = (n, str(device), window_size, poly_order)
    if key not in _SAVGOL_DEV:
        _SAVGOL_DEV[key] = savgol_matrix(n, window_size, poly_order).to(device)
    return _SAVGOL_DEV[key]


def interpolation_alphas(num_interpolation):
    """torch.linspace(0, 1, n + 2)[1:n+1] (trajectory/utils.py:79), fp32."""
    return np.ascontiguousarray(torch.linspace(0, 1, num_interpolation + 2)[1:num_interpolation + 1].numpy(),
                                dtype=np.float32)


class PostprocessResult:
    """Per-trajectory outputs of ONE mmd_postprocess_trajs launch (all device tensors)."""
    __slots__ = ("free_mask", "path_length", "smoothness", "smoothed", "waypoint_collisions")


def host_summary(n, n_robots, device):
    """One fp32 buffer [free flags (n) | picks (n_robots) | path lengths (n) | smoothness (n)]: postprocess_batch(summary=...) and
    select_best(summary=...) fill it, `.cpu()` on it is the planner call's ONE device -> host transfer."""
    return torch.empty(3 * n + n_robots, dtype=torch.float32, device=device)


def postprocess_batch(guide, trajs, n_robots=1, num_interpolation=5, margin=ROBOT_RADIUS, all_free=False, smooth=True,
                      want_waypoints=False, window_size=10, poly_order=2, summary=None):
    """trajs [n_robots*B, K*64, 4] un-normalised on the GPU (K = 1; K tiles for MPDEnsemble); `guide` supplies the resident
    map (its C-ABI descriptor).  `summary` (host_summary): path lengths / smoothness are written into it."""
    trajs = trajs.contiguous()
    n, h, d = trajs.shape
    if h % H or d != 4 or n % n_robots:
        raise ValueError(f"postprocess_batch: expected [n_robots*B, K*{H}, 4], got {tuple(trajs.shape)}")
    dev = trajs.device
    desc = guide.desc() if hasattr(guide, "desc") else guide
    r = PostprocessResult()
    r.free_mask = torch.empty(n, dtype=torch.uint8, device=dev)
    if summary is not None:
        r.path_length, r.smoothness = summary[n + n_robots:2 * n + n_robots], summary[2 * n + n_robots:]
    else:
        r.path_length = torch.empty(n, dtype=torch.float32, device=dev)
        r.smoothness = torch.empty(n, dtype=torch.float32, device=dev)
    r.smoothed = torch.empty_like(trajs) if smooth else None
    r.waypoint_collisions = (torch.empty((n, (h - 1) * num_interpolation), dtype=torch.uint8, device=dev)
                             if want_waypoints else None)
    alpha = interpolation_alphas(num_interpolation)
    sav = _savgol_dev(h, dev, window_size, poly_order) if smooth else None
    fp = C.POINTER(C.c_float)
    _lib.launch("mmd_postprocess_trajs", trajs, C.byref(desc), _lib.require_gpu(trajs, "trajs"), n_robots, n // n_robots, h, num_interpolation,
        alpha.ctypes.data_as(fp), float(margin), (C.c_float * 2)(*Q_MIN), (C.c_float * 2)(*Q_MAX), int(bool(all_free)),
        sav.data_ptr() if sav is not None else None, window_size,      # rows of the operator are zero beyond +-window
        r.waypoint_collisions.data_ptr() if r.waypoint_collisions is not None else None, r.free_mask.data_ptr(),
        r.path_length.data_ptr(), r.smoothness.data_ptr(), r.smoothed.data_ptr() if smooth else None)
    return r


def select_best(free_mask, n_robots, cost_a=None, cost_b=None, counts=None, summary=None):
    """Per robot: (index of the best free sample, number of free samples), int32 device tensors (mmd_select_best); `summary`
    (host_summary) receives the free flags and the picks as floats."""
    n = free_mask.shape[0]
    dev = free_mask.device
    idx = torch.empty(n_robots, dtype=torch.int32, device=dev)
    n_free = torch.empty(n_robots, dtype=torch.int32, device=dev)
    _lib.launch("mmd_select_best", free_mask, free_mask.data_ptr(), cost_a.data_ptr() if cost_a is not None else None,
        cost_b.data_ptr() if cost_b is not None else None,
        counts.contiguous().data_ptr() if counts is not None else None, n_robots, n // n_robots, idx.data_ptr(),
        n_free.data_ptr(), summary.data_ptr() if summary is not None else None)
    return idx, n_free


def split_free(trajs, free_mask):
    """(coll, coll_idxs, free, free_idxs) with the reference's shapes (tasks.py:258-307): idxs are [n, 1]."""
    fm = free_mask.bool()
    free_idxs = torch.argwhere(fm)
    coll_idxs = torch.argwhere(~fm)
    free = trajs[fm] if free_idxs.numel() else None
    coll = trajs[~fm] if coll_idxs.numel() else None
    return coll, coll_idxs, free, free_idxs


def get_trajs_collision_and_free(trajs, guide, num_interpolation=5, all_free=False):
    """tasks.py:236-311 for [B,H,D] batches.  Returns (coll, coll_idxs, free, free_idxs, waypoint_collisions)."""
    r = postprocess_batch(guide, trajs, num_interpolation=num_interpolation, all_free=all_free, smooth=False,
                          want_waypoints=True)
    coll, coll_idxs, free, free_idxs = split_free(trajs, r.free_mask)
    return coll, coll_idxs, free, free_idxs, r.waypoint_collisions.bool()


def compute_collision(points, guide, margin=None, map_index=0):
    """PlanningTask.compute_collision (tasks.py:141-143): points [..., >=2] on the GPU -> bool [...]; default margin =
    collision_margins + cutoff_margin (distance_fields.py:320)."""
    pts = points.to(dtype=torch.float32).contiguous()
    flat = pts.reshape(-1, pts.shape[-1])
    out = torch.empty(flat.shape[0], dtype=torch.uint8, device=pts.device)
    desc = guide.desc() if hasattr(guide, "desc") else guide
    _lib.launch("mmd_points_collision", flat, C.byref(desc), _lib.require_gpu(flat, "points"), flat.shape[0],
                                                flat.shape[1], map_index,
                                                float(getattr(guide, "margin", desc.margin) if margin is None else margin),
                                                out.data_ptr())
    return out.bool().view(pts.shape[:-1])


def compute_variance_waypoints(trajs):
    """trajectory/metrics.py:17-27 for a [B,L,4] batch on the GPU."""
    trajs = trajs.contiguous()
    var_t = torch.empty(trajs.shape[1], dtype=torch.float32, device=trajs.device)
    _lib.launch("mmd_variance_waypoints", trajs, _lib.require_gpu(trajs, "trajs"), trajs.shape[0], trajs.shape[1],
                                                  var_t.data_ptr())
    return var_t.sum()


def smooth_trajs(trajs, guide, window_size=10, poly_order=2):
    """mmd/common/trajectory_utils.py:31-40 without its GPU -> CPU scipy -> GPU round trip."""
    return postprocess_batch(guide, trajs, num_interpolation=0, all_free=True, smooth=True, window_size=window_size,
                             poly_order=poly_order).smoothed
