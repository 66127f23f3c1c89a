"""Device-side multi-agent helpers next to the sampler: robot-robot collisions of the best paths
(RobotPlanarDisk.check_rr_collisions as CBS.get_conflicts uses it, cbs.py:166-246) and the 'least_collisions' batch scan
(cbs.py:446-458).  Thin wrappers over the C ABI (mmd_rr_collisions, mmd_count_collisions)."""
import torch

from . import _lib

H = 64
ROBOT_RADIUS = 0.05
RR_MARGIN = 2.1 * ROBOT_RADIUS           # robot_planar_disk.py:186


def check_rr_collisions(paths, margin=RR_MARGIN, with_midpoints=True):
    """paths [N,T,2] un-normalised positions on the GPU -> (collisions [T,N,N] bool, midpoints [T,N,N,2] | None)."""
    n, T = paths.shape[0], paths.shape[1]
    mask = torch.empty((T, n, n), dtype=torch.uint8, device=paths.device)
    mid = torch.empty((T, n, n, 2), dtype=torch.float32, device=paths.device) if with_midpoints else None
    _lib.launch("mmd_rr_collisions", paths, _lib.require_gpu(paths.contiguous(), "paths"), n, T, float(margin),
                                             mask.data_ptr(), mid.data_ptr() if mid is not None else None)
    return mask.bool(), mid


def count_collisions(trajs, paths_all, robot0, n_local, margin=RR_MARGIN):
    """trajs [n_local*B,H,4] un-normalised sample batches of the local robots, paths_all [N,H,2] best paths of all
    robots -> int32 [n_local, B] number of (t, other robot) collision pairs per sample."""
    B = trajs.shape[0] // n_local
    counts = torch.empty(n_local * B, dtype=torch.int32, device=trajs.device)
    _lib.launch("mmd_count_collisions", trajs, _lib.require_gpu(trajs.contiguous(), "trajs"),
                                                _lib.require_gpu(paths_all.contiguous(), "paths_all"), robot0, n_local,
                                                B, paths_all.shape[0], H, float(margin), counts.data_ptr())
    return counts.view(n_local, B)


def least_collision_samples(trajs, paths_all, robot0, n_local):
    """Index of the first sample with the fewest collisions per local robot (strict '<' scan of cbs.py:452)."""
    return torch.argmin(count_collisions(trajs, paths_all, robot0, n_local), dim=1)
