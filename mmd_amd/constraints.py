"""Constraint types (mirror of reference mmd/common/constraints.py:46-85 and the CostConstraint holder,
deps/motion_planning_baselines/mp_baselines/planners/costs/cost_functions.py:275-295) and their packing into the
time-bucketed ELL table the guide kernel reads (include/mmd_amd.h: mmd_guide_desc.cons_ell_dev)."""
import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _lib

VERTEX_CONSTRAINT_RADIUS = 0.05 * 2.4        # mmd/config/mmd_params.py:52
H = 64


class MultiPointConstraint:
    """Same fields and accessors as the reference class (mmd/common/constraints.py:46-85)."""

    def __init__(self, q_l: List[torch.Tensor], t_range_l: List[Tuple[int, int]], radius_l: List[float] = None,
                 is_soft: bool = False):
        self.q_l = q_l
        self.t_range_l = t_range_l
        self.radius_l = [VERTEX_CONSTRAINT_RADIUS] * len(q_l) if radius_l is None else radius_l
        self.is_soft = is_soft

    def get_q_l(self):
        return self.q_l

    def get_t_range_l(self):
        return self.t_range_l

    def get_radius_l(self):
        return self.radius_l

    def get_is_soft(self):
        return self.is_soft

    def get_copy(self):
        return MultiPointConstraint(list(self.q_l), list(self.t_range_l), list(self.radius_l), self.is_soft)


def _points_xy(q_l) -> np.ndarray:
    """[n, 2] float32 host array of the (x, y) of every constraint point.  CBS hands over hundreds of tiny tensors per call
    (cbs.py:468-508), usually device tensors: they are stacked where they live and cross to the host in ONE copy (one by one, each
    `.cpu()` is a device synchronisation: 2 ms of a 5 ms planner call at 568 points)."""
    if torch.is_tensor(q_l):
        q = q_l
    else:
        q_l = list(q_l)
        if not q_l:            # (the reference's torch.stack(q_l) at cost_functions.py:293 fails the same way)
            raise RuntimeError("CostConstraint: stack expects a non-empty list of constraint points")
        try:                                        # (per-element Python work is what costs: one stack, no per-point calls)
            q = torch.stack(q_l)
        except (TypeError, RuntimeError):           # not all tensors, or mixed shapes / devices: the element-wise path
            return np.stack([np.asarray(torch.as_tensor(q).detach().cpu(), dtype=np.float32).reshape(-1)[:2]
                             for q in q_l]).astype(np.float32).reshape(-1, 2)
    q = q.detach().reshape(q.shape[0], -1)
    return np.ascontiguousarray(q[:, :2].to("cpu", torch.float32).numpy()).reshape(-1, 2)


def _ranges(traj_range_l) -> np.ndarray:
    """[n, 2] float32 (the reference keeps them as a float tensor too, cost_functions.py:294).  A list of (t0, t1) pairs goes through
    np.fromiter: 2.5 x faster than np.asarray on a list of tuples, and CBS hands over hundreds per call."""
    if isinstance(traj_range_l, (list, tuple)) and traj_range_l and all(type(t) is tuple and len(t) == 2 for t in traj_range_l):
        try:
            from itertools import chain
            return np.fromiter(chain.from_iterable(traj_range_l), dtype=np.float32, count=2 * len(traj_range_l)).reshape(-1, 2)
        except (TypeError, ValueError):
            pass
    return np.asarray(traj_range_l, dtype=np.float32).reshape(-1, 2)


class CostConstraint:
    """Parameter holder with the reference constructor signature (cost_functions.py:282-295).  One instance = one
    guide cost term = one ELL group (own gradient clip and weight)."""

    def __init__(self, robot=None, n_support_points=H, q_l=None, traj_range_l=None, radius_l=None, is_soft=False,
                 **kwargs):
        self.n_support_points = n_support_points
        self.qs = _points_xy(q_l)
        self.traj_ranges = _ranges(traj_range_l)
        self.radii = np.asarray(radius_l, dtype=np.float32).reshape(-1)
        self.is_soft = is_soft


def pack_constraints(per_robot_groups: Sequence[Sequence[Tuple[CostConstraint, float]]], device, return_max_slots=False):
    """per_robot_groups[r] = [(CostConstraint, weight), ...].  Returns device tensors
    (ell [n_slots,H,4] f32, grp_slot_off [G+1] i32, grp_weight [G] f32, robot_grp_off [R+1] i32) or None if empty."""
    lib = _lib.load()
    flat = [gw for groups in per_robot_groups for gw in groups]
    if not flat:
        return (None, 0) if return_max_slots else None
    G = len(flat)
    n_pts = (C.c_int32 * G)(*[g.qs.shape[0] for g, _ in flat])
    keep = []

    def ptr_array(arrs):
        arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in arrs]
        keep.append(arrs)
        return (C.c_void_p * G)(*[a.ctypes.data for a in arrs])

    q = ptr_array([g.qs for g, _ in flat])
    tr = ptr_array([g.traj_ranges for g, _ in flat])
    rad = ptr_array([g.radii for g, _ in flat])
    slots = (C.c_int32 * G)()
    _lib.check(lib.mmd_pack_constraints(G, n_pts, q, tr, rad, H, None, 0, slots))
    total = int(sum(slots))
    R = len(per_robot_groups)
    # ONE host buffer and ONE upload for the four tables (each pageable host -> device copy is a blocking call of its own, ~12 us,
    # in front of a planner call's first kernel): [ell | grp_slot_off | grp_weight | robot_grp_off], all 4-byte words
    n_ell = max(total, 1) * H * 4
    buf = np.zeros(n_ell + (G + 1) + G + (R + 1), dtype=np.float32)
    ell = buf[:n_ell].reshape(max(total, 1), H, 4)
    ell[..., 2] = -1.0
    _lib.check(lib.mmd_pack_constraints(G, n_pts, q, tr, rad, H, ell.ctypes.data, max(total, 1), slots))
    words = buf.view(np.int32)
    grp_slot_off = words[n_ell:n_ell + G + 1]
    grp_slot_off[1:] = np.cumsum(np.array(list(slots), dtype=np.int64))
    buf[n_ell + G + 1:n_ell + 2 * G + 1] = [w for _, w in flat]
    robot_grp_off = words[n_ell + 2 * G + 1:]
    robot_grp_off[1:] = np.cumsum([len(g) for g in per_robot_groups])
    dev = torch.from_numpy(buf).to(device)
    dev_words = dev.view(torch.int32)
    out = (dev[:n_ell].view(max(total, 1), H, 4), dev_words[n_ell:n_ell + G + 1], dev[n_ell + G + 1:n_ell + 2 * G + 1],
           dev_words[n_ell + 2 * G + 1:])
    if return_max_slots:                         # the most slots any one robot owns (sizes the guide kernel's on-chip table)
        per_robot = grp_slot_off[robot_grp_off[1:]] - grp_slot_off[robot_grp_off[:-1]]
        return out, int(per_robot.max())
    return out


def soft_constraints_from_paths(paths: torch.Tensor, robot0: int, n_local: int, radius=VERTEX_CONSTRAINT_RADIUS,
                                weight=2e-2):
    """Device-side all-pairs soft constraints (replaces cbs.py:468-508 for equal start times).  paths [N,H,2]
    un-normalised best-path positions of ALL robots on this device; returns the 4 constraint tensors for local
    robots [robot0, robot0+n_local)."""
    lib = _lib.load()
    n_all = paths.shape[0]
    dev = paths.device
    ell = torch.empty((n_local * (n_all - 1), H, 4), dtype=torch.float32, device=dev)
    gso = torch.empty(n_local + 1, dtype=torch.int32, device=dev)
    gw = torch.empty(n_local, dtype=torch.float32, device=dev)
    rgo = torch.empty(n_local + 1, dtype=torch.int32, device=dev)
    _lib.launch("mmd_soft_constraints_from_paths", paths, _lib.require_gpu(paths, "paths"), n_all, robot0, n_local, H,
                                                   float(radius), float(weight), ell.data_ptr(), gso.data_ptr(),
                                                   gw.data_ptr(), rgo.data_ptr())
    return ell, gso, gw, rgo, float(radius)
