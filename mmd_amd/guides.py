"""GuideManagerTrajectoriesWithVelocity: host mirror of reference mmd/models/diffusion_models/guides.py:152-259.
It only holds parameters; the gradient itself is the gfx950 kernel in mmd_amd/csrc/guide.hip.

One guide object can drive SEVERAL robots at once (the build's data-parallel restructuring, SURVEY.md headline fact
3): extra costs are kept per robot, robot r's samples are rows [r*B, (r+1)*B) of the trajectory batch."""
import ctypes as C
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .constraints import CostConstraint, pack_constraints
from .environments import LIMITS, sdf_grid_texture

ROBOT_RADIUS = 0.05                 # mmd/config/mmd_params.py:30

# Device-side sharing: one resident SDF texture per (map set, device) for every guide / task facade of the
# process (the reference precomputes the grid once per planner object, grid_map_sdf.py:34-63, N+1 times per instance).
_TEXTURES = {}
N_TEXTURE_UPLOADS = 0


def device_sdf_textures(maps, device):
    """[n_maps, n_grids=1, nx, ny, 4] float32 on `device`, uploaded once per process."""
    global N_TEXTURE_UPLOADS
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = (tuple(maps), str(device))
    if key not in _TEXTURES:
        tex = np.stack([sdf_grid_texture(m) for m in maps])[:, None]
        _TEXTURES[key] = torch.from_numpy(np.ascontiguousarray(tex)).to(device)
        N_TEXTURE_UPLOADS += 1
    return _TEXTURES[key]


class GuideManagerTrajectoriesWithVelocity:
    def __init__(self, dataset, cost=None, clip_grad=True, clip_grad_rule="norm", max_grad_norm=1.0, max_grad_value=0.1,
                 env_id="EnvEmpty2D", obstacle_cutoff_margin=0.05, robot_radius=ROBOT_RADIUS,
                 weight_grad_cost_collision=2e-2, weight_grad_cost_smoothness=8e-2, trajectory_duration=5.0,
                 n_support_points=64, sigma_gp=1.0, n_robots=1, robot_env_ids: Optional[Sequence[str]] = None,
                 extra_objects_only=False, extra_objects=None, device="cuda", tensor_args=None, **kwargs):
        # GuideManager.clip_gradient (guides.py:228-259).  NB the reference class defaults to clip_grad=False; MPD / MPDEnsemble
        # construct it with clip_grad=True, rule 'norm' (mpd.py:258-265), which is the default here
        if clip_grad and clip_grad_rule not in ("norm", "value"):
            raise NotImplementedError(clip_grad_rule)
        self.clip_grad, self.clip_grad_rule, self.max_grad_value = bool(clip_grad), clip_grad_rule, float(max_grad_value)
        self.dataset = dataset
        self.device = torch.device(tensor_args["device"] if tensor_args else device)
        self.n_robots = n_robots
        self.max_grad_norm = max_grad_norm
        self.weight_collision = weight_grad_cost_collision
        self.weight_smoothness = weight_grad_cost_smoothness
        self.dt = trajectory_duration / n_support_points                      # mpd.py:140
        self.sigma_gp = sigma_gp
        # collision_margins + cutoff_margin in fp32 (distance_fields.py:117, robot_planar_disk.py:68)
        self.margin = float(np.float32(np.float32(robot_radius * 1.1) + np.float32(obstacle_cutoff_margin)))
        env_ids = list(robot_env_ids) if robot_env_ids is not None else [env_id] * n_robots
        self.env_id = env_ids[0]
        maps = sorted(set(env_ids))
        self._grids = device_sdf_textures(maps, self.device)                  # [n_maps, n_grids=1, nx, ny, 4], shared
        self._robot_map = torch.tensor([maps.index(e) for e in env_ids], dtype=torch.int32, device=self.device)
        self._n_maps = len(maps)
        from .environments import MAP_BOXES
        self._obstacle_free = all(len(MAP_BOXES[m.replace("ExtraObjects", "")][0]) == 0 for m in maps)
        # use_guide_on_extra_objects_only (mpd.py:216-219): the only collision field is task.get_collision_fields_extra_objects()
        # = the env's extra ObjectField (EMPTY in every shipped map: env_*_extra_objects.py MultiSphereField([]), sdf == 1;
        # `extra_objects` below otherwise) -- no fixed-object grid and no workspace walls in the guide; GP prior and
        # constraints stay
        self.extra_objects_only = bool(extra_objects_only)
        # The env's obj_extra_list (env_base.py:76-89), given as {"spheres": [(cx, cy, r), ...], "boxes": [(cx, cy, size x,
        # size y), ...]} (MultiSphereField / MultiBoxField arguments; the box is the rounded one of the fixed objects): evaluated analytically by the
        # kernels, one more field next to the fixed objects' grid.  None / empty = the shipped ExtraObjects maps.
        xo = extra_objects or {}
        sph = np.asarray(xo.get("spheres", []), dtype=np.float32).reshape(-1, 3)
        box = np.asarray(xo.get("boxes", []), dtype=np.float32).reshape(-1, 4)
        self._xs = torch.from_numpy(np.concatenate([sph, np.zeros((len(sph), 1), np.float32)], 1)).to(self.device) if len(sph) else None
        self._xb = torch.from_numpy(np.concatenate([box[:, :2], box[:, 2:] / np.float32(2)], 1)).to(self.device) if len(box) else None
        # extra costs, per robot (guides.py:176-178, :228-234)
        self.extra_cost_l: List[List[CostConstraint]] = [[] for _ in range(n_robots)]
        self.extra_costs_grad_weight_l: List[List[float]] = [[] for _ in range(n_robots)]
        self._cons = None
        self._cons_dirty = True
        self._external_cons = None
        self._soft_paths = None
        self._max_slots = 0
        self._norm_limits = None

    # ---- extra costs (constraints) ----------------------------------------------------------------------------
    def add_extra_costs(self, extra_costs, extra_costs_grad_weights, robot=0):
        self.extra_cost_l[robot].extend(extra_costs)
        self.extra_costs_grad_weight_l[robot].extend(extra_costs_grad_weights)
        self._cons_dirty = True

    def reset_extra_costs(self):
        self.extra_cost_l = [[] for _ in range(self.n_robots)]
        self.extra_costs_grad_weight_l = [[] for _ in range(self.n_robots)]
        self._cons_dirty = True
        self._external_cons = None
        self._soft_paths = None

    def set_soft_paths(self, paths_all, self_index, radius=None, weight=2e-2):
        """The soft constraints from the OTHER agents' current best paths (cbs.py:468-508 for equal start times: every point t >= 1 of
        every other agent, range (t, t + 1), the vertex-constraint radius) handed over as ONE device tensor `paths_all` [N, 64, 2]
        (un-normalised positions of all agents, this one at `self_index`) instead of a MultiPointConstraint of N - 1 times 63 tiny
        tensors: the group is built on the device (mmd_soft_constraints_from_paths) and takes its place AFTER the groups added through
        add_extra_costs -- the order CBS passes them in (cbs.py:407-413: the agent's hard constraints, then the soft ones).  Bitwise
        the result of the list form.  One-robot guides only."""
        from .constraints import VERTEX_CONSTRAINT_RADIUS
        if self.n_robots != 1:
            raise NotImplementedError("set_soft_paths: a guide of one robot (MultiRobotSampler.set_other_paths is the many-robot form)")
        paths_all = paths_all.to(device=self.device, dtype=torch.float32).contiguous()
        if paths_all.dim() != 3 or paths_all.shape[1] != 64 or paths_all.shape[2] != 2 or not 0 <= int(self_index) < paths_all.shape[0]:
            raise ValueError(f"set_soft_paths: paths_all [N, 64, 2] and 0 <= self_index < N, got {tuple(paths_all.shape)}, {self_index}")
        self._soft_paths = (paths_all, int(self_index), float(VERTEX_CONSTRAINT_RADIUS if radius is None else radius), float(weight))
        self._cons_dirty = True

    def set_packed_constraints(self, cons):
        """Use device-built constraint tensors (constraints.soft_constraints_from_paths) instead of host-packed ones."""
        self._external_cons = cons
        self._max_slots = 0

    def _constraints(self):
        if self._external_cons is not None:
            return self._external_cons
        if self._cons_dirty:
            groups = [list(zip(c, w)) for c, w in zip(self.extra_cost_l, self.extra_costs_grad_weight_l)]
            self._cons, self._max_slots = pack_constraints(groups, self.device, return_max_slots=True)
            if self._soft_paths is not None and self._soft_paths[0].shape[0] > 1:
                from .constraints import soft_constraints_from_paths
                paths_all, idx, radius, weight = self._soft_paths
                soft = soft_constraints_from_paths(paths_all, idx, 1, radius, weight)        # (ell, gso, gw, rgo, radius): one group
                if self._cons is None:
                    self._cons, self._max_slots = soft, soft[0].shape[0]                     # (one radius: only (qx, qy) staged on chip)
                else:
                    ell_h, gso_h, gw_h, rgo_h = self._cons
                    n_h = self._max_slots                                                    # one robot: the host groups' slots
                    self._cons = (torch.cat((ell_h[:n_h], soft[0])), torch.cat((gso_h, soft[1][1:] + gso_h[-1:])),
                                  torch.cat((gw_h, soft[2])), rgo_h + (soft[1] > 0).to(torch.int32))
                    self._max_slots = n_h + soft[0].shape[0]
            self._cons_dirty = False
        return self._cons

    # ---- C-ABI descriptor -------------------------------------------------------------------------------------
    def desc(self):
        d = _lib.GuideDesc()
        nz = self.dataset.normalizer
        if self._norm_limits is None:                       # (host floats once: the normaliser's limits do not change)
            self._norm_limits = ([float(v) for v in nz.mins.cpu()], [float(v) for v in nz.maxs.cpu()])
        d.norm_min[:], d.norm_max[:] = self._norm_limits
        d.limits_lo[:] = LIMITS[0]
        d.limits_hi[:] = LIMITS[1]
        d.grid_nx, d.grid_ny = self._grids.shape[2], self._grids.shape[3]
        d.n_grids, d.n_maps = (0 if self._obstacle_free or self.extra_objects_only else 1), self._n_maps
        d.sdf_grids_dev = self._grids.data_ptr()
        d.robot_map_dev = self._robot_map.data_ptr() if self._n_maps > 1 else None
        d.ws_min[:] = [float(np.float32(LIMITS[0][k]) * np.float32(1.08)) for k in range(2)]      # tasks.py:81-83
        d.ws_max[:] = [float(np.float32(LIMITS[1][k]) * np.float32(1.08)) for k in range(2)]
        if self.extra_objects_only:
            d.ws_min[:], d.ws_max[:] = [-1e6, -1e6], [1e6, 1e6]      # out of reach: the workspace term is identically 0
        d.margin, d.dt, d.sigma_gp = self.margin, self.dt, self.sigma_gp
        d.weight_collision, d.weight_smoothness = self.weight_collision, self.weight_smoothness
        d.max_grad_norm = self.max_grad_norm
        d.clip_grad_rule = 2 if not self.clip_grad else (1 if self.clip_grad_rule == "value" else 0)
        d.max_grad_value = self.max_grad_value
        if self._xs is not None:
            d.extra_spheres_dev, d.n_extra_spheres = self._xs.data_ptr(), self._xs.shape[0]
        if self._xb is not None:
            d.extra_boxes_dev, d.n_extra_boxes = self._xb.data_ptr(), self._xb.shape[0]
        cons = self._constraints()
        if cons is not None:
            ell, gso, gw, rgo = cons[:4]
            # 5th element (soft_constraints_from_paths): the one radius every active point of the table has
            d.cons_uniform_radius = float(cons[4]) if len(cons) > 4 else 0.0
            d.cons_ell_dev, d.grp_slot_off_dev = ell.data_ptr(), gso.data_ptr()
            d.grp_weight_dev, d.robot_grp_off_dev = gw.data_ptr(), rgo.data_ptr()
            # upper bound of the slots any one robot owns (exact when robots own equal shares, as the all-pairs table)
            d.max_slots_per_robot = int(self._max_slots) if self._max_slots else -(-ell.shape[0] // max(self.n_robots, 1))
            self._keep = cons
        return d

    # ---- guide_gradient_steps / forward -----------------------------------------------------------------------
    def guide_steps(self, x, hard, hard_rows, n_steps, chain=None):
        """In place: n_steps x { x += guide(x); apply_hard_conditioning } (sample_functions.py:89-107).
        x [n_robots*B,H,D]; hard [n_robots, n_rows, D] normalised pinned states in ascending row order, hard_rows the 64-bit mask of
        those rows (_lib.HARD_ROWS_START_GOAL for the start / goal pair, 0 for none); chain (optional) [n_steps, n_robots*B, H, D]
        receives the state after every iteration."""
        d = self.desc()
        B = x.shape[0] // self.n_robots
        if chain is not None:
            assert chain.shape == (n_steps,) + tuple(x.shape)
        _lib.launch("mmd_guide_steps", x, C.byref(d), _lib.require_gpu(x, "x"), _lib.require_gpu(hard, "hard"),
                                               int(hard_rows) & 0xFFFFFFFFFFFFFFFF, self.n_robots, B, n_steps,
                                               _lib.require_gpu(chain, "chain") if chain is not None else None)
        return x

    def forward(self, x_normalized):
        """grad = guide(x) exactly as the reference returns it (guides.py:180-226): one step without hard
        conditioning on a copy, minus the input."""
        y = x_normalized.contiguous().clone()
        hard = torch.zeros(self.n_robots, 2, x_normalized.shape[-1], device=y.device)
        self.guide_steps(y, hard, 0, 1)
        return y - x_normalized

    __call__ = forward
