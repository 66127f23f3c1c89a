"""MPD / MPDEnsemble: the outer drop-in boundary (what CBS / PrioritizedPlanning call).

Mirrors reference mmd/planners/single_agent/{common.py:26-46, single_agent_planner_base.py:31-34, mpd.py:64-517,
mpd_ensemble.py:65-560}: same constructor keyword names, `__call__(start_state_pos, goal_state_pos, constraints_l,
experience) -> PlannerOutput`, same errors (ValueError on a start/goal mismatch, NotImplementedError on an unknown
planner_alg).  Differences, all forced by what exists offline:
  * datasets / checkpoints are not shipped, so when `<trained_models_dir>/<model_id>/args.yaml` is absent the model
    comes from `model_state_dict` (reference key names) + `model_args`, the map from `env_id` (default: the part of
    model_id before the first '-') and the normaliser limits from `normalizer_limits`;
  * the guide / sampler are the HIP kernels of libmmd_amd.so; there is no CPU fallback.
"""
import os
from math import ceil
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import synth
from .constraints import CostConstraint, MultiPointConstraint   # noqa: F401
from .diffusion_ensemble import DiffusionsEnsemble, HORIZON
from .diffusion_model import GaussianDiffusionModel, ddpm_sample_fn
from .guides import GuideManagerTrajectoriesWithVelocity
from .normalization import TrajectoryDatasetFacade
from . import postprocess as post
from .temporal_unet import UNET_DIM_MULTS, TemporalUnet


class PlannerOutput:
    """Same fields as the reference class (mmd/planners/single_agent/common.py:26-46)."""

    def __init__(self):
        self.trajs_iters = None
        self.trajs_final = None
        self.trajs_final_coll = None
        self.trajs_final_coll_idxs = None
        self.trajs_final_free = None
        self.trajs_final_free_idxs = None
        self.success_free_trajs = None
        self.fraction_free_trajs = None
        self.collision_intensity_trajs = None
        self.idx_best_traj = None
        self.traj_final_free_best = None
        self.cost_best_free_traj = None
        self.cost_smoothness = None
        self.cost_path_length = None
        self.cost_all = None
        self.variance_waypoint_trajs_final_free = None
        self.t_total = None
        self.constraints_l = None


class PathBatchExperience:
    """mmd/common/experiences.py:44: a previous batch of paths used to seed local inference."""

    def __init__(self, path_b):
        self.path_b = path_b


class _Timer:
    """TimerCUDA semantics (torch_timer.py:44-53): perf_counter bracketed by synchronisation -- of the CURRENT stream, which is where
    a planner call puts all of its work (the reference has one stream, so its device-wide synchronize() is the same thing; a
    device-wide one here would make concurrent planner calls on their own streams, plan_concurrently, wait for each other)."""

    def __enter__(self):
        import time
        torch.cuda.current_stream().synchronize()
        self._t0 = time.perf_counter()
        return self

    def __exit__(self, *a):
        import time
        torch.cuda.current_stream().synchronize()
        self.elapsed = time.perf_counter() - self._t0


class RobotPlanarDiskFacade:
    """What CBS / PrioritizedPlanning read from `planner.robot` (cbs.py:144,178,192,486; prioritized_planning.py:73,224,
    260,276; mmd/common/multi_agent_utils.py:44,74,79): the disk robot's constants, position / velocity slicing and
    `check_rr_collisions` (deps/torch_robotics/torch_robotics/robots/robot_planar_disk.py:173-203) -- the latter on the
    device through mmd_rr_collisions."""
    name = "RobotPlanarDisk"
    radius = 0.05                                    # mmd_params.py:30
    q_dim = 2

    def __init__(self, device="cuda"):
        self.device = torch.device(device)
        self.tensor_args = {"device": self.device, "dtype": torch.float32}
        self.q_min = torch.tensor([-1.0, -1.0], device=self.device)      # robot_planar_disk.py:31-33
        self.q_max = torch.tensor([1.0, 1.0], device=self.device)

    def get_position(self, x):
        return x[..., :2]

    def get_velocity(self, x):
        return x[..., 2:4]

    def check_rr_collisions(self, robot_q):
        """robot_q (..., n_robots, q_dim) -> (collisions (..., n, n) bool, collision_points (..., n, n, 2) with NaN where
        the pair does not collide)."""
        from .multi_agent import check_rr_collisions
        robot_q = torch.as_tensor(robot_q)
        assert robot_q.dim() >= 2
        lead, n = tuple(robot_q.shape[:-2]), robot_q.shape[-2]
        q = robot_q.to(device=self.device, dtype=torch.float32).reshape(-1, n, robot_q.shape[-1])[..., :2]
        coll, mid = check_rr_collisions(q.permute(1, 0, 2).contiguous(), margin=2.1 * self.radius)   # [T,n,n], [T,n,n,2]
        return (coll.view(lead + (n, n)).to(robot_q.device), mid.view(lead + (n, n, 2)).to(robot_q.device))


class PlanningTaskFacade:
    """What CBS / PP and the planner itself read from `planner.task`: `compute_collision`
    (deps/torch_robotics/torch_robotics/tasks/tasks.py:141-143, called from multi_agent_utils.py:47,84,89) and
    `get_trajs_collision_and_free` (tasks.py:236-311), both on the device against the guide's resident SDF texture."""

    def __init__(self, guide, robot, all_free=False):
        self.guide, self.robot, self.all_free = guide, robot, all_free
        self.tensor_args = robot.tensor_args

    def compute_collision(self, x, margin=None, **kwargs):
        """x [D] | [B, D] | [B, H, D] -> bool [1, 1] | [B, 1] | [B, H] (tasks.py:145-165 + the "(b h) -> b h" of
        distance_fields.py:47-48)."""
        x = torch.as_tensor(x)
        q = self.robot.get_position(x).to(device=self.guide.device, dtype=torch.float32)
        if q.ndim > 3:
            raise NotImplementedError
        if self.all_free:
            out = torch.zeros(q.shape[:-1], dtype=torch.bool, device=q.device)
        else:
            out = post.compute_collision(q, self.guide, margin=margin)
        if q.ndim == 1:
            out = out.view(1, 1)
        elif q.ndim == 2:
            out = out.view(-1, 1)
        return out.to(x.device)

    def get_trajs_collision_and_free(self, trajs, return_indices=False, num_interpolation=5):
        out = post.get_trajs_collision_and_free(trajs, self.guide, num_interpolation, all_free=self.all_free)
        return out if return_indices else (out[0], out[2])

    def compute_fraction_free_trajs(self, trajs, **kwargs):
        free_idxs = self.get_trajs_collision_and_free(trajs, return_indices=True)[3]
        return free_idxs.shape[0] / trajs.shape[0]


class PlanningTaskEnsembleFacade:
    """What CBS / PP and MPDEnsemble read from the multi-tile task (PlanningTaskEnsemble,
    deps/torch_robotics/torch_robotics/tasks/tasks_ensemble.py:39-365): the tile tasks, the tile <-> global transforms,
    `compute_collision` on global positions (tile inferred from the position, tile-frame occupancy against the tile's OWN map;
    a point outside every tile stays "in collision", :238-262) and the all-free `get_trajs_collision_and_free` stub (:271-277).
    The per-tile collision / free split of a planner call (get_traj_unnormalized, :79-88) goes through `tasks[m]`."""

    def __init__(self, task_guides: Dict[int, object], transforms: Dict[int, torch.Tensor], robot):
        self.robot, self.tensor_args = robot, robot.tensor_args
        self.tasks = {m: PlanningTaskFacade(g, robot) for m, g in task_guides.items()}
        self.transforms = {m: torch.as_tensor(t, dtype=torch.float32).cpu() for m, t in transforms.items()}

    def _transform(self, task_id, q):
        t = self.transforms[task_id].to(q.device)
        return torch.cat((t, torch.zeros(q.shape[-1] - t.shape[0], device=q.device))) if q.shape[-1] > t.shape[0] else t

    def transform_q(self, task_id: int, q):
        return q + self._transform(task_id, q)

    def inverse_transform_q(self, task_id: int, q):
        return q - self._transform(task_id, q)

    def infer_task_id_from_q_idx(self, q_idx):
        task_id = int(q_idx // HORIZON)
        return task_id, self.tasks[task_id]

    def infer_task_id_from_q(self, q):
        """q [n, 1, >=2] global positions -> [n] tile index, -1 outside every tile; where tiles touch the LATER tile wins (:355-363)."""
        q_pos = self.robot.get_position(torch.as_tensor(q))
        task_ids = torch.full((q_pos.shape[0],), -1, dtype=torch.long, device=q_pos.device)
        for i, m in enumerate(self.tasks):
            lo = self.transform_q(m, self.robot.q_min.to(q_pos.device))          # the tile env's limits (env_base limits = +-1)
            hi = self.transform_q(m, self.robot.q_max.to(q_pos.device))
            mask = torch.logical_and(q_pos >= lo, q_pos <= hi).all(dim=-1).reshape(-1)
            task_ids[mask] = i
        return task_ids

    def compute_collision(self, x, **kwargs):
        """x [2+] | [n, 2+] global -> int64 [] -> [1] | [n] (tasks_ensemble.py:227-262); [B, H, 2] with H > 1 fails in the reference
        too (the tile lookup is per leading row)."""
        x = torch.as_tensor(x)
        q = self.robot.get_position(x).to(device=self.robot.device, dtype=torch.float32)
        shape = q.shape[:-1]
        if q.ndim > 3:
            raise NotImplementedError
        if q.ndim == 3 and q.shape[1] != 1:
            raise IndexError("PlanningTaskEnsemble.compute_collision: one position per row (tasks_ensemble.py:361-363)")
        q = q.reshape(-1, 1, q.shape[-1])
        collisions = torch.ones(q.shape[0], dtype=torch.long, device=q.device)
        task_ids = self.infer_task_id_from_q(q)
        for i, m in enumerate(self.tasks):
            mask = task_ids == i
            if bool(mask.any()):
                local = self.inverse_transform_q(m, q[mask])
                collisions[mask] = self.tasks[m].compute_collision(local.reshape(-1, local.shape[-1]), **kwargs).reshape(-1).long()
        return collisions.view(shape).to(x.device)

    def get_trajs_collision_and_free(self, trajs, return_indices=False, num_interpolation=5):
        """tasks_ensemble.py:271-277: "return all valid"."""
        if return_indices:
            return (None, torch.tensor([]), trajs, torch.tensor([i for i in range(trajs.shape[0])]),
                    torch.tensor([False for _ in range(trajs.shape[1])]))
        return None, trajs

    def compute_fraction_free_trajs(self, trajs, **kwargs):
        return 1.0


def _same_state(given, stored):
    """torch.allclose(given, stored) (mpd.py:318-321) on the host in numpy: the two tiny torch calls cost 35 us per planner call."""
    g = given.detach().cpu().numpy() if torch.is_tensor(given) else np.asarray(given)
    s = stored.numpy()
    if g.shape != s.shape:
        return bool(torch.allclose(torch.as_tensor(given).cpu().float(), stored))      # (broadcasting / shape errors as torch has them)
    g = g.astype(np.float32, copy=False)
    return bool(np.all(np.abs(g - s) <= 1e-8 + 1e-5 * np.abs(s)))


def _check_states(planner, start_state_pos, goal_state_pos):
    if not _same_state(start_state_pos, planner.start_state_pos):
        raise ValueError("The start state is different from the one stored in the planner.")
    if not _same_state(goal_state_pos, planner.goal_state_pos):
        raise ValueError("The goal state is different from the one stored in the planner.")


def _fill_output(out, guide, trajs_iters, all_free=False):
    """mpd.py:344-405 / mpd_ensemble.py:385-429 on the device: ONE fused launch (collision / free split, path length,
    smoothness, SavGol) + the per-batch argmin + the waypoint variance of the free set."""
    trajs_final = trajs_iters[-1].contiguous()
    B, dev = trajs_final.shape[0], trajs_final.device
    summary = post.host_summary(B, 1, dev)
    r = post.postprocess_batch(guide, trajs_final, all_free=all_free, smooth=True, summary=summary)
    post.select_best(r.free_mask, 1, cost_a=r.path_length, cost_b=r.smoothness, summary=summary)
    # the call's ONE device -> host transfer: free mask, index of the cheapest free sample, the two costs, written by the two kernels
    # into one buffer (boolean-mask indexing, argwhere, .item() would each synchronise: nine round trips = 0.4 ms of a 3.4 ms
    # planner call at T = 25), and ONE host -> device copy of the two index lists back
    host = summary.cpu().numpy()
    fm, ib = host[:B] > 0, int(host[B])
    free_i, coll_i = np.flatnonzero(fm), np.flatnonzero(~fm)
    both = torch.from_numpy(np.concatenate((free_i, coll_i))).to(dev)
    free_idxs = both[:free_i.size].view(-1, 1)                       # [n, 1] int64, as torch.argwhere gives them (tasks.py:258-307)
    coll_idxs = both[free_i.size:].view(-1, 1)
    split = trajs_final.index_select(0, both)                        # (free rows first, then the colliding ones: one gather)
    free = split[:free_i.size] if free_i.size else None
    coll = split[free_i.size:] if coll_i.size else None
    out.trajs_iters, out.trajs_final = trajs_iters, r.smoothed
    out.trajs_final_coll, out.trajs_final_coll_idxs = coll, coll_idxs
    out.trajs_final_free, out.trajs_final_free_idxs = free, free_idxs
    out.success_free_trajs = free is not None
    out.fraction_free_trajs = 0.0 if free is None else free.shape[0] / B
    if free is not None:
        costs = summary[B + 1:].view(2, B).index_select(1, free_idxs.view(-1))      # (path lengths | smoothness of the free samples)
        out.cost_path_length, out.cost_smoothness = costs[0], costs[1]
        out.cost_all = out.cost_path_length + out.cost_smoothness
        idx_best_free = int(np.searchsorted(free_i, ib))          # position of the batch index ib among the free samples
        out.idx_best_traj = free_idxs[idx_best_free]
        out.idx_best_free_traj = idx_best_free
        out.traj_final_free_best = free[idx_best_free]
        out.cost_best_free_traj = float(np.float32(host[B + 1 + ib]) + np.float32(host[2 * B + 1 + ib]))
        out.variance_waypoint_trajs_final_free = post.compute_variance_waypoints(free)
    return out


def _fill_output_ensemble(out, task, tile_trajs_final, trajs_iters):
    """mpd_ensemble.py:385-429 + PlanningTaskEnsemble.get_traj_unnormalized / combine_trajs (tasks_ensemble.py:79-88, 162-225) on
    the device: every tile's final samples [B, 64, 4] (tile frame) are split against the tile's OWN map (one launch per tile), a
    sample is free iff it is free in every tile; costs, SavGol and the waypoint variance are those of the concatenated global-frame
    trajectories [B, K*64, 4] (one launch), the best sample the cheapest free one.  Shapes as combine_trajs leaves them: index
    tensors are int64 [n] (not [n, 1]), an empty free / colliding set is an empty tensor, and trajs_final_coll indexes the HORIZON
    axis of the final batch with the colliding sample indices (tasks_ensemble.py:190, `trajs_final[:, idxs]`) as the reference does."""
    trajs_final = trajs_iters[-1].contiguous()
    B, dev = trajs_final.shape[0], trajs_final.device
    free_mask = None
    for m, tf in tile_trajs_final.items():
        fm = post.postprocess_batch(task.tasks[m].guide, tf.contiguous(), smooth=False).free_mask
        free_mask = fm if free_mask is None else free_mask & fm
    summary = post.host_summary(B, 1, dev)
    r = post.postprocess_batch(task.tasks[next(iter(task.tasks))].guide, trajs_final, all_free=True, smooth=True, summary=summary)
    post.select_best(free_mask, 1, cost_a=r.path_length, cost_b=r.smoothness, summary=summary)
    host = summary.cpu().numpy()                                                                        # the call's one transfer
    fm, ib = host[:B] > 0, int(host[B])
    free_i, coll_i = np.flatnonzero(fm), np.flatnonzero(~fm)
    both = torch.from_numpy(np.concatenate((free_i, coll_i))).to(dev)
    free_idxs, coll_idxs = both[:free_i.size], both[free_i.size:]
    empty = torch.tensor([], dtype=torch.float32, device=dev)
    out.trajs_iters, out.trajs_final = trajs_iters, r.smoothed
    out.trajs_final_coll = trajs_final[:, coll_idxs] if coll_i.size else empty
    out.trajs_final_coll_idxs = coll_idxs
    out.trajs_final_free = trajs_final.index_select(0, free_idxs) if free_i.size else empty
    out.trajs_final_free_idxs = free_idxs
    out.success_free_trajs = 1 if free_i.size else 0
    out.fraction_free_trajs = free_i.size / B
    out.collision_intensity_trajs = 1 - out.fraction_free_trajs
    if free_i.size:
        out.cost_smoothness = r.smoothness.index_select(0, free_idxs)
        out.cost_path_length = r.path_length.index_select(0, free_idxs)
        out.cost_all = out.cost_smoothness + out.cost_path_length
        best = int(np.searchsorted(free_i, ib))
        out.idx_best_traj = free_idxs[best]
        out.traj_final_free_best = out.trajs_final_free[best]
        out.cost_best_free_traj = out.cost_all[best]
        out.variance_waypoint_trajs_final_free = post.compute_variance_waypoints(out.trajs_final_free)
    return out


def normalizer_limits_from_dataset(dataset_dir):
    """LimitsNormalizer limits exactly as TrajectoryDataset derives them (mmd/datasets/trajectories.py:84-112,
    normalization.py:91-94): min / max over every support point of every `trajs-free.pt` found under dataset_dir."""
    trajs = []
    for cur, _, files in os.walk(os.path.expanduser(dataset_dir), topdown=True):
        if "trajs-free.pt" in files:
            trajs.append(torch.load(os.path.join(cur, "trajs-free.pt"), map_location="cpu"))
    if not trajs:
        raise FileNotFoundError(f"no trajs-free.pt under {dataset_dir}")
    flat = torch.cat(trajs).float().reshape(-1, trajs[0].shape[-1])
    return flat.min(dim=0).values.numpy(), flat.max(dim=0).values.numpy()


def _load_model(model_id, trained_models_dir, model_state_dict, model_args, device):
    """mpd.py:117-177: args.yaml + checkpoint if present on disk, else the in-memory state dict."""
    args = dict(variance_schedule="exponential", n_diffusion_steps=25, predict_epsilon=True, unet_input_dim=32,
                unet_dim_mults_option=0, use_ema=True)
    model_dir = os.path.join(os.path.expanduser(trained_models_dir or ""), model_id or "")
    yaml_path = os.path.join(model_dir, "args.yaml")
    if model_state_dict is None and os.path.exists(yaml_path):
        import yaml
        with open(yaml_path) as f:
            args.update(yaml.safe_load(f))
        ckpt = "ema_model_current_state_dict.pth" if args.get("use_ema", True) else "model_current_state_dict.pth"
        model_state_dict = torch.load(os.path.join(model_dir, "checkpoints", ckpt), map_location="cpu")
    if model_args:
        args.update(model_args)
    if model_state_dict is None:
        raise FileNotFoundError(f"no checkpoint under {model_dir} and no model_state_dict given")
    unet = TemporalUnet(state_dim=4, n_support_points=HORIZON, unet_input_dim=args["unet_input_dim"],
                        dim_mults=UNET_DIM_MULTS[args["unet_dim_mults_option"]])
    unet.load_state_dict(model_state_dict)
    model = GaussianDiffusionModel(model=unet, variance_schedule=args["variance_schedule"],
                                   n_diffusion_steps=args["n_diffusion_steps"], predict_epsilon=args["predict_epsilon"])
    return model, args


class MPD:
    def __init__(self, model_id: str, planner_alg: str, start_state_pos, goal_state_pos,
                 use_guide_on_extra_objects_only: bool = False, start_guide_steps_fraction: float = 0.5,
                 n_guide_steps: int = 20, n_diffusion_steps_without_noise: int = 1,
                 weight_grad_cost_collision: float = 2e-2, weight_grad_cost_smoothness: float = 8e-2,
                 weight_grad_cost_constraints: float = 2e-1, weight_grad_cost_soft_constraints: float = 2e-2,
                 factor_num_interpolated_points_for_collision: float = 1.5, trajectory_duration: float = 5.0,
                 device: str = "cuda", debug: bool = False, seed: int = 18, results_dir: str = "logs",
                 trained_models_dir: str = "", n_samples: int = 64, n_local_inference_noising_steps: int = 3,
                 n_local_inference_denoising_steps: int = 3, model_state_dict=None, model_args=None, env_id=None,
                 normalizer_limits=None, obstacle_cutoff_margin=0.05, env_extra_objects=None, **kwargs):
        self.constraints = []
        self.weight_grad_cost_constraints = weight_grad_cost_constraints
        self.weight_grad_cost_soft_constraints = weight_grad_cost_soft_constraints
        if planner_alg == "mmd":
            self.run_prior_only, self.run_prior_then_guidance = False, False
        elif planner_alg == "diffusion_prior_then_guide":
            self.run_prior_only, self.run_prior_then_guidance = False, True
        elif planner_alg == "diffusion_prior":
            self.run_prior_only, self.run_prior_then_guidance = True, False
        else:
            raise NotImplementedError
        self.device = torch.device(device)
        self.tensor_args = {"device": self.device, "dtype": torch.float32}
        self.model, self.model_args = _load_model(model_id, trained_models_dir, model_state_dict, model_args, self.device)
        self.model.seed = seed
        self.env_id = env_id or (model_id.split("-")[0] if model_id else "EnvEmpty2D")
        if normalizer_limits is None and kwargs.get("dataset_dir"):
            normalizer_limits = normalizer_limits_from_dataset(kwargs["dataset_dir"])
        mins, maxs = normalizer_limits if normalizer_limits is not None else (synth.NORM_MINS, synth.NORM_MAXS)
        self.dataset = TrajectoryDatasetFacade(mins, maxs)
        self.robot = RobotPlanarDiskFacade(self.device)
        self.n_support_points = HORIZON
        self.start_state_pos = torch.as_tensor(start_state_pos, dtype=torch.float32).clone()
        self.goal_state_pos = torch.as_tensor(goal_state_pos, dtype=torch.float32).clone()
        self.hard_conds = self.dataset.get_hard_conditions(torch.vstack((self.start_state_pos, self.goal_state_pos)),
                                                           normalize=True)
        self.guide = GuideManagerTrajectoriesWithVelocity(
            self.dataset, env_id=self.env_id, obstacle_cutoff_margin=obstacle_cutoff_margin,
            weight_grad_cost_collision=weight_grad_cost_collision,
            weight_grad_cost_smoothness=weight_grad_cost_smoothness, trajectory_duration=trajectory_duration,
            n_support_points=HORIZON, extra_objects_only=use_guide_on_extra_objects_only,
            extra_objects=env_extra_objects, device=self.device)
        # the task's own collision checks (post-processing, compute_collision) always see the full map (tasks.py:141-311):
        # fixed objects + the env's extra objects (env.get_df_obj_list, env_base.py:76-89) + workspace boundaries
        self._task_guide = self.guide if not use_guide_on_extra_objects_only else GuideManagerTrajectoriesWithVelocity(
            self.dataset, env_id=self.env_id, obstacle_cutoff_margin=obstacle_cutoff_margin, n_support_points=HORIZON,
            extra_objects=env_extra_objects, device=self.device)
        self.task = PlanningTaskFacade(self._task_guide, self.robot)     # CBS reads planner.task (cbs.py:149)
        self.t_start_guide = ceil(start_guide_steps_fraction * self.model.n_diffusion_steps)
        self.n_guide_steps = n_guide_steps
        self.n_diffusion_steps_without_noise = n_diffusion_steps_without_noise
        self.num_samples = n_samples
        self.n_local_inference_noising_steps = n_local_inference_noising_steps
        self.n_local_inference_denoising_steps = n_local_inference_denoising_steps
        self.results_dir = results_dir
        self.context = None
        self.recent_call_data = PlannerOutput()
        self.sample_fn_kwargs = dict(
            guide=None if self.run_prior_then_guidance or self.run_prior_only else self.guide,
            n_guide_steps=self.n_guide_steps, t_start_guide=self.t_start_guide,
            noise_std_extra_schedule_fn=lambda x: 0.5)

    # ---- planner call -------------------------------------------------------------------------------------------
    def _cost_constraints(self, constraints_l):
        return [CostConstraint(self.robot, self.n_support_points, q_l=c.get_q_l(), traj_range_l=c.get_t_range_l(),
                               radius_l=c.radius_l, is_soft=c.is_soft) for c in (constraints_l or [])]

    def __call__(self, start_state_pos, goal_state_pos, constraints_l=None, experience=None, *args, soft_paths=None, **kwargs):
        """`soft_paths` (an extension; None = the reference's call): (paths_all [N, 64, 2] device tensor, this agent's index) -- the soft
        constraints from the other agents' current best paths (cbs.py:468-508) as one tensor instead of a MultiPointConstraint of
        (N - 1) x 63 tiny tensors in `constraints_l`; built on the device, placed after the groups of `constraints_l`, weighted with
        weight_grad_cost_soft_constraints: bitwise the list form's result without its 0.14 ms of host conversion per call."""
        _check_states(self, start_state_pos, goal_state_pos)
        cost_constraints_l = self._cost_constraints(constraints_l)
        self._soft_paths = soft_paths
        with _Timer() as timer:
            if experience is None:
                chain = self.run_constrained_inference(cost_constraints_l, **kwargs)
            else:
                chain = self.run_constrained_local_inference(cost_constraints_l, experience, **kwargs)
        out = PlannerOutput()
        out.t_total = timer.elapsed
        _fill_output(out, self._task_guide, self.dataset.unnormalize_trajectories(chain))
        out.constraints_l = constraints_l
        self.recent_call_data = out
        return out

    def _add_constraints(self, cost_constraints_l):
        self.guide.add_extra_costs(cost_constraints_l,
                                   [self.weight_grad_cost_soft_constraints if c.is_soft else
                                    self.weight_grad_cost_constraints for c in cost_constraints_l])
        soft_paths, self._soft_paths = getattr(self, "_soft_paths", None), None
        if soft_paths is not None:
            self.guide.set_soft_paths(soft_paths[0], soft_paths[1], weight=self.weight_grad_cost_soft_constraints)

    def _post_guidance(self, chain):
        """planner_alg 'diffusion_prior_then_guide' (mpd.py:429-453): extra guide steps after the prior sample."""
        if not self.run_prior_then_guidance:
            return chain
        n_post = (self.t_start_guide + self.n_diffusion_steps_without_noise) * self.n_guide_steps
        x = chain[-1].contiguous().clone()
        hard, mask = self.model._hard_tensor(self.hard_conds, 1, HORIZON, x.device, x.shape[-1])
        extra = torch.empty((n_post,) + tuple(x.shape), dtype=torch.float32, device=x.device)
        self.guide.guide_steps(x, hard, mask, n_post, chain=extra)        # ONE launch; every iteration lands in `extra`
        return torch.cat((chain, extra))

    def run_constrained_inference(self, cost_constraints_l, **kw):
        try:
            self._add_constraints(cost_constraints_l)
            chain = self.model.run_inference(
                self.context, self.hard_conds, n_samples=self.num_samples, horizon=self.n_support_points,
                return_chain=True, sample_fn=ddpm_sample_fn, **self.sample_fn_kwargs,
                n_diffusion_steps_without_noise=self.n_diffusion_steps_without_noise, device=self.device, **kw)
            chain = self._post_guidance(chain)
        finally:
            self.guide.reset_extra_costs()
        return chain

    def run_constrained_local_inference(self, cost_constraints_l, experience, **kw):
        try:
            self._add_constraints(cost_constraints_l)
            chain = self.model.run_local_inference(
                experience.path_b.to(self.device), self.n_local_inference_noising_steps,
                self.n_local_inference_denoising_steps, self.context, self.hard_conds, n_samples=self.num_samples,
                horizon=self.n_support_points, return_chain=True, sample_fn=ddpm_sample_fn, **self.sample_fn_kwargs,
                n_diffusion_steps_without_noise=self.n_diffusion_steps_without_noise, device=self.device, **kw)
            chain = self._post_guidance(chain)
        finally:
            self.guide.reset_extra_costs()
        return chain


class MPDEnsemble:
    """mpd_ensemble.py:65-560: K tile models chained along the horizon (K*64 support points), one guide per tile
    (obstacle_cutoff_margin 0.01, :139), constraints split per tile by time index and shifted to the tile frame."""

    def __init__(self, model_ids: tuple, transforms: Dict[int, torch.Tensor], planner_alg: str, start_state_pos,
                 goal_state_pos, use_guide_on_extra_objects_only: bool = False,
                 start_guide_steps_fraction: float = 0.5, n_guide_steps: int = 20,
                 n_diffusion_steps_without_noise: int = 1, weight_grad_cost_collision: float = 2e-2,
                 weight_grad_cost_smoothness: float = 8e-2, weight_grad_cost_constraints: float = 2e-1,
                 weight_grad_cost_soft_constraints: float = 2e-2,
                 factor_num_interpolated_points_for_collision: float = 1.5, trajectory_duration: float = 5.0,
                 device: str = "cuda", debug: bool = False, seed: int = 18, results_dir: str = "logs",
                 trained_models_dir: str = "", n_samples: int = 64, n_local_inference_noising_steps: int = 3,
                 n_local_inference_denoising_steps: int = 3, model_state_dicts=None, model_args=None, env_ids=None,
                 normalizer_limits=None, **kwargs):
        if planner_alg == "mmd":
            self.run_prior_only, self.run_prior_then_guidance = False, False
        elif planner_alg == "diffusion_prior_then_guide":
            self.run_prior_only, self.run_prior_then_guidance = False, True
        elif planner_alg == "diffusion_prior":
            self.run_prior_only, self.run_prior_then_guidance = True, False
        else:
            raise NotImplementedError
        self.weight_grad_cost_constraints = weight_grad_cost_constraints
        self.weight_grad_cost_soft_constraints = weight_grad_cost_soft_constraints
        self.device = torch.device(device)
        self.tensor_args = {"device": self.device, "dtype": torch.float32}
        mins, maxs = normalizer_limits if normalizer_limits is not None else (synth.NORM_MINS, synth.NORM_MAXS)
        self.transforms = {k: torch.as_tensor(v, dtype=torch.float32).cpu() for k, v in transforms.items()}
        self.models, self.guides, self.datasets, self.sample_kwargs, self.env_ids = {}, {}, [], {}, {}
        task_guides = {}
        for j, model_id in enumerate(model_ids):
            sd = None if model_state_dicts is None else model_state_dicts[j]
            model, _ = _load_model(model_id, trained_models_dir, sd, model_args, self.device)
            model.seed = seed + j
            self.models[j] = model
            self.env_ids[j] = (env_ids[j] if env_ids is not None else model_id.split("-")[0])
            ds = TrajectoryDatasetFacade(mins, maxs)
            self.datasets.append(ds)
            self.guides[j] = GuideManagerTrajectoriesWithVelocity(
                ds, env_id=self.env_ids[j], obstacle_cutoff_margin=0.01,
                weight_grad_cost_collision=weight_grad_cost_collision,
                weight_grad_cost_smoothness=weight_grad_cost_smoothness, trajectory_duration=trajectory_duration,
                n_support_points=HORIZON, extra_objects_only=use_guide_on_extra_objects_only, device=self.device)
            # the tile TASK always sees the tile's full map (tasks.py:141-311), whatever the guide is restricted to (:196-199)
            task_guides[j] = self.guides[j] if not use_guide_on_extra_objects_only else GuideManagerTrajectoriesWithVelocity(
                ds, env_id=self.env_ids[j], obstacle_cutoff_margin=0.01, n_support_points=HORIZON, device=self.device)
            self.sample_kwargs[j] = dict(
                guide=None if self.run_prior_then_guidance or self.run_prior_only else self.guides[j],
                n_guide_steps=n_guide_steps,
                t_start_guide=ceil(start_guide_steps_fraction * model.n_diffusion_steps),
                noise_std_extra_schedule_fn=lambda x: 0.5)
        K = len(model_ids)
        self.robot = RobotPlanarDiskFacade(self.device)
        self.task = PlanningTaskEnsembleFacade(task_guides, self.transforms, self.robot)    # mpd_ensemble.py:256-258, :328
        self.n_support_points = HORIZON
        self.start_state_pos = torch.as_tensor(start_state_pos, dtype=torch.float32).clone()
        self.goal_state_pos = torch.as_tensor(goal_state_pos, dtype=torch.float32).clone()
        start_local = self.start_state_pos - self.transforms[0]          # tasks_ensemble.inverse_transform_q
        goal_local = self.goal_state_pos - self.transforms[K - 1]
        nz = self.datasets[0].normalizer
        z = torch.zeros(2)
        self.hard_conds = {0: {0: nz.normalize(torch.cat((start_local, z)))}}
        self.hard_conds.setdefault(K - 1, {})[HORIZON - 1] = nz.normalize(torch.cat((goal_local, z)))
        self.cross_conds = {(i, i + 1): (HORIZON - 1, 0) for i in range(K - 1)}
        self.model = DiffusionsEnsemble(self.models, self.transforms)
        self.n_diffusion_steps_without_noise = n_diffusion_steps_without_noise
        self.num_samples = n_samples
        self.n_local_inference_noising_steps = n_local_inference_noising_steps
        self.n_local_inference_denoising_steps = n_local_inference_denoising_steps
        self.results_dir = results_dir
        self.recent_call_data = PlannerOutput()

    def infer_task_id_from_q_idx(self, idx):
        return int(idx) // HORIZON, int(idx) % HORIZON

    def split_cost_constraints_to_tasks(self, cost_constraints_l: List[CostConstraint]):
        """mpd_ensemble.py:431-507: one hard and one soft CostConstraint per tile, keyed by the tile of the range start."""
        buckets: Dict[Tuple[int, bool], list] = {}
        for c in cost_constraints_l:
            for j in range(c.qs.shape[0]):
                task_id, _ = self.infer_task_id_from_q_idx(c.traj_ranges[j, 0])
                buckets.setdefault((task_id, c.is_soft), []).append((c.qs[j], c.traj_ranges[j], c.radii[j]))
        out: Dict[int, list] = {}
        for soft in (False, True):
            for (task_id, is_soft), items in buckets.items():
                if is_soft != soft:
                    continue
                q_l, tr_l, r_l = zip(*items)
                out.setdefault(task_id, []).append(CostConstraint(
                    self.robot, HORIZON, q_l=[torch.from_numpy(np.asarray(q)) for q in q_l],
                    traj_range_l=[(float(t[0]), float(t[1])) for t in tr_l], radius_l=[float(r) for r in r_l],
                    is_soft=soft))
        return out

    def _add_constraints(self, cost_constraints_l):
        for task_id, cl in self.split_cost_constraints_to_tasks(cost_constraints_l).items():
            for c in cl:
                c.traj_ranges = c.traj_ranges - task_id * HORIZON               # mpd_ensemble.py:517
                c.qs = c.qs - self.transforms[task_id].numpy()                  # :518
                self.guides[task_id].add_extra_costs(
                    [c], [self.weight_grad_cost_constraints if not c.is_soft else self.weight_grad_cost_soft_constraints])

    def _reset(self):
        for g in self.guides.values():
            g.reset_extra_costs()

    def _post_guidance(self, chains):
        """planner_alg 'diffusion_prior_then_guide' (mpd_ensemble.py:540-564, :603-627): after the prior sample every tile runs
        (t_start_guide + n_diffusion_steps_without_noise) * n_guide_steps guide steps ON ITS OWN -- its guide (extra costs still
        loaded), its own hard conditions, NO cross conditioning between the tiles -- and the states after every step are appended to
        the tile's chain.  As in the reference a tile without an entry in `self.hard_conds` (a middle tile of K >= 3: only tiles 0 and
        K - 1 have one, :293-296) raises KeyError."""
        if not self.run_prior_then_guidance:
            return chains
        out = {}
        for j, guide in self.guides.items():
            skw = self.sample_kwargs[j]
            n_post = (skw["t_start_guide"] + self.n_diffusion_steps_without_noise) * skw["n_guide_steps"]
            hard_conds = self.hard_conds[j]                              # (KeyError for a tile without hard conditions, as :552)
            x = chains[j][-1].contiguous().clone()
            hard, mask = self.models[j]._hard_tensor(hard_conds, 1, HORIZON, x.device, x.shape[-1])
            extra = torch.empty((n_post,) + tuple(x.shape), dtype=torch.float32, device=x.device)
            guide.guide_steps(x, hard, mask, n_post, chain=extra)        # ONE launch per tile; every iteration lands in `extra`
            out[j] = torch.cat((chains[j], extra))
        return out

    def run_constrained_inference(self, cost_constraints_l, **kw):
        self._add_constraints(cost_constraints_l)
        try:
            return self._post_guidance(self.model.run_inference(
                None, self.hard_conds, cross_conds=self.cross_conds, n_samples=self.num_samples, return_chain=True,
                sample_fn=ddpm_sample_fn, sample_kwargs=self.sample_kwargs,
                n_diffusion_steps_without_noise=self.n_diffusion_steps_without_noise, device=self.device, **kw))
        finally:
            self._reset()

    def run_constrained_local_inference(self, cost_constraints_l, experience, **kw):
        self._add_constraints(cost_constraints_l)
        try:
            return self._post_guidance(self.model.run_local_inference(
                experience.path_b.to(self.device), self.n_local_inference_noising_steps,
                self.n_local_inference_denoising_steps, None, self.hard_conds, cross_conds=self.cross_conds,
                n_samples=self.num_samples, return_chain=True, sample_fn=ddpm_sample_fn,
                sample_kwargs=self.sample_kwargs,
                n_diffusion_steps_without_noise=self.n_diffusion_steps_without_noise, device=self.device, **kw))
        finally:
            self._reset()

    def __call__(self, start_state_pos, goal_state_pos, constraints_l=None, experience=None, *args, **kwargs):
        _check_states(self, start_state_pos, goal_state_pos)
        cl = [CostConstraint(self.robot, HORIZON, q_l=c.get_q_l(), traj_range_l=c.get_t_range_l(), radius_l=c.radius_l,
                             is_soft=c.is_soft) for c in (constraints_l or [])]
        with _Timer() as timer:
            chains = (self.run_constrained_inference(cl, **kwargs) if experience is None
                      else self.run_constrained_local_inference(cl, experience, **kwargs))
        # un-normalise per tile (the tile-frame final rows go to the tile's own collision check, tasks_ensemble.py:79-88), move to
        # the global frame, concatenate along the horizon (:162-175)
        parts, tile_final = [], {}
        for m in sorted(chains):
            tr = self.datasets[m].unnormalize_trajectories(chains[m]).clone()
            tile_final[m] = tr[-1].clone()
            tr[..., :2] += self.transforms[m].to(tr.device)
            parts.append(tr)
        trajs_iters = torch.cat(parts, dim=-2)                     # [T+2, B, K*64, D]
        out = PlannerOutput()
        out.t_total = timer.elapsed
        _fill_output_ensemble(out, self.task, tile_final, trajs_iters)
        out.constraints_l = constraints_l
        self.recent_call_data = out
        return out


def plan_concurrently(calls, seeds=None):
    """Independent planner calls issued CONCURRENTLY, one host thread and one HIP stream per call -- the re-entrancy the C ABI has per
    stream (no global state, scratch per (device, stream)) put to use where the reference loops: one planner per agent in the first
    round of CBS / PrioritizedPlanning (inference_multi_agent.py:225-237, cbs.py:316-324), the K calls of a batch of re-plans.  A call
    of 64 samples fills an eighth of the chip and is bound by the latency of its 101 dependent steps; four of them overlap almost
    completely (bench.py --workload config4).
    `calls`: a list of (planner, start_state_pos, goal_state_pos[, constraints_l[, experience]]); every planner at most once (a planner
    mutates its guide during a call: mpd.py:409,456 -- not re-entrant, as in the reference).
    `seeds`: one Philox seed per call; default: drawn from the global stream counter in list order, i.e. the seeds a sequential loop
    over `calls` would have used for single-model planners -- the results do not depend on how the threads interleave.
    Returns the PlannerOutputs in list order; an exception of any call is re-raised."""
    calls = [tuple(c) for c in calls]
    if len({id(c[0]) for c in calls}) != len(calls):
        raise ValueError("plan_concurrently: a planner appears twice (a planner call is not re-entrant)")
    if seeds is None:
        seeds = _default_seeds(calls)
    if len(seeds) != len(calls):
        raise ValueError("plan_concurrently: one seed per call")
    if not calls:
        return []
    # ("cuda" without an index = the caller's current device; the current device is per host thread, so the workers set it)
    devs = [torch.cuda.current_device() if c[0].device.index is None else c[0].device.index for c in calls]
    parents = [torch.cuda.current_stream(d) for d in devs]

    def run(j):
        planner, rest = calls[j][0], calls[j][1:]
        torch.cuda.set_device(devs[j])
        side = torch.cuda.Stream(devs[j])              # (from torch's per-device stream pool: see below why not a persistent one)
        side.wait_stream(parents[j])
        with torch.cuda.stream(side):
            out = planner(*rest, seed=int(seeds[j]))
        # The result tensors live in the side stream's pool of the caching allocator and are consumed on the caller's stream.  A block the
        # caller frees can only be handed out again by an allocation on the same pooled stream, i.e. inside a later plan_concurrently
        # call -- and every call starts by making its side stream wait for everything queued on the caller's stream (wait_stream above),
        # so work the caller queued on a result before dropping it is finished before the block is written again; the caller's stream is
        # ordered after the side stream and the host waits for it (t_total and the host-side fields are final).
        # Side streams are taken from torch's pool per CALL, not kept per worker: four persistent streams -- created per thread or up
        # front -- run config 4's round in 46 ms where pool streams taken per call need 30 - 33 (measured, round 6; the hardware-queue
        # placement of the first pool streams is the suspect).  The WORKER THREADS persist (the library keeps per-(thread, device)
        # streams for its chunked loop: a fresh set of threads per call would leave a set behind every time).
        parents[j].wait_stream(side)
        side.synchronize()
        return out
    return list(_worker_pool(len(calls)).map(run, range(len(calls))))


_POOL, _POOL_LOCK = None, __import__("threading").Lock()


def _worker_pool(n):
    """ONE persistent pool of host threads for plan_concurrently (the library keeps per-(thread, device) side streams for its chunked
    loop: a fresh pool per call would leave a set behind every time), grown when a call needs more workers."""
    global _POOL
    from concurrent.futures import ThreadPoolExecutor
    with _POOL_LOCK:
        if _POOL is None or _POOL._max_workers < n:
            if _POOL is not None:
                _POOL.shutdown(wait=True)
            _POOL = ThreadPoolExecutor(max_workers=max(n, 4), thread_name_prefix="mmd-planner")
        return _POOL


# ---- R independent planner calls as ONE launch sequence ----------------------------------------------------------------------------
def _default_seeds(calls):
    from .diffusion_model import next_stream_seed
    # (an MPD's model carries the planner's seed, an MPDEnsemble's first tile model)
    return [next_stream_seed(c[0].models[0].seed if hasattr(c[0], "models") else c[0].model.seed) for c in calls]


def _guide_key(g):
    return (g.margin, g.weight_collision, g.weight_smoothness, g.dt, g.sigma_gp, g.max_grad_norm, g.clip_grad, g.clip_grad_rule,
            g.max_grad_value, g.extra_objects_only, g._xs is None and g._xb is None,
            tuple(float(v) for v in g.dataset.normalizer.mins.cpu()), tuple(float(v) for v in g.dataset.normalizer.maxs.cpu()))


def _batch_key(call):
    """Calls with equal keys can share one launch sequence: same planner class and algorithm, the same device model(s) (weights are
    content-hashed: equal state dicts share one handle), schedule, batch size, sampler settings and guide parameters.  Maps,
    start / goal, constraints, tile transforms, seeds and the experience's seed batch may differ per call.  None: the call is run on
    its own (`diffusion_prior_then_guide`, a map with extra objects)."""
    planner = call[0]
    experience = call[4] if len(call) > 4 else None
    if planner.run_prior_then_guidance:
        return None
    # re-plans from an experience (run_local_inference: the same loop from a forward-noised seed batch) pack with each other
    local = None if experience is None else (planner.n_local_inference_noising_steps, planner.n_local_inference_denoising_steps)
    dev = planner.device
    if isinstance(planner, MPD):
        g = planner.guide
        if g._xs is not None or g._xb is not None:
            return None
        m = planner.model
        return ("MPD", str(dev), m.model.handle(m.n_diffusion_steps, dev).value, m.n_diffusion_steps, m.predict_epsilon, planner.num_samples,
                planner.n_guide_steps, planner.t_start_guide, planner.n_diffusion_steps_without_noise, planner.run_prior_only,
                _guide_key(g), m.sampler_flags, m.guide_coop_max, local)
    if isinstance(planner, MPDEnsemble):
        ms = planner.models
        return ("MPDEnsemble", str(dev), tuple(ms[j].model.handle(ms[j].n_diffusion_steps, dev).value for j in ms),
                ms[0].n_diffusion_steps, ms[0].predict_epsilon, planner.num_samples, planner.n_diffusion_steps_without_noise,
                planner.run_prior_only, tuple((planner.sample_kwargs[j]["n_guide_steps"], planner.sample_kwargs[j]["t_start_guide"])
                                              for j in ms), tuple(_guide_key(planner.guides[j]) for j in ms),
                tuple(sorted(planner.cross_conds.items())), local)
    return None


_BATCH_GUIDES = {}       # (ids of the planners' guides) -> (weakrefs, combined guide): built once per set of planners


def _combined_guide(guides):
    """ONE GuideManager over R robots (robot r = call r: its own map index, its own constraint groups) with the parameters the R
    guides share; cached per set of guide objects (the per-robot map table lives on the device)."""
    import weakref
    key = tuple(id(g) for g in guides)
    hit = _BATCH_GUIDES.get(key)
    if hit is not None and all(r() is g for r, g in zip(hit[0], guides)):
        return hit[1]
    g0 = guides[0]
    cg = GuideManagerTrajectoriesWithVelocity(
        g0.dataset, clip_grad=g0.clip_grad, clip_grad_rule=g0.clip_grad_rule, max_grad_norm=g0.max_grad_norm,
        max_grad_value=g0.max_grad_value, n_robots=len(guides), robot_env_ids=[g.env_id for g in guides],
        weight_grad_cost_collision=g0.weight_collision, weight_grad_cost_smoothness=g0.weight_smoothness,
        sigma_gp=g0.sigma_gp, extra_objects_only=g0.extra_objects_only, device=g0.device)
    cg.margin, cg.dt = g0.margin, g0.dt                       # (exactly the values the planners' own guides carry)
    while len(_BATCH_GUIDES) > 64:
        _BATCH_GUIDES.pop(next(iter(_BATCH_GUIDES)))
    _BATCH_GUIDES[key] = ([weakref.ref(g) for g in guides], cg)
    return cg


def _load_constraints(cg, per_robot):
    """per_robot[r] = [(CostConstraint, weight), ...] -> the combined guide's extra costs (one host pack for all robots)."""
    cg.reset_extra_costs()
    for r, pairs in enumerate(per_robot):
        if pairs:
            cg.add_extra_costs([c for c, _ in pairs], [w for _, w in pairs], robot=r)


def _split_outputs(calls, planners, r, summary, trajs_iters_all, B, t_total, ensemble_tasks=None):
    """The PlannerOutputs of a batched group from ONE post-processing launch + ONE device -> host transfer (post.host_summary)."""
    R, dev = len(planners), trajs_iters_all.device
    host = summary.cpu().numpy()
    fm_all, ib_all = host[:R * B] > 0, host[R * B:R * B + R].astype(np.int64)
    pl_all, sm_all = host[R * B + R:2 * R * B + R], host[2 * R * B + R:]
    outs = []
    for k, (call, planner) in enumerate(zip(calls, planners)):
        sl = slice(k * B, (k + 1) * B)
        out = PlannerOutput()
        out.t_total = t_total
        trajs_iters = trajs_iters_all[:, sl]
        trajs_final = trajs_iters[-1].contiguous()
        fm, ib = fm_all[sl], int(ib_all[k])
        free_i, coll_i = np.flatnonzero(fm), np.flatnonzero(~fm)
        ens = ensemble_tasks is not None
        free_idxs = torch.from_numpy(free_i).to(dev) if ens else torch.from_numpy(free_i).to(dev).view(-1, 1)
        coll_idxs = torch.from_numpy(coll_i).to(dev) if ens else torch.from_numpy(coll_i).to(dev).view(-1, 1)
        out.trajs_iters, out.trajs_final = trajs_iters, r.smoothed[sl]
        out.trajs_final_coll_idxs, out.trajs_final_free_idxs = coll_idxs, free_idxs
        if ens:                                             # shapes of combine_trajs (_fill_output_ensemble)
            empty = torch.tensor([], dtype=torch.float32, device=dev)
            out.trajs_final_coll = trajs_final[:, coll_idxs] if coll_i.size else empty
            out.trajs_final_free = trajs_final.index_select(0, free_idxs) if free_i.size else empty
            out.success_free_trajs = 1 if free_i.size else 0
            out.collision_intensity_trajs = 1 - free_i.size / B
        else:                                               # shapes of MPD.__call__ (_fill_output)
            out.trajs_final_coll = trajs_final.index_select(0, coll_idxs.view(-1)) if coll_i.size else None
            out.trajs_final_free = trajs_final.index_select(0, free_idxs.view(-1)) if free_i.size else None
            out.success_free_trajs = bool(free_i.size)
        out.fraction_free_trajs = free_i.size / B if ens else (0.0 if not free_i.size else free_i.size / B)
        if free_i.size:
            sel = free_idxs.view(-1) + k * B
            out.cost_smoothness = r.smoothness.index_select(0, sel)
            out.cost_path_length = r.path_length.index_select(0, sel)
            best = int(np.searchsorted(free_i, ib))
            out.idx_best_traj = free_idxs[best]
            out.traj_final_free_best = out.trajs_final_free[best]
            if ens:
                out.cost_all = out.cost_smoothness + out.cost_path_length
                out.cost_best_free_traj = out.cost_all[best]
            else:
                out.cost_all = out.cost_path_length + out.cost_smoothness
                out.idx_best_free_traj = best
                out.cost_best_free_traj = float(np.float32(pl_all[k * B + ib]) + np.float32(sm_all[k * B + ib]))
            out.variance_waypoint_trajs_final_free = post.compute_variance_waypoints(out.trajs_final_free)
        out.constraints_l = call[3] if len(call) > 3 else None
        planner.recent_call_data = out
        outs.append(out)
    return outs


def _check_start_goal(planner, call):
    _check_states(planner, call[1], call[2])


def _run_mpd_group(calls, seeds):
    planners = [c[0] for c in calls]
    p0, R, B = planners[0], len(calls), planners[0].num_samples
    dev = p0.device
    for p, c in zip(planners, calls):
        _check_start_goal(p, c)
    cg = _combined_guide([p.guide for p in planners])
    per_robot = []
    for p, c in zip(planners, calls):
        cl = p._cost_constraints(c[3] if len(c) > 3 else None)
        per_robot.append([(cc, p.weight_grad_cost_soft_constraints if cc.is_soft else p.weight_grad_cost_constraints) for cc in cl])
    _load_constraints(cg, per_robot)
    hard = {row: torch.stack([p.hard_conds[row] for p in planners]) for row in p0.hard_conds}
    experiences = [c[4] if len(c) > 4 else None for c in calls]                  # (all or none: _batch_key)
    kw = dict(n_samples=B, n_robots=R, horizon=HORIZON, return_chain=True, sample_fn=ddpm_sample_fn,
              guide=None if p0.run_prior_only else cg, n_guide_steps=p0.n_guide_steps, t_start_guide=p0.t_start_guide,
              noise_std_extra_schedule_fn=p0.sample_fn_kwargs["noise_std_extra_schedule_fn"],
              n_diffusion_steps_without_noise=p0.n_diffusion_steps_without_noise, device=dev, robot_seeds=seeds)
    with _Timer() as timer:
        try:
            if experiences[0] is None:
                chain = p0.model.run_inference(None, hard, **kw)
            else:
                # every call's seed batch forward-noised with the call's own Philox stream (q_sample under `seed`, as its own call does),
                # then ONE denoising loop over all of them
                noised = torch.cat([p.model.q_sample(e.path_b.to(dev), p.n_local_inference_noising_steps, seed=s)
                                    for p, e, s in zip(planners, experiences, seeds)])
                kw.pop("n_samples"), kw.pop("return_chain")
                chain = p0.model.conditional_sample(dict(hard), n_diffusion_steps=p0.n_local_inference_denoising_steps, batch_size=R * B,
                                                    return_chain=True, warm_start_path_b=noised, **kw)[1].transpose(0, 1)
        finally:
            cg.reset_extra_costs()
    trajs_iters = p0.dataset.unnormalize_trajectories(chain, n_tensors=R)        # [T+2, R*B, H, D]; every call's own clip decision
    tg = cg if all(p._task_guide is p.guide for p in planners) else _combined_guide([p._task_guide for p in planners])
    summary = post.host_summary(R * B, R, dev)
    r = post.postprocess_batch(tg, trajs_iters[-1].contiguous(), n_robots=R, smooth=True, summary=summary)
    post.select_best(r.free_mask, R, cost_a=r.path_length, cost_b=r.smoothness, summary=summary)
    return _split_outputs(calls, planners, r, summary, trajs_iters, B, timer.elapsed)


def _run_ensemble_group(calls, seeds):
    planners = [c[0] for c in calls]
    p0, R, B = planners[0], len(calls), planners[0].num_samples
    dev, keys = p0.device, list(p0.models.keys())
    for p, c in zip(planners, calls):
        _check_start_goal(p, c)
    cgs = {j: _combined_guide([p.guides[j] for p in planners]) for j in keys}
    per_tile = {j: [[] for _ in range(R)] for j in keys}
    for k, (p, c) in enumerate(zip(planners, calls)):
        cl = [CostConstraint(p.robot, HORIZON, q_l=cc.get_q_l(), traj_range_l=cc.get_t_range_l(), radius_l=cc.radius_l,
                             is_soft=cc.is_soft) for cc in ((c[3] if len(c) > 3 else None) or [])]
        for task_id, tile_cl in p.split_cost_constraints_to_tasks(cl).items():
            for cc in tile_cl:
                cc.traj_ranges = cc.traj_ranges - task_id * HORIZON                       # mpd_ensemble.py:517
                cc.qs = cc.qs - p.transforms[task_id].numpy()                             # :518
                per_tile[task_id][k].append((cc, p.weight_grad_cost_constraints if not cc.is_soft else p.weight_grad_cost_soft_constraints))
    for j in keys:
        _load_constraints(cgs[j], per_tile[j])
    hard = {j: {row: torch.stack([p.hard_conds[j][row] for p in planners]) for row in p0.hard_conds.get(j, {})} for j in keys}
    skw = {j: dict(p0.sample_kwargs[j], guide=None if p0.run_prior_only else cgs[j]) for j in keys}
    experiences = [c[4] if len(c) > 4 else None for c in calls]                  # (all or none: _batch_key)
    with _Timer() as timer:
        try:
            noised, n_steps = None, p0.model.n_diffusion_steps
            if experiences[0] is not None:                   # DiffusionsEnsemble.run_local_inference per call, then ONE denoising loop
                noised = torch.cat([p.models[keys[0]].q_sample(e.path_b.to(dev), p.n_local_inference_noising_steps, seed=s)
                                    for p, e, s in zip(planners, experiences, seeds)])
                n_steps = p0.n_local_inference_denoising_steps
            _, chains = p0.model.p_sample_loop(
                (R * B, HORIZON, p0.models[keys[0]].state_dim), hard, dict(p0.cross_conds), n_diffusion_steps=n_steps,
                return_chain=True, sample_fn=ddpm_sample_fn, n_diffusion_steps_without_noise=p0.n_diffusion_steps_without_noise,
                warm_start_path_b=noised, device=dev, n_robots=R, robot_seeds=seeds, robot_transforms=[p.transforms for p in planners],
                sample_kwargs=skw)
        finally:
            for j in keys:
                cgs[j].reset_extra_costs()
    # un-normalise per tile, tile-frame final rows to the tile's own collision check, global frame, concatenate (as MPDEnsemble.__call__)
    parts, free_mask = [], None
    for j in keys:
        tr = p0.datasets[j].unnormalize_trajectories(chains[j].transpose(0, 1), n_tensors=R).clone()   # [T+2, R*B, H, D]
        tgs = _combined_guide([p.task.tasks[j].guide for p in planners])
        fm = post.postprocess_batch(tgs, tr[-1].contiguous(), n_robots=R, smooth=False).free_mask
        free_mask = fm if free_mask is None else free_mask & fm
        offs = torch.stack([p.transforms[j] for p in planners]).to(tr.device).repeat_interleave(B, 0)     # [R*B, 2]
        tr[..., :2] += offs[None, :, None, :]
        parts.append(tr)
    trajs_iters = torch.cat(parts, dim=-2)                                                      # [T+2, R*B, K*64, D]
    summary = post.host_summary(R * B, R, dev)
    r = post.postprocess_batch(cgs[keys[0]], trajs_iters[-1].contiguous(), n_robots=R, all_free=True, smooth=True, summary=summary)
    post.select_best(free_mask, R, cost_a=r.path_length, cost_b=r.smoothness, summary=summary)
    return _split_outputs(calls, planners, r, summary, trajs_iters, B, timer.elapsed, ensemble_tasks=True)


def plan_batched(calls, seeds=None):
    """Independent planner calls as ONE launch sequence: the calls that share a device model, schedule and sampler settings (_batch_key)
    are packed robot-major into one [R * n_samples, H, D] batch -- each with its own start / goal, map index, constraint groups, tile
    transforms and Philox stream -- and sampled, post-processed and transferred together.  This is the reference's real call
    granularity put on the chip the way it fits: CBS / PrioritizedPlanning call one planner per agent with 64 samples
    (inference_multi_agent.py:225-237, cbs.py:316-324, mmd_params.py:33), and a UNet launch of 64 trajectories costs what one of 256
    does.  Same signature and seeds as plan_concurrently, and BITWISE the same results as the calls made one after the other with those
    seeds (mmd_sampler_desc.robot_seeds_dev: one Philox stream per robot).  Re-plans from an experience -- the two children of a CBS
    expansion re-plan two different agents independently (cbs.py:397-432) -- pack with each other.  Calls that cannot be packed
    (`diffusion_prior_then_guide`, extra objects) run on their own, in list order, with their seed."""
    calls = [tuple(c) for c in calls]
    if len({id(c[0]) for c in calls}) != len(calls):
        raise ValueError("plan_batched: a planner appears twice (a planner call is not re-entrant)")
    if seeds is None:
        seeds = _default_seeds(calls)
    if len(seeds) != len(calls):
        raise ValueError("plan_batched: one seed per call")
    groups, order = {}, []
    for j, c in enumerate(calls):
        key = _batch_key(c)
        key = ("single", j) if key is None else key
        if key not in groups:
            groups[key] = []
            order.append(key)
        groups[key].append(j)
    outs = [None] * len(calls)
    for key in order:
        js = groups[key]
        if key[0] == "single" or len(js) == 1:
            for j in js:
                outs[j] = calls[j][0](*calls[j][1:], seed=int(seeds[j]))
            continue
        run = _run_mpd_group if key[0] == "MPD" else _run_ensemble_group
        for j, out in zip(js, run([calls[j] for j in js], [int(seeds[j]) for j in js])):
            outs[j] = out
    return outs
