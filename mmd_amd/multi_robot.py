"""Data-parallel multi-robot guided sampling: the build's restructuring of the reference's sequential per-robot
planner calls (scripts/inference/inference_multi_agent.py:225-237, cbs.py:316-324) into ONE batched sampling call
per planning round, sharded over GPUs by robot.

Robots are independent inside a sampling call (other robots enter only as frozen constraint points), so
rank g owns robots [g*n_local, (g+1)*n_local) and the only exchange is ONE all-gather per round of the chosen best
paths [n_local, H, 2] -> [N, H, 2] (RCCL over xGMI on GPUs, gloo in the CPU tests), after which every rank rebuilds
its robots' soft-constraint table on device (mmd_soft_constraints_from_paths).
"""
from math import ceil

import torch

from . import synth
from .constraints import VERTEX_CONSTRAINT_RADIUS, soft_constraints_from_paths
from .diffusion_model import ddpm_sample_fn
from .guides import GuideManagerTrajectoriesWithVelocity
from .normalization import TrajectoryDatasetFacade

H, D = 64, 4


def shard_range(n_robots, rank, world_size):
    """Contiguous block of robots owned by `rank` (n_robots must divide evenly)."""
    if n_robots % world_size:
        raise ValueError(f"{n_robots} robots do not shard evenly over {world_size} ranks")
    n_local = n_robots // world_size
    return rank * n_local, n_local


def all_gather_paths(paths_local, world_size, group=None, force_collective=False):
    """[n_local,H,2] -> [n_local*world_size,H,2], rank-major (== robot order).  One collective per planning round.
    `force_collective` runs the collective even in a one-rank group (the hardware test of the RCCL branch)."""
    import torch.distributed as dist
    if world_size == 1 and not force_collective:
        # a single-rank sampler never issues a collective, whether or not a (larger) process group exists around it
        return paths_local
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("all_gather_paths: world_size > 1 needs an initialised torch.distributed process group")
    if dist.get_world_size(group) != world_size:
        raise ValueError(f"all_gather_paths: world_size={world_size} but the process group has "
                         f"{dist.get_world_size(group)} ranks")
    if paths_local.is_cuda and dist.get_backend(group) == "gloo":
        # gloo has no device collectives: stage through the host (CPU tests / single-GPU rehearsal of the N>1 path)
        parts = [torch.empty_like(paths_local, device="cpu") for _ in range(world_size)]
        dist.all_gather(parts, paths_local.cpu().contiguous(), group=group)
        return torch.cat(parts, dim=0).to(paths_local.device)
    out = torch.empty((world_size * paths_local.shape[0],) + tuple(paths_local.shape[1:]), dtype=paths_local.dtype,
                      device=paths_local.device)
    dist.all_gather_into_tensor(out, paths_local.contiguous(), group=group)
    return out


class MultiRobotSampler:
    def __init__(self, model, starts, goals, env_id="EnvEmpty2D", n_samples=64, rank=0, world_size=1,
                 norm_mins=synth.NORM_MINS, norm_maxs=synth.NORM_MAXS, n_guide_steps=20,
                 start_guide_steps_fraction=0.5, n_diffusion_steps_without_noise=1,
                 weight_grad_cost_soft_constraints=2e-2, radius=VERTEX_CONSTRAINT_RADIUS, device="cuda", group=None,
                 n_streams=0, inter_robot=True):
        self.model = model
        self.n_robots = starts.shape[0]
        self.rank, self.world_size, self.group = rank, world_size, group
        self.robot0, self.n_local = shard_range(self.n_robots, rank, world_size)
        self.n_samples = n_samples
        self.device = torch.device(device)
        self.dataset = TrajectoryDatasetFacade(norm_mins, norm_maxs)
        self.guide = GuideManagerTrajectoriesWithVelocity(self.dataset, env_id=env_id, n_robots=self.n_local,
                                                          device=device)
        sl = slice(self.robot0, self.robot0 + self.n_local)
        st = torch.as_tensor(starts[sl], dtype=torch.float32)
        go = torch.as_tensor(goals[sl], dtype=torch.float32)
        z = torch.zeros_like(st)
        nz = self.dataset.normalizer
        self.hard_conds = {0: nz.normalize(torch.cat((st, z), -1)).to(self.device),
                           H - 1: nz.normalize(torch.cat((go, z), -1)).to(self.device)}
        self.n_guide_steps = n_guide_steps
        self.t_start_guide = ceil(start_guide_steps_fraction * model.n_diffusion_steps)
        self.n_extra = n_diffusion_steps_without_noise
        self.w_soft, self.radius = weight_grad_cost_soft_constraints, radius
        self.n_streams = n_streams      # mmd_sampler_desc.n_streams (0 = the library's choice: 2 chunks above 512 trajectories)
        # False: no inter-robot term (BASELINE config 2: every robot guided by the map, the workspace and the GP prior alone) --
        # plan_round then needs no exchange step either
        self.inter_robot = inter_robot

    def set_other_paths(self, paths_all):
        """paths_all [N,H,2] un-normalised best paths of ALL robots (this device) or None (no inter-robot term)."""
        if paths_all is None or self.n_robots < 2:
            self.guide.reset_extra_costs()
        else:
            self.guide.set_packed_constraints(soft_constraints_from_paths(
                paths_all.contiguous(), self.robot0, self.n_local, self.radius, self.w_soft))

    def sample(self, seed=None, x_init=None, step_noise=None, return_chain=False):
        """One guided sampling round for the local robots: [n_local*B, H, D] normalised trajectories.  The in-kernel
        Philox noise is keyed by (seed, global trajectory index): every rank passes the SAME seed and gets exactly the
        rows the unsharded run would produce."""
        return self.model.run_inference(
            None, self.hard_conds, n_samples=self.n_samples, n_robots=self.n_local, horizon=H,
            return_chain=return_chain, sample_fn=ddpm_sample_fn, guide=self.guide, n_guide_steps=self.n_guide_steps,
            t_start_guide=self.t_start_guide, noise_std_extra_schedule_fn=lambda t: 0.5,
            n_diffusion_steps_without_noise=self.n_extra, warm_start_path_b=x_init, step_noise=step_noise, seed=seed,
            traj_index_base=self.robot0 * self.n_samples, device=self.device, n_streams=self.n_streams)

    def unnormalize(self, trajs_normalized):
        nz = self.dataset.normalizer
        mins, maxs = nz.mins.to(trajs_normalized.device), nz.maxs.to(trajs_normalized.device)
        return (torch.clip(trajs_normalized, -1, 1) + 1) / 2.0 * (maxs - mins) + mins

    def best_paths(self, trajs_normalized, paths_all=None):
        """Selection for the exchange step, on the device: samples that collide with the map or leave the joint limits
        are dropped first (PlanningTask.get_trajs_collision_and_free, tasks.py:236-311, as MPD.__call__ does at
        mpd.py:357-382); among the free ones the pick is the first sample with the fewest robot-robot collisions against
        the other robots' current best paths (CBS 'least_collisions', cbs.py:446-458), or the cheapest one (path length +
        smoothness, mpd.py:366-370) when no paths are known yet.  A robot without any free sample falls back to the same
        criterion over all its samples (`self.last_n_free` tells).  Returns un-normalised positions [n_local,H,2]."""
        from . import postprocess as post
        t = self.unnormalize(trajs_normalized).contiguous()
        r = post.postprocess_batch(self.guide, t, n_robots=self.n_local, smooth=False)
        if paths_all is None or self.n_robots < 2:
            idx, n_free = post.select_best(r.free_mask, self.n_local, cost_a=r.path_length, cost_b=r.smoothness)
        else:
            from .multi_agent import count_collisions
            counts = count_collisions(t, paths_all, self.robot0, self.n_local)
            idx, n_free = post.select_best(r.free_mask, self.n_local, counts=counts.view(-1))
        self.last_n_free, self.last_idx = n_free, idx
        tv = t.view(self.n_local, self.n_samples, H, D)
        return tv[torch.arange(self.n_local, device=t.device), idx.long()][..., :2].contiguous()

    def plan_round(self, paths_local, seed=None):
        """all-gather -> constraint table -> guided sampling -> new local best paths."""
        paths_all = all_gather_paths(paths_local, self.world_size, self.group) if self.inter_robot else None
        self.set_other_paths(paths_all)
        trajs = self.sample(seed=seed)
        return trajs, self.best_paths(trajs, paths_all)
