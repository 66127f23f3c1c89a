"""LimitsNormalizer mirror (reference mmd/datasets/normalization.py:145-168) + the dataset facade the planner uses
(mmd/datasets/trajectories.py:198-239): the real limits come from dataset files that are not available offline, so
the limits are constructor arguments (defaults: mmd_amd.synth.NORM_MINS/MAXS)."""
import torch


class LimitsNormalizer:
    def __init__(self, mins, maxs):
        self.mins = torch.as_tensor(mins, dtype=torch.float32)
        self.maxs = torch.as_tensor(maxs, dtype=torch.float32)

    def to(self, device):
        self.mins, self.maxs = self.mins.to(device), self.maxs.to(device)
        return self

    def normalize(self, x):
        x = (x - self.mins.to(x.device)) / (self.maxs.to(x.device) - self.mins.to(x.device))
        return 2 * x - 1

    def unnormalize(self, x, eps=1e-4):
        if x.max() > 1 + eps or x.min() < -1 - eps:
            x = torch.clip(x, -1, 1)
        x = (x + 1) / 2.0
        return x * (self.maxs.to(x.device) - self.mins.to(x.device)) + self.mins.to(x.device)


class TrajectoryDatasetFacade:
    """The attributes / methods of TrajectoryDataset that MPD reads: normalizer, n_support_points, state_dim,
    (un)normalize_trajectories, get_hard_conditions."""

    def __init__(self, mins, maxs, n_support_points=64, state_dim=4):
        self.normalizer = LimitsNormalizer(mins, maxs)
        self.n_support_points = n_support_points
        self.state_dim = state_dim
        self.include_velocity = True

    def unnormalize_trajectories(self, x):
        return self.normalizer.unnormalize(x)

    def normalize_trajectories(self, x):
        return self.normalizer.normalize(x)

    def get_hard_conditions(self, traj, horizon=None, normalize=False):
        """trajectories.py:216-239: start/goal positions with zero velocity at rows 0 and H-1."""
        start = torch.cat((traj[0][..., :2], torch.zeros_like(traj[0][..., :2])), dim=-1)
        goal = torch.cat((traj[-1][..., :2], torch.zeros_like(traj[-1][..., :2])), dim=-1)
        if normalize:
            start, goal = self.normalizer.normalize(start), self.normalizer.normalize(goal)
        horizon = horizon or self.n_support_points
        return {0: start, horizon - 1: goal}
