"""CPU oracle for the guided-diffusion trajectory sampler of yoraish/mmd  --  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (PyTorch-CPU fp32, closed-form guide gradients, no HIP) of the reference's hot
path.  It is the checker for the HIP path: only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import it.  The product package `mmd_amd` never imports it and fails
loudly when its HIP library is missing.

Parity status: PINNED.  Every function below is checked against outputs of the genuine reference, imported
from /root/reference in the build container by `tools/make_golden.py`, committed as `tests/golden/*.npz`
(tests/test_oracle_golden.py re-checks them on every run; no reference code travels).

Citations are relative to /root/reference/.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------------------------
# A1  variance schedules + diffusion buffers
# --------------------------------------------------------------------------------------------------------------


def exponential_beta_schedule(n_diffusion_steps, beta_start=1e-4, beta_end=1.0):
    """mmd/models/diffusion_models/helpers.py:43-49."""
    x = torch.linspace(0, n_diffusion_steps, n_diffusion_steps)
    beta_start = torch.tensor(beta_start, dtype=torch.float32)
    beta_end = torch.tensor(beta_end, dtype=torch.float32)
    a = 1 / n_diffusion_steps * torch.log(beta_end / beta_start)
    return beta_start * torch.exp(a * x)


def cosine_beta_schedule(n_diffusion_steps, s=0.008, a_min=0, a_max=0.999):
    """mmd/models/diffusion_models/helpers.py:28-40."""
    steps = n_diffusion_steps + 1
    x = np.linspace(0, steps, steps)
    alphas_cumprod = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    alphas_cumprod = alphas_cumprod / alphas_cumprod[0]
    betas = 1 - (alphas_cumprod[1:] / alphas_cumprod[:-1])
    return torch.tensor(np.clip(betas, a_min=a_min, a_max=a_max), dtype=torch.float32)


def _np_sqrt(t):
    return torch.from_numpy(np.sqrt(t.numpy()))


SCHEDULE_KEYS = ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                 "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
                 "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                 "posterior_mean_coef1", "posterior_mean_coef2")


def schedule_tables(n_diffusion_steps, variance_schedule="exponential") -> Dict[str, torch.Tensor]:
    """Buffers of GaussianDiffusionModel.__init__ (mmd/models/diffusion_models/diffusion_model_base.py:69-105)."""
    if variance_schedule == "cosine":
        betas = cosine_beta_schedule(n_diffusion_steps)
    elif variance_schedule == "exponential":
        betas = exponential_beta_schedule(n_diffusion_steps)
    else:
        raise NotImplementedError(variance_schedule)
    alphas = 1.0 - betas
    alphas_cumprod = torch.cumprod(alphas, axis=0)
    alphas_cumprod_prev = torch.cat([torch.ones(1), alphas_cumprod[:-1]])
    posterior_variance = betas * (1.0 - alphas_cumprod_prev) / (1.0 - alphas_cumprod)
    tb = {
        "betas": betas,
        "alphas_cumprod": alphas_cumprod,
        "alphas_cumprod_prev": alphas_cumprod_prev,
        "sqrt_alphas_cumprod": torch.sqrt(alphas_cumprod),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - alphas_cumprod),
        "log_one_minus_alphas_cumprod": torch.log(1.0 - alphas_cumprod),
        "sqrt_recip_alphas_cumprod": torch.sqrt(1.0 / alphas_cumprod),
        "sqrt_recipm1_alphas_cumprod": torch.sqrt(1.0 / alphas_cumprod - 1),
        "posterior_variance": posterior_variance,
        "posterior_log_variance_clipped": torch.log(torch.clamp(posterior_variance, min=1e-20)),
        # the reference uses np.sqrt on torch tensors here (:103,:105); numpy's and torch's vectorised fp32 sqrt
        # differ by 1 ulp on some inputs, so use numpy's to stay bit-identical
        "posterior_mean_coef1": betas * _np_sqrt(alphas_cumprod_prev) / (1.0 - alphas_cumprod),
        "posterior_mean_coef2": (1.0 - alphas_cumprod_prev) * _np_sqrt(alphas) / (1.0 - alphas_cumprod),
    }
    return tb


# --------------------------------------------------------------------------------------------------------------
# A4  TemporalUnet forward (state_dict keyed exactly like the reference, without the `model.` prefix)
# --------------------------------------------------------------------------------------------------------------


def _group_norm_n_groups(n_channels, target_n_groups=8):
    """mmd/models/layers/layers.py:392-398."""
    if n_channels < target_n_groups:
        return 1
    for n_groups in range(target_n_groups, target_n_groups + 10):
        if n_channels % n_groups == 0:
            return n_groups
    return 1


def sinusoidal_pos_emb(t, dim=32):
    """mmd/models/layers/layers.py:246-258."""
    half_dim = dim // 2
    emb = math.log(10000) / (half_dim - 1)
    emb = torch.exp(torch.arange(half_dim) * -emb)
    emb = t[:, None] * emb[None, :]
    return torch.cat((emb.sin(), emb.cos()), dim=-1)


def time_embedding(sd, t):
    """TimeEncoder: mmd/models/layers/layers.py:232-243."""
    e = sinusoidal_pos_emb(t)
    e = F.linear(e, sd["time_mlp.encoder.1.weight"], sd["time_mlp.encoder.1.bias"])
    e = F.mish(e)
    return F.linear(e, sd["time_mlp.encoder.3.weight"], sd["time_mlp.encoder.3.bias"])


def _conv1d_block(sd, prefix, x):
    """Conv1d(k=5,p=2) -> GroupNorm -> Mish: mmd/models/layers/layers.py:279-296."""
    w = sd[f"{prefix}.block.0.weight"]
    x = F.conv1d(x, w, sd[f"{prefix}.block.0.bias"], padding=w.shape[-1] // 2)
    x = F.group_norm(x, _group_norm_n_groups(w.shape[0]), sd[f"{prefix}.block.2.weight"],
                     sd[f"{prefix}.block.2.bias"], eps=1e-5)
    return F.mish(x)


def _rtb(sd, prefix, x, c):
    """ResidualTemporalBlock.forward: mmd/models/layers/layers.py:346-358."""
    cond = F.linear(F.mish(c), sd[f"{prefix}.cond_mlp.1.weight"], sd[f"{prefix}.cond_mlp.1.bias"])
    h = _conv1d_block(sd, f"{prefix}.blocks.0", x) + cond[:, :, None]
    h = _conv1d_block(sd, f"{prefix}.blocks.1", h)
    if f"{prefix}.residual_conv.weight" in sd:
        res = F.conv1d(x, sd[f"{prefix}.residual_conv.weight"], sd[f"{prefix}.residual_conv.bias"])
    else:
        res = x
    return h + res


def unet_levels(sd) -> int:
    """number of resolution levels = len(dim_mults) (temporal_unet.py:50-72: one `downs` entry per level)."""
    return 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("downs."))


def unet_forward(sd: Dict[str, torch.Tensor], x, t, n_levels=None):
    """TemporalUnet.forward with conditioning_type=None, self_attention=False
    (mmd/models/diffusion_models/temporal_unet.py:121-174).  x [B,H,D] fp32, t [B] (any numeric) -> [B,H,D].  With a
    float64 state dict and x it is the fp64 yardstick of the accuracy tests.  n_levels None: read from the state dict."""
    n_levels = unet_levels(sd) if n_levels is None else n_levels
    c = time_embedding(sd, t.to(x.dtype))
    x = x.transpose(1, 2)                                            # 'b h c -> b c h'
    skips = []
    for ind in range(n_levels):
        x = _rtb(sd, f"downs.{ind}.0", x, c)
        x = _rtb(sd, f"downs.{ind}.1", x, c)
        skips.append(x)
        if ind < n_levels - 1:
            x = F.conv1d(x, sd[f"downs.{ind}.4.conv.weight"], sd[f"downs.{ind}.4.conv.bias"], stride=2, padding=1)
    x = _rtb(sd, "mid_block1", x, c)
    x = _rtb(sd, "mid_block2", x, c)
    for ind in range(n_levels - 1):
        x = torch.cat((x, skips.pop()), dim=1)
        x = _rtb(sd, f"ups.{ind}.0", x, c)
        x = _rtb(sd, f"ups.{ind}.1", x, c)
        x = F.conv_transpose1d(x, sd[f"ups.{ind}.4.conv.weight"], sd[f"ups.{ind}.4.conv.bias"], stride=2, padding=1)
    x = _conv1d_block(sd, "final_conv.0", x)
    x = F.conv1d(x, sd["final_conv.1.weight"], sd["final_conv.1.bias"])
    return x.transpose(1, 2)


def state_dict_to_torch(sd_np) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v)).float() for k, v in sd_np.items()}


# --------------------------------------------------------------------------------------------------------------
# A14  map geometry -> SDF grid (value + gradient), closed form
# --------------------------------------------------------------------------------------------------------------

# MultiBoxField (= MultiRoundedBoxField, primitives.py:312-333, alias :344) centres / sizes of the fixed objects.
MAP_BOXES = {
    "EnvEmpty2D": (np.zeros((0, 2)), np.zeros((0, 2))),                                    # env_empty_2d.py:25-54
    "EnvEmptyNoWait2D": (np.zeros((0, 2)), np.zeros((0, 2))),                              # env_empty_nowait_2d.py
    "EnvHighways2D": (np.array([[0, 0.0], [0., 0.875], [0., -0.875], [0.875, 0.0], [-0.875, 0.0], [0.875, 0.875],
                                [0.875, -0.875], [-0.875, 0.875], [-0.875, -0.875]]),
                      np.array([[0.5, 0.5], [0.5, 0.25], [0.5, 0.25], [0.25, 0.5], [0.25, 0.5], [0.25, 0.25],
                                [0.25, 0.25], [0.25, 0.25], [0.25, 0.25]])),                # env_highways_2d.py:54-78
    "EnvConveyor2D": (np.array([[0, 0], [0, 0.35], [0, -0.35]]),
                      np.array([[0.8, 0.1], [1.0, 0.1], [1.0, 0.1]])),                      # env_conveyor_2d.py:53-64
    "EnvDropRegion2D": (np.array([[0.4, 0.4], [-0.4, 0.4], [0.4, -0.4], [-0.4, -0.4]]),
                        np.array([[0.4, 0.4]] * 4)),                                       # env_drop_region_2d.py:60-75
}


def rounded_boxes_sdf(x, centers, sizes):
    """MultiRoundedBoxField.compute_signed_distance_impl (primitives.py:326-333); the empty MultiSphereField that
    accompanies it in every map contributes the constant 1 (primitives.py:109-110) through ObjectField's min
    (:567-570).  x [...,2] -> sdf [...]."""
    x = torch.as_tensor(x, dtype=torch.float32)
    ones = torch.ones_like(x[..., 0])
    if len(centers) == 0:
        return ones
    centers = torch.as_tensor(centers, dtype=torch.float32)
    sizes = torch.as_tensor(sizes, dtype=torch.float32)
    half_sizes = sizes / 2
    radius = torch.min(sizes, dim=-1)[0] * 0.15
    d = torch.abs(x.unsqueeze(-2) - centers.unsqueeze(0))
    q = d - half_sizes.unsqueeze(0) + radius.unsqueeze(0).unsqueeze(-1)
    max_q = torch.amax(q, dim=-1)
    sdfs = torch.minimum(max_q, torch.zeros_like(max_q)) + torch.linalg.norm(torch.relu(q), dim=-1) - radius.unsqueeze(0)
    return torch.minimum(ones, torch.min(sdfs, dim=-1)[0])


def build_sdf_grid(map_name, limits=((-1.0, -1.0), (1.0, 1.0)), cell_size=0.005):
    """GridMapSDF.precompute_sdf (deps/torch_robotics/torch_robotics/environments/grid_map_sdf.py:34-63): sdf value
    and its autograd gradient on `linspace(lo, hi, cmap)`^2, cmap = ceil(map_dim / cell).  Returns (sdf [nx,ny],
    grad [nx,ny,2]) float32.  The reference differentiates row by row with autograd; the sdf is a min/max/norm
    composition, so one batched autograd pass over the whole grid gives the same sub-gradient selection."""
    centers, sizes = MAP_BOXES[map_name]
    lo = torch.tensor(limits[0], dtype=torch.float32)
    hi = torch.tensor(limits[1], dtype=torch.float32)
    cmap = torch.ceil(torch.abs(hi - lo) / cell_size).long()
    xs = torch.linspace(lo[0], hi[0], int(cmap[0]))
    ys = torch.linspace(lo[1], hi[1], int(cmap[1]))
    pts = torch.stack(torch.meshgrid(xs, ys, indexing="ij"), dim=-1).requires_grad_(True)
    sdf = rounded_boxes_sdf(pts, centers, sizes)
    if sdf.requires_grad:
        (grad,) = torch.autograd.grad(sdf.sum(), pts)
    else:                                              # empty map: sdf == 1 everywhere, zero gradient
        grad = torch.zeros_like(pts)
    return sdf.detach().contiguous(), grad.detach().contiguous()


# --------------------------------------------------------------------------------------------------------------
# A6-A12  guide gradient
# --------------------------------------------------------------------------------------------------------------


@dataclass
class ConstraintGroup:
    """One CostConstraint (cost_functions.py:275-326) = one MultiPointConstraint (mmd/common/constraints.py:46-85):
    n points q [n,2], time ranges [t0,t1) [n,2] (exclusive end, cost_functions.py:305), radii [n], one weight."""
    q: torch.Tensor
    t_range: torch.Tensor
    radius: torch.Tensor
    weight: float


@dataclass
class GuideParams:
    """Everything `GuideManagerTrajectoriesWithVelocity.forward` (guides.py:180-226) reads, for ONE robot."""
    norm_mins: torch.Tensor                      # [D] LimitsNormalizer mins  (normalization.py:145-168)
    norm_maxs: torch.Tensor                      # [D]
    sdf_grids: List[Tuple[torch.Tensor, torch.Tensor]]   # [(sdf [nx,ny], grad [nx,ny,2])]  df_obj_list (env_base.py:76-89)
    limits_lo: torch.Tensor = field(default_factory=lambda: torch.tensor([-1.0, -1.0]))
    limits_hi: torch.Tensor = field(default_factory=lambda: torch.tensor([1.0, 1.0]))
    ws_min: torch.Tensor = field(default_factory=lambda: torch.tensor([-1.08, -1.08]))   # tasks.py:75-86 (x1.08)
    ws_max: torch.Tensor = field(default_factory=lambda: torch.tensor([1.08, 1.08]))
    robot_radius: float = 0.05                   # mmd_params.py:30
    cutoff_margin: float = 0.05                  # mpd.py:127 (MPDEnsemble: 0.01, mpd_ensemble.py:139)
    dt: float = 5.0 / 64                         # trajectory_duration / n_support_points, mpd.py:140
    weight_collision: float = 2e-2               # mmd_params.py:40
    weight_smoothness: float = 8e-2              # mmd_params.py:41
    max_grad_norm: float = 1.0                   # guides.py:154
    sigma_gp: float = 1.0                        # mpd.py:237
    sigma_coll: float = 1.0                      # mpd.py:227
    # the env's extra objects (EnvBase.obj_extra_list, env_base.py:76-89: one ObjectField of primitive fields, identity
    # pose), evaluated analytically next to the grids of the fixed objects.  None: the env has no obj_extra_list
    extra_spheres: torch.Tensor = None           # [n,3] (cx, cy, r)          MultiSphereField (primitives.py:108-115)
    extra_boxes: torch.Tensor = None             # [n,4] (cx, cy, sx, sy)     MultiBoxField (= MultiRoundedBoxField by the alias at primitives.py:345; :326-333), sizes
    extra_only: bool = False                     # use_guide_on_extra_objects_only (mpd.py:216-219): this field alone

    @property
    def margin(self):
        """collision_margins + cutoff_margin (distance_fields.py:117); margins = 1.1 r (robot_planar_disk.py:68)."""
        return float(np.float32(np.float32(self.robot_radius * 1.1) + np.float32(self.cutoff_margin)))


def unnormalize(x, mins, maxs, eps=1e-4, clip_mode="reference"):
    """LimitsNormalizer.unnormalize (mmd/datasets/normalization.py:157-168).  clip_mode 'reference' clips the
    WHOLE tensor iff any element is out of [-1-eps, 1+eps]; 'always' clips unconditionally (what the HIP kernel
    does; they differ only for values in (1, 1+eps])."""
    if clip_mode == "always" or (x.max() > 1 + eps or x.min() < -1 - eps):
        x = torch.clip(x, -1, 1)
    x = (x + 1) / 2.0
    return x * (maxs - mins) + mins


def normalize(x, mins, maxs):
    """LimitsNormalizer.normalize (normalization.py:150-155)."""
    x = (x - mins) / (maxs - mins)
    return 2 * x - 1


def clip_grad_by_norm(g, max_grad_norm=1.0):
    """guides.py:247-253 (the +1e-6 sits INSIDE the norm, over all D state dims)."""
    n = torch.linalg.norm(g + 1e-6, dim=-1, keepdims=True)
    return torch.clip(n, 0.0, max_grad_norm) / n * g


def sdf_lookup(p, gp: GuideParams, k=0):
    """GridMapSDF.get_sdf (grid_map_sdf.py:84-114): nearest-cell value + precomputed gradient."""
    sdf, grad = gp.sdf_grids[k]
    map_dim = torch.abs(gp.limits_hi - gp.limits_lo)
    cmap = torch.tensor(sdf.shape, dtype=torch.long)
    idx = ((p - gp.limits_lo) / map_dim * cmap).floor().to(torch.int)
    idx = idx.clamp(torch.zeros(2, dtype=torch.int), (cmap - 1).to(torch.int))
    ix, iy = idx[..., 0].long(), idx[..., 1].long()
    return sdf[ix, iy], grad[ix, iy]


def extra_objects_sdf(p, gp: GuideParams):
    """ObjectField.compute_signed_distance_impl (primitives.py:554-572, identity pose) over the env's extra primitive fields:
    min over fields of their own min over spheres (|p - c| - r; an EMPTY MultiSphereField is 1 everywhere, :109-110) /
    boxes (the rounded box of the fixed objects, radius 0.15 x the smaller size).  Returns (sdf, d sdf / dp) by autograd."""
    pp = p.detach().clone().requires_grad_(True)
    fields = []
    if gp.extra_spheres is not None:
        if gp.extra_spheres.shape[0] == 0:
            fields.append(torch.ones_like(pp[..., 0]))
        else:
            d = torch.norm(pp.unsqueeze(-2) - gp.extra_spheres[:, :2], dim=-1) - gp.extra_spheres[:, 2]
            fields.append(torch.min(d, dim=-1)[0])
    if gp.extra_boxes is not None and gp.extra_boxes.shape[0] > 0:
        # MultiBoxField (alias of MultiRoundedBoxField, primitives.py:345) = the rounded box of :326-333 (radius 0.15 x the smaller size), as the fixed objects
        sizes = gp.extra_boxes[:, 2:]
        radius = torch.min(sizes, dim=-1)[0] * 0.15
        q = torch.abs(pp.unsqueeze(-2) - gp.extra_boxes[:, :2]) - sizes / 2 + radius.unsqueeze(-1)
        max_q = torch.amax(q, dim=-1)
        sdfs = torch.minimum(max_q, torch.zeros_like(max_q)) + torch.linalg.norm(torch.relu(q), dim=-1) - radius
        fields.append(torch.min(sdfs, dim=-1)[0])
    sdf = torch.min(torch.stack(fields, dim=-1), dim=-1)[0]
    with torch.enable_grad():
        g, = torch.autograd.grad(sdf.sum(), pp, allow_unused=True)
    return sdf.detach(), (torch.zeros_like(p) if g is None else g)


def grad_object_collision(xu, gp: GuideParams):
    """d/dx of CostCollision(df_collision_objects) (cost_functions.py:175-193, field_factor.py:24-48 with range
    [1,None]; distance_fields.py:110-135, :342-351): cost_b = sum_{t>=1} max_k relu(margin - sdf_k(p_t))."""
    p = xu[..., :2]
    m = gp.margin
    best = torch.zeros_like(p[..., 0])
    gbest = torch.zeros_like(p)
    fields = [] if gp.extra_only else [sdf_lookup(p, gp, k) for k in range(len(gp.sdf_grids))]
    if gp.extra_spheres is not None or gp.extra_boxes is not None:
        with torch.enable_grad():
            fields.append(extra_objects_sdf(p, gp))
    for s, gs in fields:
        v = torch.relu(m - s)
        take = v > best
        gbest = torch.where(take[..., None], -gs, gbest)
        best = torch.where(take, v, best)
    g = torch.zeros_like(xu)
    g[..., :2] = gbest
    g[..., 0, :] = 0.0                                # traj_range [1, None]
    return g


def grad_ws_boundaries(xu, gp: GuideParams):
    """d/dx of CostCollision(df_collision_ws_boundaries) (distance_fields.py:354-367): the four 'sdfs' are
    p - ws_min (x, y) and ws_max - p (x, y); cost = sum_{t>=1} max_4 relu(margin - d)."""
    p = xu[..., :2]
    m = gp.margin
    d = torch.cat((p - gp.ws_min, gp.ws_max - p), dim=-1)                 # [...,4]
    v = torch.relu(m - d)
    vmax, arg = v.max(dim=-1)
    dirs = torch.tensor([[-1.0, 0.0], [0.0, -1.0], [1.0, 0.0], [0.0, 1.0]])
    gpos = dirs[arg] * (vmax > 0)[..., None]
    g = torch.zeros_like(xu)
    g[..., :2] = gpos
    g[..., 0, :] = 0.0
    return g


def grad_gp_prior(xu, gp: GuideParams):
    """d/dx of CostGPTrajectory (cost_functions.py:532-542, gp_factor.py:4-65): e_t = s_{t+1} - Phi s_t,
    cost = sum_t e_t^T Qinv e_t  =>  g_t = w_{t-1} - Phi^T w_t,  w = 2 Qinv e."""
    dt = gp.dt
    qc = 1.0 / gp.sigma_gp ** 2
    m1, m2, m3 = 12.0 * dt ** -3.0 * qc, -6.0 * dt ** -2.0 * qc, 4.0 * dt ** -1.0 * qc
    p, v = xu[..., :2], xu[..., 2:4]
    ep = p[..., 1:, :] - (p[..., :-1, :] + dt * v[..., :-1, :])
    ev = v[..., 1:, :] - v[..., :-1, :]
    wp = 2.0 * (m1 * ep + m2 * ev)
    wv = 2.0 * (m2 * ep + m3 * ev)
    g = torch.zeros_like(xu)
    g[..., 1:, :2] += wp
    g[..., 1:, 2:4] += wv
    g[..., :-1, :2] -= wp
    g[..., :-1, 2:4] -= dt * wp + wv
    return g


def grad_constraint(xu, grp: ConstraintGroup):
    """d/dx of CostConstraint.eval (cost_functions.py:297-326), closed form:
    -sum_{c: t in [t0,t1), ||d||<=R} d/||d||, d = p_{b,t} - q_c."""
    p = xu[..., :2]                                                       # [B,H,2]
    H = p.shape[-2]
    t = torch.arange(H)
    g = torch.zeros_like(xu)
    n = grp.q.shape[0]
    CH = 256
    for c0 in range(0, n, CH):
        q = grp.q[c0:c0 + CH]
        tr = grp.t_range[c0:c0 + CH]
        r = grp.radius[c0:c0 + CH]
        active_t = (t[None, :] >= tr[:, 0:1]) & (t[None, :] < tr[:, 1:2])          # [n,H]
        d = p[None] - q[:, None, None, :]                                           # [n,B,H,2]
        dist = torch.linalg.norm(d, dim=-1)                                         # [n,B,H]
        act = active_t[:, None, :] & ~(dist > r[:, None, None])
        # a point exactly on a constraint centre: torch.norm's backward yields the zero sub-gradient there
        unit = torch.where(dist[..., None] > 0, d / dist[..., None].clamp_min(1e-38), torch.zeros_like(d))
        contrib = torch.where(act[..., None], unit, torch.zeros_like(d))
        g[..., :2] -= contrib.sum(0)
    return g


def clip_gradient(g, rule="norm", max_grad_norm=1.0, max_grad_value=0.1):
    """GuideManager.clip_gradient (guides.py:228-259): rule 'norm' | 'value' | None (clip_grad = False)."""
    if rule == "norm":
        return clip_grad_by_norm(g, max_grad_norm)
    if rule == "value":
        return torch.clip(g, -max_grad_value, max_grad_value)
    if rule is None:
        return g
    raise NotImplementedError(rule)


def guide_grad(x_norm, gp: GuideParams, cons: Sequence[ConstraintGroup] = (), clip_mode="reference",
               return_terms=False, clip_rule="norm", max_grad_value=0.1):
    """GuideManagerTrajectoriesWithVelocity.forward (guides.py:180-226), closed form: un-normalise, per-cost
    gradient w.r.t. the UN-normalised trajectory, per-point norm clip, zero rows 0 and H-1, weight, sum, negate.
    The result is added to the NORMALISED x by the caller (sample_functions.py:104) -- reproduced as is."""
    xu = unnormalize(x_norm, gp.norm_mins, gp.norm_maxs, clip_mode=clip_mode)
    terms = [(grad_object_collision(xu, gp), gp.weight_collision)]
    if not gp.extra_only:                             # (mpd.py:216-219: the extra-objects field is then the ONLY collision cost)
        terms.append((grad_ws_boundaries(xu, gp), gp.weight_collision))
    terms.append((grad_gp_prior(xu, gp), gp.weight_smoothness))
    for grp in cons:
        terms.append((grad_constraint(xu, grp), grp.weight))
    total = torch.zeros_like(x_norm)
    clipped_terms = []
    for g, w in terms:
        gc = clip_gradient(g, clip_rule, gp.max_grad_norm, max_grad_value).clone()
        gc[..., 0, :] = 0.0
        gc[..., -1, :] = 0.0
        clipped_terms.append(gc)
        total = total + w * gc
    if return_terms:
        return -1.0 * total, clipped_terms
    return -1.0 * total


# --------------------------------------------------------------------------------------------------------------
# The guide's DISCRETE decisions (tests/test_gpu_flips.py).  Everything in GuideManagerTrajectoriesWithVelocity.forward that is
# not continuous in x -- the nearest-cell SDF index (grid_map_sdf.py:84-114), the collision hinges and their arg max
# (distance_fields.py:110-135, :354-367), the active set of every CostConstraint (cost_functions.py:305-312) -- plus the two
# continuous switches (gradient clip active, guides.py:247-253; un-normalisation clip, normalization.py:161-163) as one record per
# (trajectory, support point).  guide_decisions() returns the sets guide_grad() takes; guide_grad_forced() evaluates the same guide
# with the sets GIVEN -- with the HIP kernel's own sets it tells whether a step that differs by more than the tolerance differs
# through a branch or through arithmetic.
# --------------------------------------------------------------------------------------------------------------


@dataclass
class GuideSets:
    cell: torch.Tensor          # [B,H] long   ix * ny + iy of the grid lookup
    obj_active: torch.Tensor    # [B,H] bool   margin - sdf > 0 for the winning field
    obj_win: torch.Tensor       # [B,H] long   index of the winning field (7: the env's extra objects)
    ws_active: torch.Tensor     # [B,H] bool
    ws_arg: torch.Tensor        # [B,H] long   0: x-min, 1: y-min, 2: x-max, 3: y-max
    clip: torch.Tensor          # [B,H,3+G] bool  gradient clip active on (objects, boundaries, GP, group 0, ...)
    unnorm_clip: torch.Tensor   # [B,H,4] bool
    cons: List[torch.Tensor]    # per group [B,H,S] bool: slot s active (slot_table)


def slot_table(grp: ConstraintGroup, H=64):
    """[S,H] long: the point of `grp` that sits in slot s at support point t (-1: none).  A point covers the integer t with
    t0 <= t < t1 (cost_functions.py:304-305); slot = its rank among the points covering t, in list order -- the layout of the
    product's time-bucketed table (mmd_pack_constraints / mmd_soft_constraints_from_paths)."""
    n = grp.q.shape[0]
    t0 = torch.ceil(grp.t_range[:, 0]).long().clamp_min(0)
    t1 = torch.ceil(grp.t_range[:, 1]).long().clamp_max(H)
    cover = (torch.arange(H)[None, :] >= t0[:, None]) & (torch.arange(H)[None, :] < t1[:, None])     # [n,H]
    rank = torch.cumsum(cover.long(), dim=0) - 1                                                    # [n,H]
    S = int(cover.sum(0).max()) if n else 0
    tab = torch.full((max(S, 1), H), -1, dtype=torch.long)
    pts, ts = torch.nonzero(cover, as_tuple=True)
    tab[rank[pts, ts], ts] = pts
    return tab


def _cell_index(p, gp: GuideParams):
    sdf = gp.sdf_grids[0][0]
    map_dim = torch.abs(gp.limits_hi - gp.limits_lo)
    cmap = torch.tensor(sdf.shape, dtype=torch.long)
    idx = ((p - gp.limits_lo) / map_dim * cmap).floor().to(torch.int)
    idx = idx.clamp(torch.zeros(2, dtype=torch.int), (cmap - 1).to(torch.int))
    return idx[..., 0].long() * int(cmap[1]) + idx[..., 1].long()


def _constraint_dist(p, grp: ConstraintGroup, tab):
    """dist [S,B,H] of every (slot, trajectory, support point) and the unit vectors d / ||d|| [S,B,H,2] (zero at d = 0)."""
    q = grp.q[tab.clamp_min(0)]                                  # [S,H,2]
    d = p[None] - q[:, None]                                     # [S,B,H,2]
    dist = torch.linalg.norm(d, dim=-1)
    unit = torch.where(dist[..., None] > 0, d / dist[..., None].clamp_min(1e-38), torch.zeros_like(d))
    return dist, unit


def guide_decisions(x_norm, gp: GuideParams, cons: Sequence[ConstraintGroup] = (), clip_mode="always") -> GuideSets:
    """The sets guide_grad(x_norm, gp, cons, clip_mode) takes (norm clipping rule; the env's extra objects are not covered)."""
    assert gp.extra_spheres is None and gp.extra_boxes is None and not gp.extra_only
    H = x_norm.shape[-2]
    xu = unnormalize(x_norm, gp.norm_mins, gp.norm_maxs, clip_mode=clip_mode)
    p = xu[..., :2]
    m = gp.margin
    best = torch.zeros_like(p[..., 0])
    win = torch.zeros_like(best, dtype=torch.long)
    for k in range(len(gp.sdf_grids)):
        v = torch.relu(m - sdf_lookup(p, gp, k)[0])
        take = v > best
        win = torch.where(take, torch.full_like(win, k), win)
        best = torch.where(take, v, best)
    d = torch.cat((p - gp.ws_min, gp.ws_max - p), dim=-1)
    vmax, arg = torch.relu(m - d).max(dim=-1)
    _, terms_raw = _raw_terms(xu, gp, cons)
    clip = torch.stack([torch.linalg.norm(g + 1e-6, dim=-1) > gp.max_grad_norm for g in terms_raw], dim=-1)
    masks = []
    for grp in cons:
        tab = slot_table(grp, H)
        dist, _ = _constraint_dist(p, grp, tab)
        r = grp.radius[tab.clamp_min(0)]                          # [S,H]
        act = (tab >= 0)[:, None, :] & ~(dist > r[:, None, :])
        masks.append(act.permute(1, 2, 0).contiguous())
    return GuideSets(cell=_cell_index(p, gp) if gp.sdf_grids else torch.zeros_like(win), obj_active=best > 0, obj_win=win,
                     ws_active=vmax > 0, ws_arg=arg, clip=clip,
                     unnorm_clip=(x_norm.abs() > 1) if clip_mode == "always" else torch.zeros_like(x_norm, dtype=torch.bool),
                     cons=masks)


def _raw_terms(xu, gp: GuideParams, cons):
    terms = [grad_object_collision(xu, gp), grad_ws_boundaries(xu, gp), grad_gp_prior(xu, gp)]
    for grp in cons:
        terms.append(grad_constraint(xu, grp))
    weights = [gp.weight_collision, gp.weight_collision, gp.weight_smoothness] + [grp.weight for grp in cons]
    return weights, terms


def guide_grad_forced(x_norm, gp: GuideParams, cons: Sequence[ConstraintGroup], sets: GuideSets):
    """guide_grad with every decision taken from `sets` instead of from x: the clip of the un-normalisation where
    sets.unnorm_clip says so, the SDF value / gradient of cell sets.cell and field sets.obj_win iff sets.obj_active, the
    boundary direction sets.ws_arg iff sets.ws_active, the constraint points of sets.cons, the norm clip where sets.clip."""
    H = x_norm.shape[-2]
    xc = torch.where(sets.unnorm_clip, torch.clip(x_norm, -1, 1), x_norm)
    xu = (xc + 1) / 2.0 * (gp.norm_maxs - gp.norm_mins) + gp.norm_mins
    p = xu[..., :2]
    terms = []
    g = torch.zeros_like(xu)
    if gp.sdf_grids:
        grads = torch.stack([gr.reshape(-1, 2) for _, gr in gp.sdf_grids], dim=0)       # [K, nx*ny, 2]
        gsel = grads[sets.obj_win.clamp_max(len(gp.sdf_grids) - 1), sets.cell]             # [B,H,2]
        g[..., :2] = torch.where(sets.obj_active[..., None], -gsel, torch.zeros_like(gsel))
    g[..., 0, :] = 0.0
    terms.append((g, gp.weight_collision))
    dirs = torch.tensor([[-1.0, 0.0], [0.0, -1.0], [1.0, 0.0], [0.0, 1.0]])
    g = torch.zeros_like(xu)
    g[..., :2] = dirs[sets.ws_arg] * sets.ws_active[..., None]
    g[..., 0, :] = 0.0
    terms.append((g, gp.weight_collision))
    terms.append((grad_gp_prior(xu, gp), gp.weight_smoothness))
    for grp, act in zip(cons, sets.cons):
        tab = slot_table(grp, H)
        _, unit = _constraint_dist(p, grp, tab)                                          # [S,B,H,2]
        a = act.permute(2, 0, 1)[: tab.shape[0]] & (tab >= 0)[:, None, :]
        g = torch.zeros_like(xu)
        g[..., :2] = -torch.where(a[..., None], unit, torch.zeros_like(unit)).sum(0)
        terms.append((g, grp.weight))
    total = torch.zeros_like(x_norm)
    for k, (g, w) in enumerate(terms):
        n = torch.linalg.norm(g + 1e-6, dim=-1, keepdims=True)
        gc = torch.where(sets.clip[..., k:k + 1], gp.max_grad_norm / n * g, g).clone()
        gc[..., 0, :] = 0.0
        gc[..., -1, :] = 0.0
        total = total + w * gc
    return -1.0 * total


def guide_grad_dense_autograd(x_norm, gp: GuideParams, cons: Sequence[ConstraintGroup] = ()):
    """Reference-SHAPED evaluation of the same guide: the dense (n,B,H,2) CostConstraint broadcast
    (cost_functions.py:297-326) and one torch.autograd.grad per cost term (guides.py:207-211).  Used (a) to
    cross-check the closed forms above without the reference, (b) as the CPU baseline in bench.py, because this
    is what the reference's CPU path actually pays for."""
    x = x_norm.clone()
    with torch.enable_grad():
        x.requires_grad_(True)
        xu = unnormalize(x, gp.norm_mins, gp.norm_maxs)
        p = xu[..., :2]
        m = gp.margin
        costs = []
        # objects
        ps = p[:, 1:, :]
        vals = []
        for k in range(len(gp.sdf_grids)):
            s, gs = sdf_lookup(ps.detach(), gp, k)
            s = s + (ps * gs).sum(-1) - (ps.detach() * gs).sum(-1)           # surrogate, grid_map_sdf.py:108-112
            vals.append(torch.relu(-(s - m)))
        costs.append((torch.stack(vals, dim=-1).max(-1)[0].sum(1), gp.weight_collision))
        # workspace boundaries
        d = torch.cat((ps - gp.ws_min, gp.ws_max - ps), dim=-1)
        costs.append((torch.relu(-(d - m)).max(-1)[0].sum(1), gp.weight_collision))
        # GP prior
        dt = gp.dt
        I2, Z2 = torch.eye(2), torch.zeros(2, 2)
        phi = torch.cat((torch.cat((I2, dt * I2), 1), torch.cat((Z2, I2), 1)), 0)
        qc = torch.eye(2) / gp.sigma_gp ** 2
        qinv = torch.cat((torch.cat((12.0 * dt ** -3.0 * qc, -6.0 * dt ** -2.0 * qc), -1),
                          torch.cat((-6.0 * dt ** -2.0 * qc, 4.0 * dt ** -1.0 * qc), -1)), -2)
        s1 = xu[:, :-1].unsqueeze(-1)
        s2 = xu[:, 1:].unsqueeze(-1)
        err = s2 - phi @ s1
        costs.append(((err.transpose(2, 3) @ qinv.reshape(1, 1, 4, 4) @ err).sum(1).squeeze(), gp.weight_smoothness))
        # constraints (dense broadcast)
        for grp in cons:
            H = p.shape[1]
            mask = torch.arange(H).unsqueeze(0).unsqueeze(0)
            mask = (mask >= grp.t_range[:, 0].view(-1, 1, 1)) & (mask < grp.t_range[:, 1].view(-1, 1, 1))
            qp = p.unsqueeze(0).expand(grp.q.shape[0], -1, -1, -1)
            qpm = qp * mask.unsqueeze(-1)
            dist = torch.norm(qpm - grp.q.view(-1, 1, 1, 2), dim=-1)
            dist = torch.where(dist > grp.radius.view(-1, 1, 1), torch.zeros_like(dist), dist)
            costs.append(((grp.radius.view(-1, 1, 1) - dist).sum(dim=-1).sum(), grp.weight))
        grad = 0
        for c, w in costs:
            g = torch.autograd.grad([c.sum()], [xu], retain_graph=True)[0]
            g = clip_grad_by_norm(g, gp.max_grad_norm)
            g[..., 0, :] = 0.0
            g[..., -1, :] = 0.0
            grad = grad + w * g
    return -1.0 * grad


# --------------------------------------------------------------------------------------------------------------
# A2, A3, A5  DDPM sampling
# --------------------------------------------------------------------------------------------------------------


def apply_hard_conditioning(x, hard_conds):
    """sample_functions.py:8-14.  hard_conds {row: [D] or [B,D]}."""
    for t, val in hard_conds.items():
        x[:, t, :] = val
    return x


def ddpm_sample_step(sd, tb, x, hard_conds, i, *, guide=None, n_guide_steps=1, t_start_guide=float("inf"),
                     noise=None, noise_std_extra=1.0, n_levels=None, eps_rel_perturb=None, scale_grad_by_std=False,
                     predict_epsilon=True):
    """ddpm_sample_fn (sample_functions.py:40-86) + p_mean_variance / predict_start_from_noise / q_posterior
    (diffusion_model_base.py:126-160) with predict_epsilon=True, clip_denoised=True.  `i` is the loop index (may be
    negative: t := 0, sample_functions.py:53-54); `noise` [B,H,D] is the injected randn_like draw; `guide` is a
    callable x_norm -> grad.  `eps_rel_perturb` [B,H,D] (tests only): the UNet output is multiplied by 1 + it -- the probe of
    how far a guided step moves under a rounding-sized change of eps (the sensitivity the golden chains store as `sens`)."""
    B = x.shape[0]
    t = max(i, 0)
    tt = torch.full((B,), t, dtype=torch.long)
    eps = unet_forward(sd, x, tt, n_levels)
    if eps_rel_perturb is not None:
        eps = eps * (1.0 + eps_rel_perturb)
    if predict_epsilon:
        x_recon = tb["sqrt_recip_alphas_cumprod"][t] * x - tb["sqrt_recipm1_alphas_cumprod"][t] * eps
    else:
        x_recon = eps                                                     # the model predicts x0 directly, :131-141
    x_recon = x_recon.clamp(-1.0, 1.0)
    mean = tb["posterior_mean_coef1"][t] * x_recon + tb["posterior_mean_coef2"][t] * x
    model_std = torch.exp(0.5 * tb["posterior_log_variance_clipped"][t])
    model_var = torch.exp(tb["posterior_log_variance_clipped"][t])
    x = mean
    if guide is not None and i < t_start_guide:
        for _ in range(n_guide_steps):                                    # guide_gradient_steps, :89-107
            x = x + (model_var * guide(x) if scale_grad_by_std else guide(x))
            x = apply_hard_conditioning(x, hard_conds)
    if noise is None:
        noise = torch.zeros_like(x)
    if t == 0:
        noise = torch.zeros_like(x)                                       # noise[t == 0] = 0, :76
    return x + model_std * noise * noise_std_extra


def p_sample_loop(sd, tb, x_init, hard_conds, n_diffusion_steps, step_noise, *, guide=None, n_guide_steps=20,
                  t_start_guide=float("inf"), noise_std_extra=0.5, n_diffusion_steps_without_noise=0, n_levels=None,
                  scale_grad_by_std=False, predict_epsilon=True):
    """GaussianDiffusionModel.p_sample_loop (diffusion_model_base.py:162-211), with the torch.randn draws injected:
    x_init [B,H,D] is x_T (or the warm start), step_noise [n_steps,B,H,D] one draw per loop iteration in order.
    Returns chain [n_steps+1, B,H,D] (chain[0] = conditioned x_init, chain[-1] = result)."""
    x = apply_hard_conditioning(x_init.clone(), hard_conds)
    chain = [x]
    k = 0
    for i in reversed(range(-n_diffusion_steps_without_noise, n_diffusion_steps)):
        x = ddpm_sample_step(sd, tb, x, hard_conds, i, guide=guide, n_guide_steps=n_guide_steps,
                             t_start_guide=t_start_guide, noise=step_noise[k], noise_std_extra=noise_std_extra,
                             n_levels=n_levels, scale_grad_by_std=scale_grad_by_std, predict_epsilon=predict_epsilon)
        x = apply_hard_conditioning(x, hard_conds)
        chain.append(x)
        k += 1
    return torch.stack(chain, dim=0)


def ddim_times(n_diffusion_steps):
    """Time pairs of ddim_sample (diffusion_model_base.py:225-235): sampling_timesteps = T // 5, eta = 0."""
    S = n_diffusion_steps // 5
    times = torch.linspace(0, n_diffusion_steps - 1, steps=S + 1)
    times = torch.cat((torch.tensor([-1.0]), times))
    return list(reversed(times.int().tolist()))


def ddim_sample(sd, tb, x_init, hard_conds, n_diffusion_steps, *, guide=None, t_start_guide=float("inf"), n_levels=None,
                predict_epsilon=True):
    """GaussianDiffusionModel.ddim_sample (diffusion_model_base.py:213-290), eta = 0 (sigma = 0: the per-step randn_like draw
    is multiplied by 0).  predict_epsilon=False: the network output is x_start and pred_noise = (a x - out) / b
    (predict_noise_from_start, :114-124).  x_init [B,H,D] is the injected x_T.  Quirk kept: the guide runs ONE
    gradient step per sampling step -- ddim_sample binds `n_guide_steps` itself and forwards only **sample_kwargs, so
    guide_gradient_steps (sample_functions.py:89) runs with its default n_guide_steps=1.  x_start is NOT clamped here.
    Returns chain [n_pairs+1, B,H,D]."""
    times = ddim_times(n_diffusion_steps)
    x = apply_hard_conditioning(x_init.clone(), hard_conds)
    chain = [x]
    B = x.shape[0]
    for time, time_next in zip(times[:-1], times[1:]):
        eps = unet_forward(sd, x, torch.full((B,), time, dtype=torch.long), n_levels)
        if predict_epsilon:
            x_start = tb["sqrt_recip_alphas_cumprod"][time] * x - tb["sqrt_recipm1_alphas_cumprod"][time] * eps
        else:
            x_start = eps
            eps = (tb["sqrt_recip_alphas_cumprod"][time] * x - x_start) / tb["sqrt_recipm1_alphas_cumprod"][time]
        if time_next < 0:
            x = apply_hard_conditioning(x_start, hard_conds)
            chain.append(x)
            break
        alpha_next = tb["alphas_cumprod"][time_next]
        c = (1 - alpha_next).sqrt()
        x = x_start * alpha_next.sqrt() + c * eps
        if guide is not None and time_next < t_start_guide:
            x = x + guide(x)
            x = apply_hard_conditioning(x, hard_conds)
        x = apply_hard_conditioning(x, hard_conds)
        chain.append(x)
    return torch.stack(chain, dim=0)


def q_sample(tb, x_start, t, noise):
    """diffusion_model_base.py:425-433."""
    return tb["sqrt_alphas_cumprod"][t] * x_start + tb["sqrt_one_minus_alphas_cumprod"][t] * noise


# --------------------------------------------------------------------------------------------------------------
# A13  ensemble cross-conditioning
# --------------------------------------------------------------------------------------------------------------


def apply_cross_conditioning(x: Dict[int, torch.Tensor], conditions, transforms):
    """sample_functions.py:17-31.  conditions {(m1,m2): (ind1,ind2)}, transforms {m: [2]}."""
    for (m1, m2), (ind1, ind2) in conditions.items():
        rel = transforms[m2] - transforms[m1]
        D = x[m1].shape[2]
        if D > rel.shape[0]:
            rel = torch.cat([rel, torch.zeros(D - rel.shape[0])])
        boundary = rel / torch.norm(rel, keepdim=True)
        boundary[boundary == 0] = 1e6
        x[m1][:, ind1, :] = torch.min(x[m2][:, ind2, :] + rel, boundary)
        x[m2][:, ind2, :] = torch.max(x[m1][:, ind1, :] - rel, -boundary)
    return x


def ensemble_p_sample_loop(sds, tb, x_init: Dict[int, torch.Tensor], hard_conds, cross_conds, transforms, n_diffusion_steps,
                           step_noise, guides=None, n_guide_steps=20, t_start_guide=float("inf"), noise_std_extra=0.5,
                           n_diffusion_steps_without_noise=0):
    """DiffusionsEnsemble.p_sample_loop (mmd/models/diffusion_models/diffusion_ensemble.py:55-106) with injected noise: per outer
    step the tiles step IN ORDER (ddpm_sample_fn, hard conditioning), each followed by the cross conditioning of ALL pairs.
    sds / guides: per tile; x_init {m: [B,H,D]}; step_noise [n_steps, K, B,H,D].  Returns (x, chains {m: [n_steps+1, B,H,D]}).
    The chains are collected the way the reference collects them -- `chains[m].append(x[m])` stores the tensor OBJECT, and
    apply_cross_conditioning writes into x[m] in place -- so row k of a tile that has not stepped yet in outer step k+1 also carries
    the boundary rows stitched after the EARLIER tiles' steps of step k+1 (visible from 3 tiles on: re-stitching two not-yet-stepped
    neighbours is not idempotent when the later tile's first row was fresh from its own step); the dynamics are unaffected, the
    last row is clean."""
    keys = list(x_init.keys())
    x = {m: apply_hard_conditioning(x_init[m].clone(), hard_conds.get(m, {})) for m in keys}
    x = apply_cross_conditioning(x, cross_conds, transforms)
    chains = {m: [x[m]] for m in keys}
    for k, i in enumerate(reversed(range(-n_diffusion_steps_without_noise, n_diffusion_steps))):
        for j, m in enumerate(keys):
            guide = guides[m] if guides is not None else None
            x[m] = ddpm_sample_step(sds[m], tb, x[m], hard_conds.get(m, {}), i, guide=guide, n_guide_steps=n_guide_steps,
                                    t_start_guide=t_start_guide, noise=step_noise[k, j], noise_std_extra=noise_std_extra)
            x[m] = apply_hard_conditioning(x[m], hard_conds.get(m, {}))
            x = apply_cross_conditioning(x, cross_conds, transforms)
        for m in keys:
            chains[m].append(x[m])
    return x, {m: torch.stack(v, dim=0) for m, v in chains.items()}


def ensemble_warm_start(tb, seed_trajectory_b, n_noising_steps, q_noise, transforms, horizon=64):
    """The warm start of DiffusionsEnsemble.run_local_inference (diffusion_ensemble.py:265-300 + p_sample_loop :66-72): the seed batch
    [B, K*horizon, D] (GLOBAL frame, as the caller hands it over) is forward-noised as a whole (q_sample with ONE draw of the full
    shape; None: taken as it is), tile m takes rows [m*horizon, (m+1)*horizon) and moves its positions into the tile frame."""
    noised = seed_trajectory_b if n_noising_steps is None else q_sample(tb, seed_trajectory_b, n_noising_steps, q_noise)
    x = {}
    for m in transforms:
        xm = noised[:, m * horizon:(m + 1) * horizon, :].clone()
        xm[:, :, :2] -= torch.as_tensor(transforms[m])
        x[m] = xm
    return x


def soft_constraints_from_paths(paths, agent_id, radius, weight, start_times=None):
    """CBS.create_soft_constraints_from_other_agents_paths (mmd/planners/multi_agent/cbs.py:468-508) for equal start
    times: every other robot's position at t (1 <= t <= H-1) constrains this robot at [t, t+1)."""
    N, H, _ = paths.shape
    q, tr = [], []
    for j in range(N):
        if j == agent_id:
            continue
        for t in range(H):
            if 1 <= t <= H - 1:
                q.append(paths[j, t])
                tr.append((t, t + 1))
    return ConstraintGroup(q=torch.stack(q), t_range=torch.tensor(tr, dtype=torch.float32),
                           radius=torch.full((len(q),), radius), weight=weight)


# --------------------------------------------------------------------------------------------------------------
# §8f-1  multi-agent layer next to the sampler
# --------------------------------------------------------------------------------------------------------------


def check_rr_collisions(robot_q, radius=0.05):
    """RobotPlanarDisk.check_rr_collisions (deps/torch_robotics/torch_robotics/robots/robot_planar_disk.py:173-203).
    robot_q [..., n_robots, 2] -> (collisions [..., n, n] bool, midpoints [..., n, n, 2] with NaN where no collision)."""
    margin = 2.1 * radius
    p1, p2 = robot_q.unsqueeze(-2), robot_q.unsqueeze(-3)
    coll = torch.norm(p1 - p2, dim=-1) < margin
    coll = coll & ~torch.eye(coll.shape[-1], dtype=coll.dtype)
    mid = (p1 + p2) / 2
    mid = mid * coll.unsqueeze(-1)
    mid[~coll.unsqueeze(-1).expand_as(mid)] = float("nan")
    return coll, mid


def count_collisions_with_others(samples_pos, paths, self_idx, radius=0.05):
    """#{(t, j != self): ||x_b(t) - p_j(t)|| < 2.1 r} per sample b -- what CBS's 'least_collisions' scan (cbs.py:446-458)
    ranks by: get_conflicts(state with sample b) has nonzero(collisions) = const + 2 * this count."""
    others = torch.cat((paths[:self_idx], paths[self_idx + 1:]))                    # [N-1,H,2]
    d = torch.norm(samples_pos[:, None] - others[None], dim=-1)                     # [B,N-1,H]
    return (d < 2.1 * radius).sum(dim=(1, 2))


# --------------------------------------------------------------------------------------------------------------
# 8f-2  post-sampling selection (the step right after the hot path), pinned by golden g9 / g12
# --------------------------------------------------------------------------------------------------------------


def interpolate_traj_via_points(trajs, num_interpolation=10):
    """deps/torch_robotics/torch_robotics/trajectory/utils.py:73-86."""
    H, D = trajs.shape[-2:]
    if num_interpolation <= 0:
        return trajs
    alpha = torch.linspace(0, 1, num_interpolation + 2).type_as(trajs)[1:num_interpolation + 1]
    alpha = alpha.view((1,) * len(trajs.shape[:-1]) + (-1, 1))
    out = trajs[..., 0:H - 1, None, :] * alpha + trajs[..., 1:H, None, :] * (1 - alpha)
    return out.view(trajs.shape[:-2] + (-1, D))


def compute_collision(pos, gp: GuideParams, margin=None):
    """occupancy branch of PlanningTask._compute_collision_or_cost (tasks.py:141-143, :204-232): a point collides iff a
    fixed-object SDF (nearest grid cell, distance_fields.py:318-326, :342-351) or a workspace-boundary distance
    (:361-367) is below `margin` (default: collision_margins + cutoff_margin, :320)."""
    margin = gp.margin if margin is None else margin
    coll = torch.zeros(pos.shape[:-1], dtype=torch.bool)
    for k in range(len(gp.sdf_grids)):
        coll = coll | (sdf_lookup(pos, gp, k)[0] < margin)
    if gp.extra_spheres is not None or gp.extra_boxes is not None:      # env.get_df_obj_list(): fixed grid + obj_extra_list
        with torch.enable_grad():
            coll = coll | (extra_objects_sdf(pos, gp)[0] < margin)
    ws = torch.cat((pos - gp.ws_min, gp.ws_max - pos), dim=-1)
    return coll | (ws < margin).any(dim=-1)


def get_trajs_collision_and_free(trajs, gp: GuideParams, num_interpolation=5, all_free=False, q_min=(-1.0, -1.0),
                                 q_max=(1.0, 1.0)):
    """tasks.py:236-311 for [B,H,D] batches (margin = robot radius, :251-253).  Returns (coll, coll_idxs, free, free_idxs,
    waypoint_collisions); all_free: PlanningTaskEnsemble (tasks_ensemble.py:271-277)."""
    B = trajs.shape[0]
    if all_free:
        coll_pts = torch.zeros(B, 1, dtype=torch.bool)
    else:
        coll_pts = compute_collision(interpolate_traj_via_points(trajs, num_interpolation)[..., :2], gp, gp.robot_radius)
    in_coll = coll_pts.any(dim=-1)
    pos = trajs[..., :2]
    inside = (pos >= torch.tensor(q_min)) & (pos <= torch.tensor(q_max))
    free_mask = ~in_coll & inside.all(dim=-1).all(dim=-1)
    if all_free:
        free_mask = torch.ones_like(in_coll)
    free_idxs = torch.argwhere(free_mask)
    coll_idxs = torch.argwhere(~free_mask)
    free = trajs[free_mask] if free_mask.any() else None
    coll = trajs[~free_mask] if (~free_mask).any() else None
    return coll, coll_idxs, free, free_idxs, coll_pts


def compute_path_length(trajs):
    """trajectory/metrics.py:7-15."""
    return torch.linalg.norm(torch.diff(trajs[..., :2], dim=-2), dim=-1).sum(-1)


def compute_smoothness(trajs):
    """trajectory/metrics.py:30-39."""
    return torch.linalg.norm(torch.diff(trajs[..., 2:4], dim=-2), dim=-1).sum(-1)


def compute_variance_waypoints(trajs):
    """trajectory/metrics.py:17-27."""
    pos = trajs[..., :2]
    total = 0.0
    for via in pos.permute(1, 0, 2):
        d = torch.cdist(via, via, p=2)
        total = total + torch.var(torch.triu(d, diagonal=1).view(-1))
    return total


def smooth_trajs(trajs, window_size=10, poly_order=2):
    """mmd/common/trajectory_utils.py:31-40 (scipy savgol_filter along the horizon, mode='interp')."""
    from scipy.signal import savgol_filter
    return torch.from_numpy(savgol_filter(trajs.numpy(), window_size, poly_order, axis=1))


# --------------------------------------------------------------------------------------------------------------
# MPDEnsemble.__call__ post-processing (multi-tile planner output)
# --------------------------------------------------------------------------------------------------------------


def ensemble_planner_output(chains_norm: Dict[int, torch.Tensor], gps: Dict[int, GuideParams], transforms, mins, maxs):
    """mmd/planners/single_agent/mpd_ensemble.py:385-429 + PlanningTaskEnsemble.get_traj_unnormalized / combine_trajs
    (deps/torch_robotics/torch_robotics/tasks/tasks_ensemble.py:79-88, 162-225): every tile's chain [T+2,B,H,D] is un-normalised
    and split into colliding / free samples against the tile's OWN map in the tile frame (PlanningTask.get_trajs_collision_and_free,
    tasks.py:236-311); a sample is free iff no tile reports it; positions move to the global frame and the tiles are concatenated
    along the horizon; costs, best sample and waypoint variance are those of the concatenated free samples; trajs_final is the
    SavGol-smoothed last row.  Returned as a dict of the PlannerOutput fields.  (tasks_ensemble.py:190 indexes the HORIZON axis of
    trajs_final with the colliding sample indices -- `trajs_final[:, idxs]` -- that is reproduced.)"""
    keys = list(chains_norm.keys())
    iters, tile_coll = {}, {}
    for m in keys:
        tr = unnormalize(chains_norm[m], mins, maxs)
        _, coll_idxs, _, _, _ = get_trajs_collision_and_free(tr[-1], gps[m])
        tile_coll[m] = set(int(i) for i in coll_idxs.reshape(-1))
        tr = tr.clone()
        tr[..., :2] += torch.as_tensor(transforms[m])
        iters[m] = tr
    trajs_iters = torch.cat([iters[m] for m in keys], dim=-2)
    final = trajs_iters[-1]
    B = final.shape[0]
    coll = [b for b in range(B) if any(b in tile_coll[m] for m in keys)]
    free = [b for b in range(B) if b not in coll]
    out = dict(trajs_iters=trajs_iters, trajs_final=smooth_trajs(final), tile_coll_idxs={m: sorted(tile_coll[m]) for m in keys},
               trajs_final_coll_idxs=torch.tensor(coll, dtype=torch.long), trajs_final_free_idxs=torch.tensor(free, dtype=torch.long),
               trajs_final_coll=final[:, coll] if coll else torch.tensor([]), trajs_final_free=final[free] if free else torch.tensor([]),
               success_free_trajs=1 if free else 0, fraction_free_trajs=len(free) / B, collision_intensity_trajs=1 - len(free) / B,
               idx_best_traj=None, traj_final_free_best=None, cost_best_free_traj=None, cost_smoothness=None, cost_path_length=None,
               cost_all=None, variance_waypoint_trajs_final_free=None)
    if free:
        f = out["trajs_final_free"]
        out["cost_smoothness"], out["cost_path_length"] = compute_smoothness(f), compute_path_length(f)
        out["cost_all"] = out["cost_smoothness"] + out["cost_path_length"]
        best = int(torch.argmin(out["cost_all"]))
        out["idx_best_traj"] = out["trajs_final_free_idxs"][best]
        out["traj_final_free_best"] = f[best]
        out["cost_best_free_traj"] = out["cost_all"][best]
        out["variance_waypoint_trajs_final_free"] = compute_variance_waypoints(f)
    return out
