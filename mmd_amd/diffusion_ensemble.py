"""DiffusionsEnsemble: host mirror of reference mmd/models/diffusion_models/diffusion_ensemble.py:37-313 (composition
of tile models along the horizon).  The whole sampling loop is ONE C-ABI call (mmd_p_sample_loop_ensemble): per outer
step every tile runs its UNet + fused DDPM/guide kernel in order, each followed by the cross-conditioning kernel that
stitches the tile boundaries (sample_functions.py:17-31); chains are written on the device."""
import ctypes as C
from copy import deepcopy
from typing import Dict, Tuple

import torch

from . import _lib
from .diffusion_model import GaussianDiffusionModel, ddpm_sample_fn, next_stream_seed

HORIZON = 64     # mmd/config/mmd_params.py:34


def _rel_boundary(transforms, m1, m2, D):
    """relative transform padded to D and its clamp boundary (sample_functions.py:19-27)."""
    rel = (torch.as_tensor(transforms[m2]) - torch.as_tensor(transforms[m1])).float().cpu()
    if D > rel.shape[0]:
        rel = torch.cat([rel, torch.zeros(D - rel.shape[0])])
    boundary = rel / torch.norm(rel, keepdim=True)
    boundary[boundary == 0] = 1e6
    return [float(v) for v in rel], [float(v) for v in boundary]


def apply_cross_conditioning(x: Dict[int, torch.Tensor], conditions, transforms):
    """sample_functions.py:17-31 on device tensors (in place)."""
    lib = _lib.load()
    for (m1, m2), (ind1, ind2) in conditions.items():
        rel, boundary = _rel_boundary(transforms, m1, m2, x[m1].shape[2])
        _lib.launch("mmd_cross_condition", x[m1], _lib.require_gpu(x[m1], "x[m1]"), _lib.require_gpu(x[m2], "x[m2]"),
                                           int(ind1) % x[m1].shape[1], int(ind2) % x[m2].shape[1],
                                           (C.c_float * 4)(*rel), (C.c_float * 4)(*boundary), x[m1].shape[0])
    return x


class DiffusionsEnsemble:
    def __init__(self, models: Dict[int, GaussianDiffusionModel], transforms: Dict[int, torch.Tensor],
                 context_model=None, **kwargs):
        self.models = models
        assert len(set(m.n_diffusion_steps for m in models.values())) == 1
        self.n_diffusion_steps = models[0].n_diffusion_steps
        assert len(set(m.predict_epsilon for m in models.values())) == 1
        self.transforms = transforms

    @torch.no_grad()
    def p_sample_loop(self, shape, hard_conds, cross_conds, n_diffusion_steps=None, contexts=None, return_chain=False,
                      sample_fn=ddpm_sample_fn, n_diffusion_steps_without_noise=0, warm_start_path_b=None,
                      x_init=None, step_noise=None, device="cuda", seed=None, n_robots=1, robot_seeds=None,
                      robot_transforms=None, **sample_kwargs):
        """diffusion_ensemble.py:55-106.  `x_init` {m: [B,H,D]} / `step_noise` [n_steps, n_models, B,H,D] inject the
        Gaussian draws (parity tests); otherwise Philox (`seed` + tile index, or the global stream counter).
        Batched form (planners.plan_batched: R independent MPDEnsemble calls as ONE launch sequence): `n_robots` = R, shape[0] =
        R * samples, hard conditions [R, D] per row, `robot_seeds` [R] (call r's tile j draws with robot_seeds[r] + j, as its own
        call would) and `robot_transforms` [R] = each call's {tile: transform} (the cross conditioning then takes its relative
        transform per robot)."""
        if sample_fn is not ddpm_sample_fn:
            raise NotImplementedError("only ddpm_sample_fn")
        if contexts is not None:
            raise NotImplementedError("contexts")
        lib = _lib.load()
        keys = list(self.models.keys())
        K = len(keys)
        pos = {m: j for j, m in enumerate(keys)}
        device = torch.device(device)
        B, H, D = shape
        if B % n_robots:
            raise ValueError(f"batch {B} is not a multiple of n_robots = {n_robots}")
        n_total = n_diffusion_steps + n_diffusion_steps_without_noise
        kw = sample_kwargs["sample_kwargs"]
        x, chains, keep = {}, {}, []
        init_noise = 0
        for m in keys:
            if warm_start_path_b is not None:
                x[m] = warm_start_path_b[:, m * HORIZON:(m + 1) * HORIZON, :].to(device=device, dtype=torch.float32).contiguous().clone()
                if robot_transforms is None:
                    x[m][:, :, :2] -= torch.as_tensor(self.transforms[m]).to(x[m].device)
                else:                                         # every call's seed batch into ITS tile frame
                    offs = torch.stack([torch.as_tensor(tr[m], dtype=torch.float32) for tr in robot_transforms]).to(device)
                    x[m][:, :, :2] -= offs.repeat_interleave(B // n_robots, 0)[:, None, :]
            elif x_init is not None:
                x[m] = x_init[m].to(device=device, dtype=torch.float32).contiguous().clone()
            else:
                x[m] = torch.empty(shape, dtype=torch.float32, device=device)
                init_noise = 1
            hard_conds.setdefault(m, {})
        tiles = (_lib.EnsembleTile * K)()
        ws_bytes = 0
        for j, m in enumerate(keys):
            model, skw = self.models[m], kw[m]
            guide = skw.get("guide")
            hard, mask = model._hard_tensor(hard_conds[m], n_robots, H, device, D)
            seeds_dev = None if robot_seeds is None else model.robot_seed_tensor([int(v) + j for v in robot_seeds], n_robots, device)
            sd = model._sampler_desc(skw.get("n_guide_steps", 1), skw.get("t_start_guide", float("inf")),
                                     skw.get("noise_std_extra_schedule_fn"), mask,
                                     scale_grad_by_std=skw.get("scale_grad_by_std", False), robot_seeds=seeds_dev)
            gd = guide.desc() if guide is not None else None
            noise_m = None
            if step_noise is not None:
                noise_m = step_noise[:, j].to(device=device, dtype=torch.float32).contiguous()
                assert noise_m.shape == (n_total,) + tuple(shape)
            chains[m] = torch.empty((n_total + 1,) + tuple(shape), dtype=torch.float32, device=device) if return_chain else None
            keep.append((hard, sd, gd, noise_m, seeds_dev))
            t = tiles[j]
            t.unet = model.model.handle(model.n_diffusion_steps, device)
            t.sampler = C.pointer(sd)
            t.guide = C.pointer(gd) if gd is not None else None
            t.x_dev, t.hard_dev = x[m].data_ptr(), hard.data_ptr()
            t.step_noise_dev = noise_m.data_ptr() if noise_m is not None else None
            t.chain_dev = chains[m].data_ptr() if chains[m] is not None else None
            t.seed = 0 if robot_seeds is not None else ((int(seed) + j) if seed is not None else next_stream_seed(model.seed))
            ws_bytes = max(ws_bytes, lib.mmd_sampler_workspace_bytes(t.unet, B))
        cc = (_lib.CrossCond * max(len(cross_conds), 1))()
        for c, ((m1, m2), (ind1, ind2)) in enumerate(cross_conds.items()):
            rel, boundary = _rel_boundary(self.transforms, m1, m2, D)
            cc[c].m1, cc[c].m2, cc[c].ind1, cc[c].ind2 = pos[m1], pos[m2], int(ind1) % H, int(ind2) % H
            cc[c].rel[:], cc[c].boundary[:] = rel, boundary
            if robot_transforms is not None:
                table = torch.tensor([sum(_rel_boundary(tr, m1, m2, D), []) for tr in robot_transforms], dtype=torch.float32).to(device)
                keep.append(table)
                cc[c].by_robot_dev = table.data_ptr()
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
        _lib.launch("mmd_p_sample_loop_ensemble", ws, tiles, K, cc, len(cross_conds), n_robots, B // n_robots, n_diffusion_steps,
                                                  n_diffusion_steps_without_noise, init_noise, ws.data_ptr(), ws.numel())
        if return_chain:
            return x, {m: chains[m].transpose(0, 1) for m in keys}            # [B, steps+1, H, D] like torch.stack(dim=1)
        return x

    def ddim_sample(self, shape, hard_conds, n_diffusion_steps=None, context=None, return_chain=False, t_start_guide=float("inf"),
                    guide=None, n_guide_steps=1, **sample_kwargs):
        """diffusion_ensemble.py:109-221 is a copy of the single model's DDIM loop that reads `self.betas` / `self.model` /
        `self.alphas_cumprod`, none of which a DiffusionsEnsemble has: in the reference the call raises AttributeError at its first
        statement (run on the genuine class: tests/test_host_logic.py records what was observed), and
        joint_conditional_sampling(ddim=True) does not even get that far (below).  There is no behaviour to reproduce but the error;
        single-model DDIM is GaussianDiffusionModel.ddim_sample (mmd_ddim_sample, golden g11 / g18)."""
        raise AttributeError("'DiffusionsEnsemble' object has no attribute 'betas'")

    def joint_conditional_sampling(self, hard_conds, cross_conds, n_diffusion_steps=None, batch_size=1, ddim=False,
                                   warm_start_path_b=None, **sample_kwargs):
        """diffusion_ensemble.py:223-244.  ddim=True fails in the reference with a TypeError: it passes cross_conds as the third
        positional argument of ddim_sample, which is n_diffusion_steps, next to the keyword of the same name."""
        shape = (batch_size, HORIZON, self.models[0].state_dim)
        if n_diffusion_steps is None:
            raise ValueError("n_diffusion_steps must be provided.")
        if ddim:
            if warm_start_path_b is not None:
                raise ValueError("warm_start_path_b is not supported for ddim sampling.")
            raise TypeError("DiffusionsEnsemble.ddim_sample() got multiple values for argument 'n_diffusion_steps'")
        return self.p_sample_loop(shape, hard_conds, cross_conds, n_diffusion_steps=n_diffusion_steps,
                                  warm_start_path_b=warm_start_path_b, **sample_kwargs)

    @torch.no_grad()
    def run_inference(self, contexts=None, hard_conds=None, cross_conds=None, n_samples=1, return_chain=False,
                      **diffusion_kwargs):
        """diffusion_ensemble.py:223-263: dict model -> [T+2, B, H, D] (return_chain) or [B, H, D]."""
        hard_conds = deepcopy(hard_conds)
        x, chains = self.joint_conditional_sampling(hard_conds, deepcopy(cross_conds), n_diffusion_steps=self.n_diffusion_steps,
                                                    batch_size=n_samples, return_chain=True, **diffusion_kwargs)
        chains = {m: c.transpose(0, 1) for m, c in chains.items()}
        return chains if return_chain else {m: c[-1] for m, c in chains.items()}

    @torch.no_grad()
    def run_local_inference(self, seed_trajectory_b, n_noising_steps, n_denoising_steps, contexts=None, hard_conds=None,
                            cross_conds=None, n_samples=1, return_chain=False, q_noise=None, **diffusion_kwargs):
        """diffusion_ensemble.py:265-313.  `q_noise` [B, K*64, D]: the injected draw of the forward noising (parity tests)."""
        hard_conds = deepcopy(hard_conds)
        noised = None if n_noising_steps is None else self.models[0].q_sample(seed_trajectory_b, n_noising_steps, noise=q_noise,
                                                                              seed=diffusion_kwargs.get("seed"))
        x, chains = self.joint_conditional_sampling(hard_conds, deepcopy(cross_conds), n_diffusion_steps=n_denoising_steps,
                                                    batch_size=n_samples, return_chain=True, warm_start_path_b=noised,
                                                    **diffusion_kwargs)
        chains = {m: c.transpose(0, 1) for m, c in chains.items()}
        return chains if return_chain else {m: c[-1] for m, c in chains.items()}
