"""DiffusionsEnsemble: host mirror of reference mmd/models/diffusion_models/diffusion_ensemble.py:37-313 (composition
of tile models along the horizon).  Per outer step every tile runs one fused DDPM step (mmd_ddpm_step) and the tile
boundaries are stitched by the cross-conditioning kernel (mmd_cross_condition, sample_functions.py:17-31)."""
import ctypes as C
from copy import deepcopy
from typing import Dict, Tuple

import torch

from . import _lib
from .diffusion_model import GaussianDiffusionModel, ddpm_sample_fn

HORIZON = 64     # mmd/config/mmd_params.py:34


def apply_cross_conditioning(x: Dict[int, torch.Tensor], conditions, transforms):
    """sample_functions.py:17-31 on device tensors (in place)."""
    lib = _lib.load()
    for (m1, m2), (ind1, ind2) in conditions.items():
        rel = (torch.as_tensor(transforms[m2]) - torch.as_tensor(transforms[m1])).float().cpu()
        D = x[m1].shape[2]
        if D > rel.shape[0]:
            rel = torch.cat([rel, torch.zeros(D - rel.shape[0])])
        boundary = rel / torch.norm(rel, keepdim=True)
        boundary[boundary == 0] = 1e6
        relc = (C.c_float * 4)(*[float(v) for v in rel])
        bndc = (C.c_float * 4)(*[float(v) for v in boundary])
        _lib.check(lib.mmd_cross_condition(_lib.require_gpu(x[m1], "x[m1]"), _lib.require_gpu(x[m2], "x[m2]"),
                                           int(ind1) % x[m1].shape[1], int(ind2) % x[m2].shape[1], relc, bndc,
                                           x[m1].shape[0], _lib.current_stream_ptr()))
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
                      x_init=None, step_noise=None, device="cuda", **sample_kwargs):
        """diffusion_ensemble.py:55-106.  `x_init` {m: [B,H,D]} / `step_noise` [n_steps, n_models, B,H,D] inject the
        Gaussian draws (parity tests); otherwise Philox."""
        if sample_fn is not ddpm_sample_fn:
            raise NotImplementedError("only ddpm_sample_fn")
        keys = list(self.models.keys())
        x = {}
        for m in keys:
            if warm_start_path_b is not None:
                x[m] = warm_start_path_b[:, m * HORIZON:(m + 1) * HORIZON, :].clone().to(device).contiguous()
                x[m][:, :, :2] -= torch.as_tensor(self.transforms[m]).to(x[m].device)
            elif x_init is not None:
                x[m] = x_init[m].to(device=device, dtype=torch.float32).contiguous().clone()
            else:
                x[m] = self.models[m].p_sample_loop(shape, {}, 0, device=device)        # Philox N(0,1)
            for row, val in hard_conds.get(m, {}).items():
                x[m][:, row, :] = torch.as_tensor(val, device=x[m].device)
            hard_conds.setdefault(m, {})
        x = apply_cross_conditioning(x, cross_conds, self.transforms)
        chains = {m: [x[m].clone()] for m in keys} if return_chain else None
        kw = sample_kwargs["sample_kwargs"]
        k = 0
        for i in reversed(range(-n_diffusion_steps_without_noise, n_diffusion_steps)):
            for j, m in enumerate(keys):
                skw = kw[m]
                self.models[m].sample_step(
                    x[m], hard_conds[m], i, guide=skw.get("guide"), n_guide_steps=skw.get("n_guide_steps", 1),
                    t_start_guide=skw.get("t_start_guide", float("inf")),
                    noise_std_extra_schedule_fn=skw.get("noise_std_extra_schedule_fn"),
                    noise=step_noise[k, j] if step_noise is not None else None)
                x = apply_cross_conditioning(x, cross_conds, self.transforms)
            if return_chain:
                for m in keys:
                    chains[m].append(x[m].clone())
            k += 1
        if return_chain:
            return x, {m: torch.stack(v, dim=1) for m, v in chains.items()}
        return x

    @torch.no_grad()
    def run_inference(self, contexts=None, hard_conds=None, cross_conds=None, n_samples=1, return_chain=False,
                      **diffusion_kwargs):
        """diffusion_ensemble.py:223-263: dict model -> [T+2, B, H, D] (return_chain) or [B, H, D]."""
        hard_conds = deepcopy(hard_conds)
        x, chains = self.p_sample_loop((n_samples, HORIZON, self.models[0].state_dim), hard_conds, deepcopy(cross_conds),
                                       n_diffusion_steps=self.n_diffusion_steps, return_chain=True, **diffusion_kwargs)
        chains = {m: c.transpose(0, 1) for m, c in chains.items()}
        return chains if return_chain else {m: c[-1] for m, c in chains.items()}

    @torch.no_grad()
    def run_local_inference(self, seed_trajectory_b, n_noising_steps, n_denoising_steps, contexts=None, hard_conds=None,
                            cross_conds=None, n_samples=1, return_chain=False, **diffusion_kwargs):
        """diffusion_ensemble.py:265-313."""
        hard_conds = deepcopy(hard_conds)
        noised = None if n_noising_steps is None else self.models[0].q_sample(seed_trajectory_b, n_noising_steps)
        x, chains = self.p_sample_loop((n_samples, HORIZON, self.models[0].state_dim), hard_conds, deepcopy(cross_conds),
                                       n_diffusion_steps=n_denoising_steps, return_chain=True,
                                       warm_start_path_b=noised, **diffusion_kwargs)
        chains = {m: c.transpose(0, 1) for m, c in chains.items()}
        return chains if return_chain else {m: c[-1] for m, c in chains.items()}
