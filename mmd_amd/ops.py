"""PyTorch-ROCm custom ops over the C ABI (include/mmd_amd.h): `torch.ops.mmd_amd.{unet_forward, guide_steps,
p_sample_loop, ddim_sample}`.  They take tensors instead of raw pointers, run on torch's CURRENT HIP stream
without any host synchronisation (so they can be captured into a hipGraph with torch.cuda.graph) and register fake
(meta) implementations so that torch.compile / FakeTensor tracing sees their output shapes.  The C header stays the ABI
of record: every op is a thin wrapper over the same entry point the host mirror classes call through ctypes.

Models and guides are opaque device-side objects (an mmd_unet_t handle + schedule tables; the guide's descriptor with its
resident SDF texture and constraint tables).  Ops receive them as integer tokens from `register(obj)`; the registry
holds weak references, so a token dies with its object.
"""
import ctypes as C
import itertools
import weakref

import torch

from . import _lib

_REGISTRY = weakref.WeakValueDictionary()
_NEXT_TOKEN = itertools.count(1)


def register(obj) -> int:
    """Token for a GaussianDiffusionModel / TemporalUnet / GuideManagerTrajectoriesWithVelocity to pass to the ops.
    Tokens come from a process-wide counter and are stored on the object (registering twice returns the same token), so
    a token is never reused: after its object is gone the ops raise "expired token" instead of resolving to whatever
    CPython later allocated at the same address."""
    tok = getattr(obj, "_mmd_amd_op_token", None)
    if tok is None:
        tok = next(_NEXT_TOKEN)
        obj._mmd_amd_op_token = tok
    _REGISTRY[tok] = obj
    return tok


def _get(token, what):
    obj = _REGISTRY.get(int(token))
    if obj is None:
        raise RuntimeError(f"mmd_amd op: unknown or expired {what} token {token}")
    return obj


def _check_traj(x, name="x"):
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.ndim == 3 and x.shape[1] == 64
            and x.shape[2] == 4):
        raise RuntimeError(f"mmd_amd op: {name} must be a contiguous float32 CUDA(HIP) tensor [n_traj, 64, 4]")


# ---- unet_forward --------------------------------------------------------------------------------------------------
@torch.library.custom_op("mmd_amd::unet_forward", mutates_args=(), device_types="cuda")
def unet_forward(x: torch.Tensor, t: int, n_timesteps: int, unet: int) -> torch.Tensor:
    """eps = TemporalUnet(x, t) (temporal_unet.py:121); `unet` = register(TemporalUnet), table sized n_timesteps."""
    _check_traj(x)
    model = _get(unet, "unet")
    out = torch.empty_like(x)
    ws = model.workspace(x.shape[0], x.device)
    _lib.launch("mmd_unet_forward", x, model.handle(n_timesteps, x.device), x.data_ptr(), int(t), out.data_ptr(), x.shape[0],
                                            ws.data_ptr(), ws.numel())
    return out


@unet_forward.register_fake
def _(x, t, n_timesteps, unet):
    return torch.empty_like(x)


# ---- guide_steps ---------------------------------------------------------------------------------------------------
@torch.library.custom_op("mmd_amd::guide_steps", mutates_args=("x",), device_types="cuda")
def guide_steps(x: torch.Tensor, hard: torch.Tensor, hard_rows: int, n_steps: int, guide: int) -> None:
    """In place: n_steps x { x += guide(x); apply_hard_conditioning } (sample_functions.py:89-107); hard [n_robots, n_rows, 4] + the 64-bit row
    mask as a SIGNED int64 (_lib.signed64(_lib.HARD_ROWS_START_GOAL) for the start / goal pair)."""
    _check_traj(x)
    g = _get(guide, "guide")
    d = g.desc()
    _lib.launch("mmd_guide_steps", x, C.byref(d), x.data_ptr(), _lib.require_gpu(hard, "hard"), int(hard_rows) & 0xFFFFFFFFFFFFFFFF,
                                           g.n_robots, x.shape[0] // g.n_robots, int(n_steps), None)


# ---- p_sample_loop -------------------------------------------------------------------------------------------------
@torch.library.custom_op("mmd_amd::p_sample_loop", mutates_args=("x",), device_types="cuda")
def p_sample_loop(x: torch.Tensor, hard: torch.Tensor, hard_rows: int, model: int, guide: int, n_robots: int,
                  n_steps: int, n_steps_without_noise: int, init_noise: bool, step_noise: torch.Tensor | None, seed: int,
                  n_guide_steps: int, t_start_guide: int, noise_std_extra: float, traj_index_base: int,
                  return_chain: bool) -> torch.Tensor:
    """GaussianDiffusionModel.p_sample_loop (diffusion_model_base.py:162-211) on x [n_traj,64,4] in place (x_T or the warm
    start on entry unless init_noise; the final sample on exit).  model = register(GaussianDiffusionModel); guide =
    register(guide) or 0.  Returns the chain [n_steps + n_steps_without_noise + 1, n_traj, 64, 4] (empty if not
    return_chain)."""
    _check_traj(x)
    m = _get(model, "model")
    g = _get(guide, "guide") if guide else None
    n_total = n_steps + n_steps_without_noise
    s = m._sampler_desc(n_guide_steps, t_start_guide, None, hard_rows, 0, traj_index_base)
    s.noise_std_extra, s.noise_std_extra_by_t = float(noise_std_extra), None
    gd = g.desc() if g is not None else None
    chain = (torch.empty((n_total + 1,) + tuple(x.shape), dtype=torch.float32, device=x.device) if return_chain
             else torch.empty(0, dtype=torch.float32, device=x.device))
    if step_noise is not None and tuple(step_noise.shape) != (n_total,) + tuple(x.shape):
        raise RuntimeError("mmd_amd::p_sample_loop: step_noise must be [n_steps_total, n_traj, 64, 4]")
    ws = m.model.workspace(x.shape[0], x.device, sampler=True)
    _lib.launch("mmd_p_sample_loop", x, m.model.handle(m.n_diffusion_steps, x.device), C.byref(s), C.byref(gd) if gd is not None else None, x.data_ptr(),
        _lib.require_gpu(hard, "hard"), int(n_robots), x.shape[0] // int(n_robots), int(n_steps), int(n_steps_without_noise),
        int(bool(init_noise)), _lib.require_gpu(step_noise, "step_noise") if step_noise is not None else None,
        C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), chain.data_ptr() if return_chain else None, ws.data_ptr(), ws.numel())
    return chain


@p_sample_loop.register_fake
def _(x, hard, hard_rows, model, guide, n_robots, n_steps, n_steps_without_noise, init_noise, step_noise, seed,
      n_guide_steps, t_start_guide, noise_std_extra, traj_index_base, return_chain):
    n_total = n_steps + n_steps_without_noise
    return x.new_empty((n_total + 1,) + tuple(x.shape)) if return_chain else x.new_empty(0)


# ---- ddim_sample ---------------------------------------------------------------------------------------------------
@torch.library.custom_op("mmd_amd::ddim_sample", mutates_args=("x",), device_types="cuda")
def ddim_sample(x: torch.Tensor, hard: torch.Tensor, hard_rows: int, model: int, guide: int, n_robots: int,
                n_diffusion_steps: int, init_noise: bool, seed: int, t_start_guide: int, traj_index_base: int,
                return_chain: bool) -> torch.Tensor:
    """GaussianDiffusionModel.ddim_sample (diffusion_model_base.py:213-290, eta = 0) on x in place; chain [n_times, ...]."""
    import numpy as np
    _check_traj(x)
    m = _get(model, "model")
    g = _get(guide, "guide") if guide else None
    s = m._sampler_desc(1, t_start_guide, None, hard_rows, 0, traj_index_base)
    times = np.asarray(m.ddim_times(n_diffusion_steps), dtype=np.int32)
    acp = np.ascontiguousarray(m._tables["alphas_cumprod"], dtype=np.float32)
    gd = g.desc() if g is not None else None
    chain = (torch.empty((len(times),) + tuple(x.shape), dtype=torch.float32, device=x.device) if return_chain
             else torch.empty(0, dtype=torch.float32, device=x.device))
    ws = m.model.workspace(x.shape[0], x.device, sampler=True)
    _lib.launch("mmd_ddim_sample", x, m.model.handle(m.n_diffusion_steps, x.device), C.byref(s), acp.ctypes.data, times.ctypes.data, len(times),
        C.byref(gd) if gd is not None else None, x.data_ptr(), _lib.require_gpu(hard, "hard"), int(n_robots),
        x.shape[0] // int(n_robots), int(bool(init_noise)), C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF),
        chain.data_ptr() if return_chain else None, ws.data_ptr(), ws.numel())
    return chain


@ddim_sample.register_fake
def _(x, hard, hard_rows, model, guide, n_robots, n_diffusion_steps, init_noise, seed, t_start_guide, traj_index_base,
      return_chain):
    n_times = n_diffusion_steps // 5 + 2
    return x.new_empty((n_times,) + tuple(x.shape)) if return_chain else x.new_empty(0)
