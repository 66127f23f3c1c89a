"""GaussianDiffusionModel: host mirror of reference mmd/models/diffusion_models/diffusion_model_base.py:48-433
(sampling side) + sample_functions.py.  The p_sample_loop is ONE C-ABI call (mmd_p_sample_loop) that enqueues the
UNet kernels and the fused posterior/guide/noise kernel of every step on the current HIP stream."""
import threading
import ctypes as C
from copy import copy

import numpy as np
import torch

from . import _lib
from .guides import GuideManagerTrajectoriesWithVelocity
from .schedules import SCHEDULE_KEYS, diffusion_buffers


def ddpm_sample_fn(*args, **kwargs):
    """Marker with the reference's name (sample_functions.py:40): the DDPM step is fused into the sampler kernels;
    passing any other sample_fn raises."""
    raise RuntimeError("ddpm_sample_fn is executed inside libmmd_amd.so; it is only a marker on the host side")


# Every sampling call that draws its own noise consumes one Philox stream id from this process-wide counter, the way
# every reference call advances torch's global RNG: planners built with the same `seed` (the reference seeds once,
# fix_random_seed, then draws from one global stream) never repeat each other's noise.
_GLOBAL_DRAWS = 0


_DRAW_LOCK = threading.Lock()
_HARD_CACHE, _HARD_LOCK = {}, threading.Lock()      # GaussianDiffusionModel._hard_tensor


def next_stream_seed(base_seed):
    global _GLOBAL_DRAWS
    with _DRAW_LOCK:                 # (planner calls may run on several host threads: planners.plan_concurrently)
        seed = (int(base_seed) << 24) + _GLOBAL_DRAWS
        _GLOBAL_DRAWS += 1
    return seed & 0xFFFFFFFFFFFFFFFF


def make_timesteps(batch_size, i, device):
    return torch.full((batch_size,), i, device=device, dtype=torch.long)


class GaussianDiffusionModel:
    def __init__(self, model=None, variance_schedule="exponential", n_diffusion_steps=100, clip_denoised=True,
                 predict_epsilon=False, loss_type="l2", context_model=None, **kwargs):
        if not clip_denoised or context_model is not None:
            raise NotImplementedError("kernels implement clip_denoised=True, no context (the configuration of the released MPD "
                                      "checkpoints; the reference itself asserts on clip_denoised=False, diffusion_model_base.py:156)")
        self.model = model
        self.n_diffusion_steps = n_diffusion_steps
        if hasattr(model, "max_timesteps"):
            model.max_timesteps = n_diffusion_steps              # the time-embedding table is sized to the schedule
        self.state_dim = model.state_dim
        self.clip_denoised, self.predict_epsilon = clip_denoised, predict_epsilon
        for k, v in diffusion_buffers(n_diffusion_steps, variance_schedule).items():
            setattr(self, k, v)                                    # CPU float32 [T] buffers, reference names
        self._tables = {k: np.ascontiguousarray(getattr(self, k).numpy()) for k in SCHEDULE_KEYS}
        self.seed = 0
        self.profiler = None                                       # optional mmd_profiler_t (bench.py), see mmd_amd_debug.h
        self.sampler_flags = 0                                     # mmd_sampler_desc.flags (_lib.SAMPLER_*: measurement switches)
        self.guide_coop_max = 0                                    # mmd_sampler_desc.guide_coop_max (0 = the library's default)
        self._noise_tables = {}

    # ---- parameters -------------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict=True):
        self.model.load_state_dict({k[len("model."):]: v for k, v in state_dict.items() if k.startswith("model.")})
        return self

    def state_dict(self):
        sd = {k: getattr(self, k) for k in SCHEDULE_KEYS}
        sd.update({"model." + k: v for k, v in self.model.state_dict().items()})
        return sd

    def eval(self):
        return self

    def warmup(self, horizon=64, device="cuda"):
        x = torch.randn((2, horizon, self.state_dim), device=device)
        self.model(x, 1, context=None)

    # ---- descriptors ------------------------------------------------------------------------------------------
    def _noise_std_table(self, fn):
        """noise_std_extra_schedule_fn evaluated for every t of the schedule (the reference calls it with t_single on
        every step, sample_functions.py:83-86); None -> 1.0."""
        key = id(fn)
        if key not in self._noise_tables:
            vals = [1.0 if fn is None else float(fn(t)) for t in range(self.n_diffusion_steps)]
            self._noise_tables[key] = (fn, np.ascontiguousarray(vals, dtype=np.float32))   # holds fn: its id stays unique
            while len(self._noise_tables) > 32:     # descriptors built earlier in the same call keep pointing at live tables
                self._noise_tables.pop(next(iter(self._noise_tables)))
        return self._noise_tables[key][1]

    def _sampler_desc(self, n_guide_steps, t_start_guide, noise_fn, hard_rows, n_streams=0, traj_index_base=0,
                      scale_grad_by_std=False, robot_seeds=None):
        s = _lib.SamplerDesc()
        s.n_diffusion_steps = self.n_diffusion_steps
        fp = C.POINTER(C.c_float)
        for name in ("sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1",
                     "posterior_mean_coef2", "posterior_log_variance_clipped"):
            setattr(s, name, self._tables[name].ctypes.data_as(fp))
        s.n_guide_steps = int(n_guide_steps)
        tsg = t_start_guide
        s.t_start_guide = int(min(tsg, 2 ** 30)) if tsg != float("inf") else 2 ** 30
        table = self._noise_std_table(noise_fn)
        s.noise_std_extra = float(table[0])
        s.noise_std_extra_by_t = table.ctypes.data_as(fp)
        s.hard_rows = int(hard_rows) & 0xFFFFFFFFFFFFFFFF      # (torch custom ops carry the mask as a signed int64)
        s.n_streams = int(n_streams)
        s.traj_index_base = int(traj_index_base)
        s.profiler = self.profiler
        s.scale_grad_by_std = int(bool(scale_grad_by_std))
        s.model_predicts_x0 = 0 if self.predict_epsilon else 1
        s.flags, s.guide_coop_max = int(self.sampler_flags), int(self.guide_coop_max)
        if robot_seeds is not None:                              # int64 device tensor [n_robots] (kept alive by the caller)
            s.robot_seeds_dev = robot_seeds.data_ptr()
        return s

    @staticmethod
    def robot_seed_tensor(robot_seeds, n_robots, device):
        """[n_robots] Philox seeds -> the int64 device tensor mmd_sampler_desc.robot_seeds_dev points at (the same 64 bits), or None."""
        if robot_seeds is None:
            return None
        seeds = [_lib.signed64(int(v)) for v in robot_seeds]
        if len(seeds) != n_robots:
            raise ValueError(f"robot_seeds: {len(seeds)} seeds for {n_robots} robots")
        return torch.tensor(seeds, dtype=torch.int64).to(device)

    @staticmethod
    def _hard_tensor(hard_conds, n_robots, horizon, device, D):
        """apply_hard_conditioning's dict (sample_functions.py:8-14) {row: [D] | [n_robots, D] | [B_total, D]} -> ([n_robots, n_rows, D]
        float32 in ascending row order, the 64-bit row mask of include/mmd_amd.h).  Per-sample hard conditions must be constant
        within a robot (they are: run_inference repeats one state, diffusion_model_base.py:327-329).
        A planner passes the SAME tensor objects call after call (its stored start / goal): the device copy is kept per (objects,
        their in-place version counters) -- building it is two host -> device copies and four small kernels, 60 us in front of
        a planner call's first launch."""
        rows = sorted(int(r) for r in hard_conds)
        if rows and not 0 <= rows[0] <= rows[-1] < horizon:
            raise ValueError(f"hard condition rows {rows} outside [0, {horizon})")
        vals = [hard_conds[r] for r in rows]
        key = None
        if all(torch.is_tensor(v) for v in vals):
            device = torch.device(device)
            index = device.index if device.index is not None or device.type != "cuda" else torch.cuda.current_device()
            key = (tuple((r, id(v), v._version) for r, v in zip(rows, vals)), n_robots, horizon, device.type, index, D)
            with _HARD_LOCK:
                hit = _HARD_CACHE.get(key)
            if hit is not None:
                if hit[1].is_cuda:          # (a later call may run on another stream: the allocator must not recycle it under that stream)
                    hit[1].record_stream(torch.cuda.current_stream(hit[1].device))
                return hit[1], hit[2]
        hard, mask = GaussianDiffusionModel._build_hard_tensor(hard_conds, rows, n_robots, device, D)
        if key is not None:
            with _HARD_LOCK:
                while len(_HARD_CACHE) >= 256:
                    _HARD_CACHE.pop(next(iter(_HARD_CACHE)))
                _HARD_CACHE[key] = (vals, hard, mask)            # (holds the value tensors: their ids stay unique while cached)
        return hard, mask

    @staticmethod
    def _build_hard_tensor(hard_conds, rows, n_robots, device, D):
        hard = torch.zeros(n_robots, max(len(rows), 1), D, dtype=torch.float32, device=device)
        mask = 0
        for slot, row in enumerate(rows):
            val = torch.as_tensor(hard_conds[row], dtype=torch.float32, device=device)
            if val.ndim == 1:
                val = val[None].expand(n_robots, D)
            elif val.shape[0] != n_robots:
                per = val.shape[0] // n_robots
                val = val[::per]
            hard[:, slot], mask = val, mask | (1 << row)
        return hard.contiguous(), mask

    # ---- sampling ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def p_sample_loop(self, shape, hard_conds, n_diffusion_steps, context=None, return_chain=False,
                      sample_fn=ddpm_sample_fn, n_diffusion_steps_without_noise=0, warm_start_path_b=None,
                      guide=None, n_guide_steps=1, t_start_guide=float("inf"), noise_std_extra_schedule_fn=None,
                      n_robots=1, step_noise=None, seed=None, device="cuda", n_streams=0, traj_index_base=0,
                      scale_grad_by_std=False, robot_seeds=None, **sample_kwargs):
        """diffusion_model_base.py:162-211.  Extensions: `n_robots` (batch = n_robots * n_samples, robot-major),
        `step_noise` [n_steps_total, B, H, D] + `warm_start_path_b` as x_T to inject every Gaussian draw (parity
        tests), `seed` for the in-kernel Philox stream otherwise, `traj_index_base` = global index of this call's first
        trajectory (a rank sampling robots [r0, r1) of a bigger instance passes r0 * n_samples and draws exactly the noise
        those rows get in the unsharded call); `robot_seeds` [n_robots]: one Philox stream per robot -- the batch then draws
        exactly what n_robots separate one-robot calls with those seeds draw (planners.plan_batched)."""
        if sample_fn is not ddpm_sample_fn:
            raise NotImplementedError("only ddpm_sample_fn is implemented (DDIM: conditional_sample(ddim=True) / ddim_sample)")
        if context is not None:
            raise NotImplementedError("context")
        if guide is not None and not isinstance(guide, GuideManagerTrajectoriesWithVelocity):
            raise NotImplementedError("guide must be a mmd_amd GuideManagerTrajectoriesWithVelocity")
        B_total, H, D = shape
        device = torch.device(device)
        lib = _lib.load()
        hard, mask = self._hard_tensor(hard_conds, n_robots, H, device, D)
        seeds_dev = self.robot_seed_tensor(robot_seeds, n_robots, device)
        s = self._sampler_desc(n_guide_steps, t_start_guide, noise_std_extra_schedule_fn, mask, n_streams, traj_index_base,
                               scale_grad_by_std=scale_grad_by_std, robot_seeds=seeds_dev)
        n_total = n_diffusion_steps + n_diffusion_steps_without_noise
        if warm_start_path_b is not None:
            x = warm_start_path_b.to(device=device, dtype=torch.float32).contiguous().clone()
            init_noise = 0
        else:
            x = torch.empty(shape, dtype=torch.float32, device=device)
            init_noise = 1
        chain = torch.empty((n_total + 1,) + tuple(shape), dtype=torch.float32, device=device) if return_chain else None
        if step_noise is not None:
            step_noise = step_noise.to(device=device, dtype=torch.float32).contiguous()
            assert step_noise.shape == (n_total,) + tuple(shape)
        gd = guide.desc() if guide is not None else None
        ws = self.model.workspace(B_total, device, sampler=True)
        if seed is None:
            seed = next_stream_seed(self.seed)
        _lib.launch("mmd_p_sample_loop", x, self.model.handle(self.n_diffusion_steps, device), C.byref(s), C.byref(gd) if gd is not None else None,
            x.data_ptr(), hard.data_ptr(), n_robots, B_total // n_robots, n_diffusion_steps,
            n_diffusion_steps_without_noise, init_noise, step_noise.data_ptr() if step_noise is not None else None,
            C.c_uint64(seed), chain.data_ptr() if chain is not None else None, ws.data_ptr(), ws.numel())
        if return_chain:
            return x, chain.transpose(0, 1)                      # [B, steps+1, H, D] like torch.stack(chain, dim=1)
        return x

    @staticmethod
    def ddim_times(n_diffusion_steps):
        """[T-1, ..., 0, -1]: diffusion_model_base.py:225-235 (sampling_timesteps = T // 5), computed as the reference does."""
        times = torch.linspace(0, n_diffusion_steps - 1, steps=n_diffusion_steps // 5 + 1)
        times = torch.cat((torch.tensor([-1.0]), times))
        return list(reversed(times.int().tolist()))

    @torch.no_grad()
    def ddim_sample(self, shape, hard_conds, n_diffusion_steps, context=None, return_chain=False,
                    t_start_guide=float("inf"), guide=None, n_guide_steps=1, n_robots=1, x_init=None, seed=None,
                    device="cuda", traj_index_base=0, **sample_kwargs):
        """diffusion_model_base.py:213-290 (eta = 0).  As in the reference, `n_guide_steps` is accepted and NOT forwarded
        to guide_gradient_steps: one guide step per sampling step.  Extensions: `n_robots`, `x_init` (injected x_T),
        `seed` for the Philox draw of x_T otherwise."""
        if context is not None:
            raise NotImplementedError("context")
        if guide is not None and not isinstance(guide, GuideManagerTrajectoriesWithVelocity):
            raise NotImplementedError("guide must be a mmd_amd GuideManagerTrajectoriesWithVelocity")
        B_total, H, D = shape
        device = torch.device(device)
        lib = _lib.load()
        hard, mask = self._hard_tensor(hard_conds, n_robots, H, device, D)
        s = self._sampler_desc(1, t_start_guide, None, mask, traj_index_base=traj_index_base)
        times = np.asarray(self.ddim_times(n_diffusion_steps), dtype=np.int32)
        acp = np.ascontiguousarray(self._tables["alphas_cumprod"], dtype=np.float32)
        if x_init is not None:
            x = x_init.to(device=device, dtype=torch.float32).contiguous().clone()
            init_noise = 0
        else:
            x = torch.empty(shape, dtype=torch.float32, device=device)
            init_noise = 1
        chain = torch.empty((len(times),) + tuple(shape), dtype=torch.float32, device=device) if return_chain else None
        gd = guide.desc() if guide is not None else None
        ws = self.model.workspace(B_total, device, sampler=True)
        if seed is None:
            seed = next_stream_seed(self.seed)
        _lib.launch("mmd_ddim_sample", x, self.model.handle(self.n_diffusion_steps, device), C.byref(s), acp.ctypes.data, times.ctypes.data, len(times),
            C.byref(gd) if gd is not None else None, x.data_ptr(), hard.data_ptr(), n_robots, B_total // n_robots,
            init_noise, C.c_uint64(seed), chain.data_ptr() if chain is not None else None, ws.data_ptr(), ws.numel())
        if return_chain:
            return x, chain.transpose(0, 1)
        return x

    @torch.no_grad()
    def sample_step(self, x, hard_conds, i, guide=None, n_guide_steps=1, t_start_guide=float("inf"),
                    noise_std_extra_schedule_fn=None, n_robots=1, noise=None, seed=None, traj_index_base=0,
                    scale_grad_by_std=False):
        """One `ddpm_sample_fn` call + the `apply_hard_conditioning` that follows it in the loop
        (sample_functions.py:40-86, diffusion_model_base.py:199-203), in place on x; `i` is the loop index
        (negative: t = 0).  This is what DiffusionsEnsemble interleaves across tiles (diffusion_ensemble.py:86-100)."""
        B_total, H, D = x.shape
        hard, mask = self._hard_tensor(hard_conds, n_robots, H, x.device, D)
        s = self._sampler_desc(n_guide_steps, t_start_guide, noise_std_extra_schedule_fn, mask,
                               traj_index_base=traj_index_base, scale_grad_by_std=scale_grad_by_std)
        gd = guide.desc() if guide is not None else None
        ws = self.model.workspace(B_total, x.device, sampler=True)
        if seed is None:
            seed = next_stream_seed(self.seed)
        _lib.launch("mmd_ddpm_step", x, self.model.handle(self.n_diffusion_steps, x.device), C.byref(s), C.byref(gd) if gd is not None else None,
            _lib.require_gpu(x, "x"), hard.data_ptr(), n_robots, B_total // n_robots, int(i),
            _lib.require_gpu(noise.contiguous(), "noise") if noise is not None else None, C.c_uint64(seed),
            C.c_uint32(int(i) & 0xFFFFFFFF), ws.data_ptr(), ws.numel())
        return x

    @torch.no_grad()
    def conditional_sample(self, hard_conds, n_diffusion_steps, horizon=None, batch_size=1, ddim=False,
                           warm_start_path_b=None, **sample_kwargs):
        shape = (batch_size, horizon or 64, self.state_dim)
        if ddim:
            if warm_start_path_b is not None:
                raise ValueError("warm_start_path_b is not supported for ddim sampling.")     # diffusion_model_base.py:303
            return self.ddim_sample(shape, hard_conds, n_diffusion_steps=n_diffusion_steps, **sample_kwargs)
        return self.p_sample_loop(shape, hard_conds, n_diffusion_steps=n_diffusion_steps,
                                  warm_start_path_b=warm_start_path_b, **sample_kwargs)

    @torch.no_grad()
    def run_inference(self, context=None, hard_conds=None, n_samples=1, return_chain=False, n_robots=1,
                      **diffusion_kwargs):
        """diffusion_model_base.py:320-351.  Returns [T+2, B, H, D] (return_chain) or [B, H, D]."""
        samples, chain = self.conditional_sample(copy(hard_conds), n_diffusion_steps=self.n_diffusion_steps,
                                                 context=context, batch_size=n_samples * n_robots, return_chain=True,
                                                 n_robots=n_robots, **diffusion_kwargs)
        chain = chain.transpose(0, 1)                            # 'b diffsteps h d -> diffsteps b h d'
        return chain if return_chain else chain[-1]

    @torch.no_grad()
    def run_local_inference(self, seed_trajectory_b, n_noising_steps, n_denoising_steps, context=None,
                            hard_conds=None, n_samples=1, return_chain=False, n_robots=1, q_noise=None,
                            **diffusion_kwargs):
        """diffusion_model_base.py:353-421: forward-noise the seed batch n_noising_steps, then denoise."""
        if n_noising_steps is None:
            noised = None
        else:
            noised = self.q_sample(seed_trajectory_b, n_noising_steps, noise=q_noise, seed=diffusion_kwargs.get("seed"))
        samples, chain = self.conditional_sample(copy(hard_conds), n_diffusion_steps=n_denoising_steps, context=context,
                                                 batch_size=n_samples * n_robots, return_chain=True,
                                                 warm_start_path_b=noised, n_robots=n_robots, **diffusion_kwargs)
        chain = chain.transpose(0, 1)
        return chain if return_chain else chain[-1]

    def q_sample(self, x_start, t, noise=None, traj_index_base=0, seed=None):
        """diffusion_model_base.py:425-433 (t: int or a constant [B] tensor).  x_start is [B, K*64, 4]: K = 1 for a
        single model, K tiles chained along the horizon for the ensemble's seed (diffusion_ensemble.py:279-281) -- every
        one of the B*K*64 points is noised.  `seed`: the call's Philox seed (the draw has its own stream index, so a sampling
        loop under the same seed does not repeat it); default: the next one of the global stream."""
        t = int(t[0].item()) if torch.is_tensor(t) else int(t)
        x_start = x_start.to(dtype=torch.float32).contiguous()
        if x_start.ndim != 3 or x_start.shape[1] % 64 or x_start.shape[2] != self.state_dim:
            raise ValueError(f"q_sample: expected [B, K*64, {self.state_dim}], got {tuple(x_start.shape)}")
        out = torch.empty_like(x_start)
        _lib.launch("mmd_q_sample", x_start, out.data_ptr(), _lib.require_gpu(x_start, "x_start"),
            _lib.require_gpu(noise.contiguous(), "noise") if noise is not None else None,
            float(self.sqrt_alphas_cumprod[t]), float(self.sqrt_one_minus_alphas_cumprod[t]),
            C.c_uint64(next_stream_seed(self.seed) if seed is None else int(seed) & 0xFFFFFFFFFFFFFFFF), 0xFFFFFFFE, C.c_int64(int(traj_index_base)),
            x_start.numel() // (64 * self.state_dim))
        return out
