"""TemporalUnet: host mirror of reference mmd/models/diffusion_models/temporal_unet.py:23-174 whose forward runs as
hand-written gfx950 kernels (mmd_amd/csrc/unet.hip) behind the C ABI (include/mmd_amd.h: mmd_unet_*)."""
import ctypes as C
import hashlib
import os
import threading
import weakref
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from .unet_spec import UNET_DIM_MULTS, unet_param_spec   # noqa: F401

# Device-side sharing: the reference builds N+1 planner objects, each reloading the same checkpoint
# (scripts/inference/inference_multi_agent.py:186-237).  Here every TemporalUnet with the same parameters (content hash),
# the same number of diffusion steps and the same device shares ONE device model: one packed weight blob + one
# time-embedding table.  The cache holds weak references: the device model is freed when its last user goes away.
_DEVICE_MODELS = weakref.WeakValueDictionary()
_CREATE_LOCK = threading.Lock()     # device models are created one at a time (concurrent planner calls: planners.plan_concurrently)
MAX_WORKSPACES = 8                   # scratch buffers a TemporalUnet keeps: one per (device, stream), least recently used dropped
N_DEVICE_MODELS_CREATED = 0          # number of mmd_unet_create calls made by this process (tests / constructor reports)


class _DeviceModel:
    """Owner of one mmd_unet_t."""

    def __init__(self, handle):
        self.handle = handle

    def __del__(self):
        try:
            if self.handle is not None:
                _lib.load().mmd_unet_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class TemporalUnet:
    def __init__(self, n_support_points=64, state_dim=4, unet_input_dim=32, dim_mults=(1, 2, 4), time_emb_dim=32,
                 self_attention=False, conditioning_type=None, max_timesteps=1000, layered=False, layered_valu=False,
                 rtb_fused=-1, mconv_max_cs=0, two_per_workgroup_max=0, **kwargs):
        if self_attention or conditioning_type not in (None, "None"):
            raise NotImplementedError("only the configuration MPD/MPDEnsemble instantiate is supported "
                                      "(no self-attention, no context conditioning)")
        if n_support_points != 64 or state_dim != 4 or time_emb_dim != 32:
            raise NotImplementedError("kernels are instantiated for H=64, state_dim=4, time_emb_dim=32")
        self.state_dim, self.n_support_points = state_dim, n_support_points
        self.unet_input_dim, self.dim_mults = unet_input_dim, tuple(dim_mults)
        # UNET_DIM_MULTS (mmd/models/__init__.py:8-11) = {0: (1, 2, 4), 1: (1, 2, 4, 8)}: option 0 with unet_input_dim 32 runs the
        # fused kernel, every other doubling ladder the layer-by-layer kernels (csrc/unet_layers.hip)
        if self.dim_mults != (1, 2, 4, 8)[:len(self.dim_mults)] or not self.dim_mults:
            raise NotImplementedError(f"dim_mults {self.dim_mults}: only a prefix of (1, 2, 4, 8) (each level doubles the channels)")
        if unet_input_dim % 8 or not 8 <= unet_input_dim <= 64:
            raise NotImplementedError("unet_input_dim must be a multiple of 8 in [8, 64]")
        self.spec = unet_param_spec(state_dim, unet_input_dim, self.dim_mults)
        self.max_timesteps = max_timesteps
        # layered=True: the layer-by-layer kernels also for the fused kernel's own configuration (the A/B of the two implementations,
        # tests/test_gpu_dim_mults.py).  These choices are fixed for the life of the object and travel to mmd_unet_create as its
        # mmd_unet_options argument (include/mmd_amd.h) -- nothing is read from, or written to, the process environment.
        self.layered = bool(layered)
        self.options = (int(bool(layered)) * _lib.UNET_LAYERED | int(bool(layered_valu)) * _lib.UNET_LAYERED_VALU, int(rtb_fused),
                        int(mconv_max_cs), int(two_per_workgroup_max))
        self._sd = None
        self._sd_hash = None
        self._models = {}               # (n_timesteps, device index) -> _DeviceModel (shared through _DEVICE_MODELS)
        self._ws = {}                   # (device index, stream) -> scratch buffer

    # ---- parameters -------------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict=True):
        """Accepts the reference's keys with or without the `model.` prefix; torch tensors or numpy arrays."""
        sd = OrderedDict()
        for k, shape in self.spec.items():
            v = state_dict.get(k, state_dict.get("model." + k))
            if v is None:
                raise KeyError(f"missing parameter {k}")
            v = np.ascontiguousarray(torch.as_tensor(v).detach().cpu().numpy(), dtype=np.float32)
            if tuple(v.shape) != tuple(shape):
                raise ValueError(f"{k}: shape {v.shape} != {shape}")
            sd[k] = v
        self._sd = sd
        h = hashlib.blake2b(digest_size=16)
        for v in sd.values():
            h.update(v.tobytes())
        self._sd_hash = h.hexdigest()
        self._models = {}
        return self

    def state_dict(self):
        return OrderedDict((k, torch.from_numpy(v.copy())) for k, v in self._sd.items())

    @staticmethod
    def _device_index(device):
        if device is None:
            return torch.cuda.current_device()
        d = torch.device(device)
        return torch.cuda.current_device() if d.index is None else d.index

    def handle(self, n_timesteps=None, device=None):
        """Device model (packed weights + time-embedding table for t in [0, n_timesteps)), shared by every TemporalUnet
        of this process that holds the same parameters on the same device.  `device`: where the caller's tensors live
        (default: the current device); one TemporalUnet may serve several devices, each gets its own handle."""
        if self._sd is None:
            raise RuntimeError("TemporalUnet has no parameters: call load_state_dict first")
        dev = self._device_index(device)
        layered = self.options
        if n_timesteps is not None:
            T = int(n_timesteps)
        else:
            have = [t for (t, d, l) in self._models if d == dev and l == layered]
            T = max(have) if have else self.max_timesteps
        if (T, dev, layered) not in self._models:
            with _CREATE_LOCK:
                self._create_locked(T, dev, layered)
        return self._models[(T, dev, layered)].handle

    def _create_locked(self, T, dev, layered):
        global N_DEVICE_MODELS_CREATED
        if (T, dev, layered) in self._models:
            return
        key = (self._sd_hash, self.unet_input_dim, self.dim_mults, T, dev, layered)
        dm = _DEVICE_MODELS.get(key)
        if dm is None:
            lib = _lib.load()
            n = len(self._sd)
            ptrs = (C.c_void_p * n)(*[v.ctypes.data for v in self._sd.values()])
            numels = (C.c_int64 * n)(*[v.size for v in self._sd.values()])
            h = C.c_void_p()
            opt = _lib.UnetOptions(*layered)
            with torch.cuda.device(dev):
                _lib.check(lib.mmd_unet_create(C.byref(h), self.unet_input_dim, len(self.dim_mults), T, ptrs, numels, n,
                                               C.byref(opt), _lib.current_stream_ptr()))
            N_DEVICE_MODELS_CREATED += 1
            dm = _DeviceModel(h)
            _DEVICE_MODELS[key] = dm
        self._models[(T, dev, layered)] = dm

    def workspace(self, n_traj, device, sampler=False):
        """Scratch for one call: one buffer per (device, stream), so calls issued on different streams never share the
        eps block (grown on demand, reused by later calls on the same stream)."""
        lib = _lib.load()
        dev = self._device_index(device)
        nbytes = (lib.mmd_sampler_workspace_bytes if sampler else lib.mmd_unet_workspace_bytes)(self.handle(device=dev), n_traj)
        key = (dev, _lib.raw_stream(dev))
        ws = self._ws.pop(key, None)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=torch.device("cuda", dev))
        self._ws[key] = ws                       # (re-inserted: most recently used last)
        while len(self._ws) > MAX_WORKSPACES:    # streams come and go: keep the buffers of the most recent ones only
            self._ws.pop(next(iter(self._ws)))
        return ws

    # ---- forward ----------------------------------------------------------------------------------------------
    def forward(self, x, time, context=None):
        """x [B,H,D] float32 on the GPU; time: int, or a [B] tensor (temporal_unet.py:121) -- identical entries as make_timesteps
        produces (diffusion_model_base.py:27-29) are one launch, distinct entries one launch per distinct timestep; returns eps
        [B,H,D]."""
        if context is not None:
            raise NotImplementedError("context conditioning is not used by MPD/MPDEnsemble")
        x = x.contiguous()
        if torch.is_tensor(time) and time.numel() > 1:
            if time.numel() != x.shape[0]:
                raise ValueError(f"TemporalUnet.forward: time has {time.numel()} entries for a batch of {x.shape[0]}")
            tt = time.reshape(-1).to("cpu", torch.int64)
            uniq = torch.unique(tt)
            if uniq.numel() > 1:
                # per-sample timesteps (temporal_unet.py:121, t [B]; the training loss draws them per sample): the kernel takes one
                # time embedding per launch, so the batch is run per distinct timestep and scattered back.  The sampling path
                # (make_timesteps) is uniform and never comes here.
                out = torch.empty_like(x)
                for t in uniq.tolist():
                    rows = torch.nonzero(tt == t).reshape(-1).to(x.device)
                    out.index_copy_(0, rows, self.forward(x.index_select(0, rows), int(t)))
                return out
            t = int(uniq[0])
        else:
            t = int(time.item()) if torch.is_tensor(time) else int(time)
        out = torch.empty_like(x)
        ws = self.workspace(x.shape[0], x.device)
        _lib.launch("mmd_unet_forward", x, self.handle(device=x.device), _lib.require_gpu(x, "x"), t, out.data_ptr(), x.shape[0],
                                                ws.data_ptr(), ws.numel())
        return out

    __call__ = forward
