"""ctypes binding of libmmd_amd.so (include/mmd_amd.h).  There is NO fallback: if the HIP library is missing the
import of any compute entry point raises, and on a GPU box every op fails loudly rather than running on the CPU."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmmd_amd.so")
ABI_VERSION = 7
HARD_ROWS_START_GOAL = (1 << 63) | 1     # hard_rows of MPD's {0: start, H-1: goal} (include/mmd_amd.h: bit t = support point t pinned)


def signed64(mask):
    """a 64-bit row mask as the signed int64 a torch custom-op `int` argument carries (the C side sees the same bits)."""
    mask = int(mask) & 0xFFFFFFFFFFFFFFFF
    return mask - (1 << 64) if mask >> 63 else mask



class GuideDesc(C.Structure):
    _fields_ = [
        ("norm_min", C.c_float * 4), ("norm_max", C.c_float * 4),
        ("limits_lo", C.c_float * 2), ("limits_hi", C.c_float * 2),
        ("grid_nx", C.c_int32), ("grid_ny", C.c_int32), ("n_grids", C.c_int32), ("n_maps", C.c_int32),
        ("sdf_grids_dev", C.c_void_p), ("robot_map_dev", C.c_void_p),
        ("ws_min", C.c_float * 2), ("ws_max", C.c_float * 2),
        ("margin", C.c_float), ("dt", C.c_float), ("sigma_gp", C.c_float),
        ("weight_collision", C.c_float), ("weight_smoothness", C.c_float), ("max_grad_norm", C.c_float),
        ("cons_ell_dev", C.c_void_p), ("grp_slot_off_dev", C.c_void_p), ("grp_weight_dev", C.c_void_p),
        ("robot_grp_off_dev", C.c_void_p), ("max_slots_per_robot", C.c_int32), ("cons_uniform_radius", C.c_float),
        ("extra_spheres_dev", C.c_void_p), ("extra_boxes_dev", C.c_void_p),
        ("n_extra_spheres", C.c_int32), ("n_extra_boxes", C.c_int32),
        ("clip_grad_rule", C.c_int32), ("max_grad_value", C.c_float),
    ]


class SamplerDesc(C.Structure):
    _fields_ = [
        ("n_diffusion_steps", C.c_int32),
        ("sqrt_recip_alphas_cumprod", C.POINTER(C.c_float)),
        ("sqrt_recipm1_alphas_cumprod", C.POINTER(C.c_float)),
        ("posterior_mean_coef1", C.POINTER(C.c_float)),
        ("posterior_mean_coef2", C.POINTER(C.c_float)),
        ("posterior_log_variance_clipped", C.POINTER(C.c_float)),
        ("n_guide_steps", C.c_int32), ("t_start_guide", C.c_int32),
        ("noise_std_extra", C.c_float), ("hard_rows", C.c_uint64), ("n_streams", C.c_int32),
        ("traj_index_base", C.c_int64), ("noise_std_extra_by_t", C.POINTER(C.c_float)), ("profiler", C.c_void_p),
        ("scale_grad_by_std", C.c_int32), ("model_predicts_x0", C.c_int32),
        ("flags", C.c_uint32), ("guide_coop_max", C.c_int32), ("robot_seeds_dev", C.c_void_p),
    ]


SAMPLER_NO_FUSED_STEP, SAMPLER_PERSIST = 1, 2          # mmd_sampler_desc.flags
UNET_LAYERED, UNET_LAYERED_VALU = 1, 2                 # mmd_unet_options.flags


class UnetOptions(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("rtb_fused", C.c_int32), ("mconv_max_cs", C.c_int32), ("two_per_workgroup_max", C.c_int32)]


class EnsembleTile(C.Structure):
    _fields_ = [
        ("unet", C.c_void_p), ("sampler", C.POINTER(SamplerDesc)), ("guide", C.POINTER(GuideDesc)),
        ("x_dev", C.c_void_p), ("hard_dev", C.c_void_p), ("step_noise_dev", C.c_void_p), ("chain_dev", C.c_void_p),
        ("seed", C.c_uint64),
    ]


class CrossCond(C.Structure):
    _fields_ = [("m1", C.c_int32), ("m2", C.c_int32), ("ind1", C.c_int32), ("ind2", C.c_int32),
                ("rel", C.c_float * 4), ("boundary", C.c_float * 4), ("by_robot_dev", C.c_void_p)]


_SIGNATURES = {
    "mmd_abi_version": (C.c_int, []),
    "mmd_last_error": (C.c_char_p, []),
    "mmd_unet_num_tensors": (C.c_int, [C.c_int, C.c_int]),
    "mmd_unet_tensor_numel": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "mmd_unet_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                  C.POINTER(C.c_int64), C.c_int, C.POINTER(UnetOptions), C.c_void_p]),
    "mmd_unet_destroy": (C.c_int, [C.c_void_p]),
    "mmd_unet_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "mmd_unet_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                                   C.c_void_p]),
    "mmd_pack_constraints": (C.c_int, [C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int32)]),
    "mmd_soft_constraints_from_paths": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmd_guide_steps": (C.c_int, [C.POINTER(GuideDesc), C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_void_p]),
    "mmd_sampler_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "mmd_ddpm_step": (C.c_int, [C.c_void_p, C.POINTER(SamplerDesc), C.POINTER(GuideDesc), C.c_void_p, C.c_void_p,
                                C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_size_t,
                                C.c_void_p]),
    "mmd_p_sample_loop": (C.c_int, [C.c_void_p, C.POINTER(SamplerDesc), C.POINTER(GuideDesc), C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p,
                                    C.c_void_p, C.c_size_t, C.c_void_p]),
    "mmd_ddim_sample": (C.c_int, [C.c_void_p, C.POINTER(SamplerDesc), C.c_void_p, C.c_void_p, C.c_int, C.POINTER(GuideDesc),
                                  C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p,
                                  C.c_size_t, C.c_void_p]),
    "mmd_p_sample_loop_ensemble": (C.c_int, [C.POINTER(EnsembleTile), C.c_int, C.POINTER(CrossCond), C.c_int, C.c_int,
                                             C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mmd_q_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_uint64, C.c_uint32,
                               C.c_int64, C.c_int, C.c_void_p]),
    "mmd_rr_collisions": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmd_count_collisions": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                       C.c_void_p, C.c_void_p]),
    "mmd_postprocess_trajs": (C.c_int, [C.POINTER(GuideDesc), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int,
                                        C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "mmd_select_best": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "mmd_points_collision": (C.c_int, [C.POINTER(GuideDesc), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                       C.c_void_p]),
    "mmd_variance_waypoints": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mmd_unnormalize_trajs": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                        C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmd_cross_condition": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float),
                                      C.POINTER(C.c_float), C.c_int, C.c_void_p]),
}

# include/mmd_amd_debug.h: measurement hooks (bench.py / tools), not part of the drop-in boundary
_DEBUG_SIGNATURES = {
    "mmd_profiler_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "mmd_profiler_create_windowed": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int]),
    "mmd_profiler_intervals": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int,
                                         C.POINTER(C.c_int)]),
    "mmd_profiler_destroy": (C.c_int, [C.c_void_p]),
    "mmd_profiler_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "mmd_sampler_stream_chunks": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "mmd_unet_flops_per_trajectory": (C.c_double, []),
    "mmd_unet_mfma_flops_per_trajectory": (C.c_double, []),
    "mmd_unet_f16x2_flops_per_trajectory": (C.c_double, []),
    "mmd_unet_forward_profiled": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                                            C.c_void_p, C.c_void_p]),
    "mmd_unet_weight_bytes": (C.c_size_t, [C.c_void_p]),
    "mmd_debug_ddpm_step_trace": (C.c_int, [C.c_void_p, C.POINTER(SamplerDesc), C.POINTER(GuideDesc), C.c_void_p, C.c_void_p,
                                            C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_size_t,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}
TRACE_WORDS = 12                        # MMD_TRACE_WORDS of include/mmd_amd_debug.h

EXPORTED_SYMBOLS = tuple(_SIGNATURES)
DEBUG_SYMBOLS = tuple(_DEBUG_SIGNATURES)
_lib = None


def load():
    """Load (once) and return the CDLL; raises ImportError with the build hint if the library is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: the HIP extension is required (run ./build.sh or "
                              f"`python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in list(_SIGNATURES.items()) + list(_DEBUG_SIGNATURES.items()):
            fn = getattr(lib, name)          # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        if lib.mmd_abi_version() != ABI_VERSION:
            raise ImportError(f"libmmd_amd.so ABI {lib.mmd_abi_version()} != expected {ABI_VERSION}")
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError("libmmd_amd: " + load().mmd_last_error().decode())


def raw_stream(device_index=None):
    """hipStream_t (as an int) of torch's current stream on `device_index` (default: the current device).  torch's raw accessor where it
    exists: torch.cuda.current_stream() builds a Stream object per call, 9 us each and ten per planner call."""
    import torch
    if device_index is None:
        device_index = torch.cuda.current_device()
    get = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    return int(get(device_index)) if get is not None else int(torch.cuda.current_stream(device_index).cuda_stream)


def current_stream_ptr():
    return C.c_void_p(raw_stream())


def launch(name, on, *args):
    """Call entry point `name`, whose LAST parameter is the stream, on the GPU that owns tensor `on`: that device is made
    current for the call (the library's own streams, events and kernel launches follow hip's current device) and ITS current
    stream is passed -- so a call on tensors of cuda:1 while cuda:0 is torch's current device lands on cuda:1, with the
    model handle / workspace the host layer resolved for cuda:1 (TemporalUnet.handle / .workspace key on the same device)."""
    import torch
    dev = on.device
    if dev.type != "cuda":
        raise ValueError(f"{name}: tensors must live on a CUDA(HIP) device, got {dev}")
    fn = getattr(_lib if _lib is not None else load(), name)
    cur = torch.cuda.current_device()
    if dev.index is None or dev.index == cur:       # (the common case: no device switch around the call)
        rc = fn(*args, C.c_void_p(raw_stream(cur)))
    else:
        with torch.cuda.device(dev):
            rc = fn(*args, C.c_void_p(raw_stream(dev.index)))
    if rc != 0:
        check(rc)


def require_gpu(t, name="tensor"):
    import torch
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError(f"{name} must be a contiguous float32 CUDA(HIP) tensor")
    return C.c_void_p(t.data_ptr())
