"""When do the workgroups of unet_kernel end?  -DMMD_TRACE side build (MMD_AMD_LIB=<that .so>), RUNS traced launches of n
trajectories: kernel span, mean end of the first / second dispatch half, and the launch-to-launch time of untraced launches.
Usage: trace_ends.py [n_traj]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from mmd_amd import _lib, synth
_lib.LIB_PATH = os.environ["MMD_AMD_LIB"]
from mmd_amd.temporal_unet import TemporalUnet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
lib = _lib.load()
unet = TemporalUnet(two_per_workgroup_max=int(os.environ.get('MMD_AMD_UNET_NS2_MAX', '0')))   # (this TOOL's knob, passed on as mmd_unet_options: the library reads no environment)
unet.load_state_dict(synth.synth_unet_state_dict(0))
x = torch.randn(n, 64, 4, device="cuda")
ns2_max = int(os.environ.get("MMD_AMD_UNET_NS2_MAX", "512"))      # unet_kernel<2> (two trajectories per workgroup) up to here
nb = (n + 1) // 2 if n <= ns2_max else (n + 3) // 4
trace = torch.zeros(nb * 4 * 256, dtype=torch.int64, device="cuda")
for _ in range(3):
    unet(x, 50)
torch.cuda.synchronize()
lib.mmd_debug_set_trace.argtypes = [C.c_void_p]
spans, h0, h1 = [], [], []
for run in range(int(os.environ.get("RUNS", "8"))):
    trace.zero_()
    torch.cuda.synchronize()
    assert lib.mmd_debug_set_trace(trace.data_ptr()) == 0
    unet(x, 50)
    torch.cuda.synchronize()
    lib.mmd_debug_set_trace(None)
    t = trace.cpu().numpy().reshape(nb, 4, 256).astype(np.float64) * 10.0   # ns
    tags = [i for i in range(256) if (t[:, :, i] > 0).all()]
    t0 = t[:, :, tags[0]].min()
    end = (t[:, :, tags[-1]].max(axis=1) - t0) / 1e3
    spans.append(end.max())
    h0.append(end[:nb // 2].mean())
    h1.append(end[nb // 2:].mean())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    unet(x, 50)
e1.record()
torch.cuda.synchronize()
f = lambda v: " ".join(f"{a:.1f}" for a in v)
print(f"n={n}: span us {f(spans)} | mean {np.mean(spans):.1f}; first half ends {np.mean(h0):.1f}, second half {np.mean(h1):.1f}; "
      f"untraced launch-to-launch {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
