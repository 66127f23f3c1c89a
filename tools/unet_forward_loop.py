"""N back-to-back TemporalUnet forwards of n trajectories (the body of the PMC / A-B passes).  MMD_AMD_LIB selects the
.so.  Usage: python tools/unet_forward_loop.py [n_traj ...]   -> per n: mean unet_kernel time by HIP events.  REPS=<n>: launches
per size (30); SECS=<s>: keep launching for about s seconds instead (power / throttle sampling sessions)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmd_amd import _lib, synth
if os.environ.get("MMD_AMD_LIB"):
    _lib.LIB_PATH = os.environ["MMD_AMD_LIB"]
from mmd_amd.temporal_unet import TemporalUnet

lib = _lib.load()
unet = TemporalUnet(two_per_workgroup_max=int(os.environ.get('MMD_AMD_UNET_NS2_MAX', '0')))   # (this TOOL's knob, passed on as mmd_unet_options: the library reads no environment)
unet.load_state_dict(synth.synth_unet_state_dict(0))
reps = int(os.environ.get("REPS", "30"))
for n in [int(a) for a in sys.argv[1:]] or [2048]:
    x = torch.randn(n, 64, 4, device="cuda")
    for _ in range(3):
        unet(x, 50)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    secs = float(os.environ.get("SECS", "0"))
    t0, done = time.perf_counter(), 0
    e0.record()
    while True:
        for _ in range(reps):
            unet(x, 50)
        done += reps
        if secs <= 0:
            break
        torch.cuda.synchronize()                 # (bounds the launch queue; one sync per `reps` launches)
        if time.perf_counter() - t0 >= secs:
            break
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / done * 1e3
    fl, mf = lib.mmd_unet_flops_per_trajectory() * n, lib.mmd_unet_mfma_flops_per_trajectory() * n
    hf = lib.mmd_unet_f16x2_flops_per_trajectory() * n
    busy_us = ((mf - hf) / 157.3e12 + 3.0 * hf / 2516.6e12) * 1e6      # MFMA issue time at the spec clock (bench.py's accounting)
    print(f"n={n:5d}: unet forward {us:8.1f} us  algorithmic {fl / us / 1e6:6.1f} TF  fp32-equivalent GEMM {mf / us / 1e6:6.1f} TF  "
          f"matrix pipe busy {busy_us / us:.3f}  [{os.environ.get('MMD_AMD_LIB', 'default lib')}]", flush=True)
