"""Per-layer UNet timing (HIP events inside libmmd_amd.so) for n trajectories.  MMD_AMD_LIB selects the .so."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmd_amd import _lib, synth
if os.environ.get("MMD_AMD_LIB"):
    _lib.LIB_PATH = os.environ["MMD_AMD_LIB"]
from mmd_amd.temporal_unet import TemporalUnet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
lib = _lib.load()
unet = TemporalUnet()
unet.load_state_dict(synth.synth_unet_state_dict(0))
x = torch.randn(n, 64, 4, device="cuda")
eps = torch.empty_like(x)
ws = unet.workspace(n, x.device)
nl = lib.mmd_unet_num_layers()
ms = (C.c_float * nl)()
for _ in range(2):
    _lib.check(lib.mmd_unet_profile(unet.handle(100), x.data_ptr(), 50, eps.data_ptr(), n, ws.data_ptr(), ws.numel(), 20,
                                    ms, _lib.current_stream_ptr()))
tot = 0.0
tf = 0.0
for i in range(nl):
    fl = lib.mmd_unet_layer_flops(i) * n
    tot += ms[i]
    tf += fl
    print(f"{i:2d} {lib.mmd_unet_layer_name(i).decode():5s} {ms[i] * 1e3:8.2f} us  {fl / (ms[i] * 1e-3) / 1e12:7.1f} TF")
print(f"TOTAL {tot * 1e3:.1f} us  {tf / (tot * 1e-3) / 1e12:.1f} TF  ({os.environ.get('MMD_AMD_LIB', 'default lib')})")
# end-to-end forward without events
torch.cuda.synchronize()
import time
for _ in range(3):
    unet(x, 50)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    unet(x, 50)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print(f"forward wall {dt * 1e6:.1f} us  {tf / dt / 1e12:.1f} TF")
