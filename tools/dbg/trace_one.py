"""Stage timeline of unet_kernel_1 from a -DMMD_TRACE side build: python tools/dbg/trace_one.py [n_traj]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mmd_amd import _lib, synth
_lib.LIB_PATH = os.environ["MMD_AMD_LIB"]
os.environ["MMD_AMD_UNET_KERNEL"] = "one"
from mmd_amd.temporal_unet import TemporalUnet
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = _lib.load()
unet = TemporalUnet(); unet.load_state_dict(synth.synth_unet_state_dict(0))
x = torch.randn(n, 64, 4, device="cuda")
trace = torch.zeros(n * 4 * 256, dtype=torch.int64, device="cuda")
for _ in range(3): unet(x, 50)
torch.cuda.synchronize()
lib.mmd_debug_set_trace.argtypes = [C.c_void_p]
assert lib.mmd_debug_set_trace(trace.data_ptr()) == 0
unet(x, 50); torch.cuda.synchronize(); lib.mmd_debug_set_trace(None)
t = trace.cpu().numpy().reshape(n, 4, 256).astype(np.float64) * 10.0
names = {200: "start", 201: "downs.0 done", 202: "downs.1 done", 203: "downs.2+mid done", 204: "ups.0 done", 205: "ups.1 done", 206: "final done"}
tags = [k for k in sorted(names) if (t[:, :, k] > 0).all()]
t0 = t[:, :, tags[0]].min()
prev = None
for k in tags:
    cur = t[:, :, k]
    if prev is not None:
        d = (cur - prev) / 1e3
        print(f"{names[k]:20s} {d.mean():8.2f} us (min {d.min():.2f} max {d.max():.2f})   cumulative {(cur.mean() - t0) / 1e3:8.1f}")
    prev = cur
print(f"n={n}: span {(t[:, :, tags[-1]].max() - t0) / 1e3:.1f} us")
