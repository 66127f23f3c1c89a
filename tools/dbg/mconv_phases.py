"""Where a workgroup of mconv_kernel (csrc/unet_layers.hip) spends its clocks: the -DMCONV_TIMING build's per-phase clock64() sums of wave 0,
by (KIND, l_in, slice).  Build first (on any box with hipcc): tools/dbg/mconv_phases.py --build; run on the GPU: mconv_phases.py [n]"""
import ctypes as C
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "build_tmp", "libmmd_amd_timing.so")
if "--build" in sys.argv:
    import __graft_entry__ as g
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize", "-w",
                    "-DMCONV_TIMING"] + [os.path.join(ROOT, s) for s in g.SOURCES] + ["-o", LIB], check=True)
    sys.exit(0)
import numpy as np
import torch
from mmd_amd import _lib, synth
_lib.LIB_PATH = LIB
from mmd_amd.temporal_unet import TemporalUnet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
u = TemporalUnet(dim_mults=(1, 2, 4, 8))
u.load_state_dict(synth.synth_unet_state_dict(0, dim_mults=(1, 2, 4, 8)))
x = torch.randn(n, 64, 4, device="cuda")
lib = _lib.load()
lib.mmd_debug_mconv_clocks.restype, lib.mmd_debug_mconv_clocks.argtypes = C.c_int, [C.c_void_p]
clk = np.zeros((5, 4, 9, 8), dtype=np.uint64)
for _ in range(3):
    u(x, 5)
torch.cuda.synchronize()
assert lib.mmd_debug_mconv_clocks(clk.ctypes.data) == 0
REPS = 5
for _ in range(REPS):
    u(x, 5)
torch.cuda.synchronize()
assert lib.mmd_debug_mconv_clocks(clk.ctypes.data) == 0
names = ("loads", "convert", "gemm", "exch", "stats", "tail")   # (KIND 4: exch / stats / tail of both Conv1dBlocks, gemm of the first, the second GEMM under "tail")
print(f"n = {n}: mean clock64() ticks per workgroup (wave 0) by phase; workgroups per forward")
for kind in range(5):
    for li in range(4):
        for nit in range(9):
            t = clk[kind, li, nit].astype(np.float64)
            if t[7] == 0:
                continue
            wg = t[7]
            print(f"KIND {kind} l_in {8 << li:2d} cs {16 * nit:3d}: " + "  ".join(f"{nm} {t[i] / wg:8.0f}" for i, nm in enumerate(names)) +
                  f"  | whole {t[6] / wg:8.0f}  wgs/forward {wg / REPS:7.0f}")
