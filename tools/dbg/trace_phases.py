"""Phase timeline of unet_kernel from a -DMMD_TRACE side build (MMD_AMD_LIB=<that .so>): per tag, the time since the
previous tag of the same wave (100 MHz wall clock -> ns), averaged over blocks/waves.  Usage: trace_phases.py [n_traj]"""
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
ns1_max = int(os.environ.get("MMD_NS1_MAX", "256"))               # unet_kernel<1> (one per workgroup) up to here: the side build's -DMMD_NS1_MAX
nb = n if n <= min(ns1_max, ns2_max) else (n + 1) // 2 if n <= ns2_max else (n + 3) // 4
trace = torch.zeros(nb * 4 * 256, dtype=torch.int64, device="cuda")
for _ in range(3):
    unet(x, 50)
torch.cuda.synchronize()
lib.mmd_debug_set_trace.argtypes = [C.c_void_p]
assert lib.mmd_debug_set_trace(trace.data_ptr()) == 0
unet(x, 50)
torch.cuda.synchronize()
lib.mmd_debug_set_trace(None)
t = trace.cpu().numpy().reshape(nb, 4, 256).astype(np.float64) * 10.0   # ns
tags = [i for i in range(256) if (t[:, :, i] > 0).all()]
t0 = t[:, :, tags[0]].min()
print(f"n={n} blocks={nb}; tags {len(tags)}; kernel span {(t[:, :, tags[-1]].max() - t0) / 1e3:.1f} us")
names = {}
for base, nm in ((0, "D0"), (40, "D1")):
    for j, w in enumerate(("start (input staged)", "convA + res + gn", "convB + gn", "id convA + gn", "id convB + gn", "tail conv + hand-off")):
        names[base + j] = f"{nm} {w}"
for j, w in enumerate(("start", "rtb0 convA+res done", "gn+write+barrier", "rtb0 convB done", "gn+res", "id convA done", "id convB done", "gn+write -> tail")):
    names[136 + j] = f"U0 {w}"
for j, w in enumerate(("maxima + barrier", "convA (2 chunks, 4 barriers) + res + gn", "convB + gn", "id convA + gn", "id convB + gn", "tail + final block's input stored")):
    names[146 + j] = f"U1 {w}"
names.update({152: "U1 x chunk scaled + stored", 153: "  barrier", 154: "  chunk 0 taps + ring", 155: "  barrier + skip chunk stored", 156: "  barrier",
              157: "  chunk 1 taps"})
names.update({160: "U0 x0 scaled + stored", 161: "  barrier", 162: "  chunk 0 taps + ring", 163: "  barrier + x1 stored", 164: "  barrier", 165: "  chunk 1 taps"})
names.update({6: "D0 id convA (two per workgroup): start", 7: "  slab store", 8: "  barrier", 9: "  taps (60 MFMAs)", 10: "  weight preload issued"})
names.update({130: "-> U0 start", 131: "-> U1 start", 133: "final block + output store"})
# direct f16x2 body of downs.2 + mid (tags 90..93 are overwritten by every conv: the values are the LAST conv's, mid_block2 conv B)
names.update({80: "D2 start", 81: "D2 rtb0 convA+res+gn", 90: "last conv: start (prev gn done)", 91: "  ring + slab store", 92: "  barrier",
              93: "  MFMAs (5 taps x 4 chunks)", 98: "  GN + Mish (+ dyn max)"})
prev = None
print(f"{'tag':>4s} {'phase':28s} {'mean dt us':>10s} {'min':>8s} {'max':>8s}   cumulative(mean) us")
for tg in tags:
    cur = t[:, :, tg]
    if prev is not None:
        d = (cur - prev) / 1e3
        print(f"{tg:4d} {names.get(tg, ''):28s} {d.mean():10.2f} {d.min():8.2f} {d.max():8.2f}   {(cur.mean() - t0) / 1e3:8.1f}")
    prev = cur
# start skew between blocks
print("block start spread us:", (t[:, :, tags[0]].max() - t0) / 1e3)
# ---- where does the tail come from?  end time of every workgroup (last tag) against its placement guesses
end = (t[:, :, tags[-1]].max(axis=1) - t0) / 1e3
start = (t[:, :, tags[0]].min(axis=1) - t0) / 1e3
q = np.quantile(end, [0, 0.1, 0.5, 0.9, 1.0])
print(f"workgroup end time us: min {q[0]:.1f} p10 {q[1]:.1f} median {q[2]:.1f} p90 {q[3]:.1f} max {q[4]:.1f}; duration mean {np.mean(end - start):.1f}")
b = np.arange(nb)
for name, key in (("blockIdx % 8 (XCD)", b % 8), ("blockIdx // (nb/2) (dispatch half)", b // max(nb // 2, 1)), ("(blockIdx // 8) % 2", (b // 8) % 2)):
    print(f"  mean end by {name}: " + " ".join(f"{end[key == k].mean():.1f}" for k in np.unique(key)))
if nb >= 512:
    pair = np.abs(end[:nb // 2] - end[nb // 2:])
    print(f"  |end(b) - end(b + nb/2)| mean {pair.mean():.1f} us; corr(end(b), end(b+nb/2)) = {np.corrcoef(end[:nb // 2], end[nb // 2:])[0, 1]:.2f}")
