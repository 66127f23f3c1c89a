"""unet_kernel<1> (launches of <= 128 trajectories) against unet_kernel<2> on the same rows: where do they differ?"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmd_amd import _lib, synth
if os.environ.get("MMD_AMD_LIB"):
    _lib.LIB_PATH = os.environ["MMD_AMD_LIB"]
from mmd_amd.temporal_unet import TemporalUnet

unet = TemporalUnet()
unet.load_state_dict(synth.synth_unet_state_dict(0))
x = torch.from_numpy(synth.synth_noise(41, (300, 64, 4))).cuda()
big = unet(x, 37)
for n in (1, 2, 3, 5):
    small = unet(x[:n].contiguous(), 37)
    d = (small - big[:n]).abs()
    print(f"n={n}: finite {bool(torch.isfinite(small).all())}  max abs diff {float(d.max()):.3e}  rel {float(d.norm() / big[:n].norm()):.3e}")
    if float(d.max()) > 0:
        bad = (d > 0).nonzero()
        print("   differing elements:", bad.shape[0], "of", d.numel(), " rows", sorted(set(bad[:, 0].tolist())))
