"""A few forwards of the option-1 (dim_mults (1, 2, 4, 8)) TemporalUnet on the layer-by-layer path, for rocprofv3 passes: layered_loop.py [n] [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmd_amd import synth
from mmd_amd.temporal_unet import TemporalUnet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
u = TemporalUnet(dim_mults=(1, 2, 4, 8))
u.load_state_dict(synth.synth_unet_state_dict(0, dim_mults=(1, 2, 4, 8)))
x = torch.randn(n, 64, 4, device="cuda")
for _ in range(reps):
    u(x, 5)
torch.cuda.synchronize()
