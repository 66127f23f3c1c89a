"""Forward time of the layer-by-layer TemporalUnet path (csrc/unet_layers.hip) by batch size: option 1 = UNET_DIM_MULTS[1] = (1, 2, 4, 8),
and option 0 forced onto the path (TemporalUnet(layered=True)) beside the fused kernel.  MMD_AMD_LAYERED_VALU=1 (read HERE by this tool and passed to TemporalUnet(layered_valu=...) -> mmd_unet_options; the library reads no environment): the
vector-ALU Conv1dBlock kernel instead of conv5_mfma_kernel.  Usage: layered_time.py [n ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmd_amd import synth
from mmd_amd.temporal_unet import TemporalUnet


def net(dm, layered=None):
    u = TemporalUnet(dim_mults=dm, layered=bool(layered), layered_valu=os.environ.get('MMD_AMD_LAYERED_VALU', '0') == '1')
    u.load_state_dict(synth.synth_unet_state_dict(0, dim_mults=dm))
    return u


nets = (("option1 (1,2,4,8) layered", net((1, 2, 4, 8))), ("option0 (1,2,4) layered", net((1, 2, 4), True)), ("option0 fused", net((1, 2, 4), False)))
for n in [int(a) for a in sys.argv[1:]] or [64, 256, 1024, 4096]:
    x = torch.randn(n, 64, 4, device="cuda")
    row = []
    for name, u in nets:
        for _ in range(3):
            u(x, 5)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            u(x, 5)
        e1.record()
        torch.cuda.synchronize()
        row.append(f"{name} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us")
    print(f"n={n:5d}: " + "   ".join(row) + f"   [MMD_AMD_LAYERED_VALU={os.environ.get('MMD_AMD_LAYERED_VALU', '0')}]", flush=True)
