"""What would a split-bf16 (bf16x3 / bf16x6, fp32 accumulate) TemporalUnet cost in accuracy?  CPU emulation: every conv of
the oracle UNet is replaced by the sum of bf16-rounded partial products."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, torch.nn.functional as F
from oracle import mmd_oracle as O
from mmd_amd import synth
from cases import rel_l2

def split(t, n):
    parts, r = [], t.clone()
    for _ in range(n):
        p = r.bfloat16().float(); parts.append(p); r = r - p
    return parts

def make_conv(orig, mode):
    def conv(x, w, b=None, **kw):
        n = 2 if mode == 3 else 3
        xs, ws = split(x, n), split(w, n)
        pairs = [(0, 0), (0, 1), (1, 0)] if mode == 3 else [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]
        out = None
        for i, j in pairs:
            o = orig(xs[i], ws[j], None, **kw)
            out = o if out is None else out + o
        return out if b is None else out + b.view(1, -1, 1)
    return conv

sd = O.state_dict_to_torch(synth.synth_unet_state_dict(0))
x = torch.from_numpy(synth.synth_noise(5, (16, 64, 4)))
ref = {t: O.unet_forward(sd, x, torch.full((16,), t)) for t in (0, 37, 99)}
c1, ct = F.conv1d, F.conv_transpose1d
for mode in (3, 6):
    F.conv1d, F.conv_transpose1d = make_conv(c1, mode), make_conv(ct, mode)
    try:
        for t in (0, 37, 99):
            out = O.unet_forward(sd, x, torch.full((16,), t))
            print(f"bf16x{mode} t={t}: rel-L2 vs fp32 {rel_l2(out, ref[t]):.2e}")
    finally:
        F.conv1d, F.conv_transpose1d = c1, ct
