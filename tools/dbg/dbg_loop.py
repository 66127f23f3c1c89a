import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from math import ceil
import cases, gpu_common
from cases import rel_l2, H, D
from oracle import mmd_oracle as O
from mmd_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "empty_T50"
case = cases.sample_case(name)
xT, steps = cases.sample_inputs(case)
chain = gpu_common.hip_run_inference(case, xT, steps).cpu()
ref = cases.oracle_run_inference(case, clip_mode="always")
T = case["T"]
for r in range(T + 2):
    e = rel_l2(chain[r], ref[r])
    if e > 1e-5 or r in (0, T + 1):
        print("row", r, "i=", T - r, "rel", e, "maxabs", float((chain[r] - ref[r]).abs().max()))
# first guided step in isolation: start from oracle's chain row before it
r0 = T - ceil(0.5 * T) + 1      # chain row index of x before the first guided step (i = tsg-1)
x_prev = ref[r0].clone()
sd = O.state_dict_to_torch(synth.synth_unet_state_dict(0)); tb = O.schedule_tables(T)
gp = cases.guide_params(case["map"])
hc = cases.hard_conds_for(case["start"], case["goal"])
i = ceil(0.5 * T) - 1
# oracle pieces
eps = O.unet_forward(sd, x_prev, torch.full((case["B"],), i, dtype=torch.long))
x0 = (tb["sqrt_recip_alphas_cumprod"][i] * x_prev - tb["sqrt_recipm1_alphas_cumprod"][i] * eps).clamp(-1, 1)
mean = tb["posterior_mean_coef1"][i] * x0 + tb["posterior_mean_coef2"][i] * x_prev
print("mean range", float(mean.min()), float(mean.max()))
y = mean.clone()
guide = gpu_common.hip_guide(case["map"], [case["cons"]])
hardt = torch.stack([hc[0], hc[H - 1]])[None].cuda().contiguous()
yg = mean.clone().cuda()
for it in range(20):
    y = y + O.guide_grad(y, gp, case["cons"], clip_mode="always")
    y = O.apply_hard_conditioning(y, hc)
    guide.guide_steps(yg, hardt, 3, 1)
    e = rel_l2(yg.cpu(), y)
    d = (yg.cpu() - y).abs()
    idx = np.unravel_index(int(d.argmax()), d.shape)
    print("guide it", it, "rel", e, "maxabs", float(d.max()), "at", idx)
    if e > 1e-4:
        b, t, dd = idx
        print(" oracle", y[b, t], "hip", yg.cpu()[b, t])
        break
