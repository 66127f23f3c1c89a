"""Four waves per trajectory (ddpm_guide_coop_kernel) against one (ddpm_guide_kernel) at the reference's call size, by constraint slots:
a T = 25 sampling call with EVERY step guided; guide_coop_max = 0 (the library's rule) / -1 (never cooperative)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import cases
import gpu_common
from mmd_amd import synth
from mmd_amd.diffusion_model import ddpm_sample_fn

H, T = 64, 25
model = gpu_common.hip_model(T)
for n_others in (0, 1, 3, 9, 15, 31, 63):
    N = max(n_others + 1, 2)
    starts, goals = synth.start_goal_circle(N, 0.45)
    paths = synth.straight_line_paths(starts, goals, H)
    for n in (64, 256):
        row = []
        for coop in (0, -1):
            model.guide_coop_max = coop
            guide = gpu_common.hip_guide("EnvHighways2D", [[cases.soft_group(paths, 0)]] if n_others else [[]])
            hc = cases.hard_conds_for(starts[0], goals[0])
            kw = dict(n_samples=n, horizon=H, return_chain=False, sample_fn=ddpm_sample_fn, guide=guide, n_guide_steps=20, t_start_guide=T,
                      noise_std_extra_schedule_fn=lambda t: 0.5, n_diffusion_steps_without_noise=1, seed=5)
            for _ in range(2):
                model.run_inference(None, hc, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                out = model.run_inference(None, hc, **kw)
            e1.record()
            torch.cuda.synchronize()
            row.append((e0.elapsed_time(e1) / 5 * 1e3, out.clone()))
        same = torch.equal(row[0][1], row[1][1])
        print(f"{n_others:2d} slots  n={n:4d}: {T + 1} guided steps  four waves per trajectory {row[0][0]:7.1f} us   one wave {row[1][0]:7.1f} us   "
              f"(per step {row[0][0] / (T + 1):5.1f} vs {row[1][0] / (T + 1):5.1f}; bitwise equal {same})")
model.guide_coop_max = 0
