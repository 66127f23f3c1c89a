"""Where does the kernel's guided step leave the oracle's?  For the batch of tests/test_gpu_flips.py (two Highways robots, soft + hard
constraints) per trajectory and guide iteration: |x_k(HIP) - x_k(oracle)| and |x_k(HIP) - x_k(oracle on HIP's decisions)|, the first
iteration whose decision sets differ, and the support point / term that carries the largest difference.  Usage: flip_probe.py [i] [seed]"""
import os
import sys
from math import ceil

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch                                    # noqa: E402
import cases                                    # noqa: E402
import gpu_common as gc                         # noqa: E402
from mmd_amd import synth                       # noqa: E402
from oracle import mmd_oracle as O              # noqa: E402
from test_gpu_flips import _highways            # noqa: E402

H, D, B = 64, 4, 16
i = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 310
model, guide, groups, hc, starts, goals = _highways(B)
sd, tb, gp = O.state_dict_to_torch(synth.synth_unet_state_dict(0)), O.schedule_tables(25), cases.guide_params("EnvHighways2D")
x = torch.from_numpy(synth.synth_noise(seed, (2 * B, H, D))) * 0.5
for r in range(2):
    x[r * B:(r + 1) * B, 0], x[r * B:(r + 1) * B, -1] = hc[0][r], hc[H - 1][r]
nz = torch.from_numpy(synth.synth_noise(seed + 1, (2 * B, H, D)))
y, mu, gchain, tr = gc.hip_step_with_trace(model, x, hc, i, guide, 13, 2, nz)
t = max(i, 0)
for idx in range(2 * B):
    r = idx // B
    hcr = cases.hard_conds_for(starts[r], goals[r])
    slots = [O.slot_table(g).shape[0] for g in groups[r]]
    hip_sets = gc.decode_trace(tr[:, idx], slots)
    xi = x[idx:idx + 1]
    eps = O.unet_forward(sd, xi, torch.full((1,), t, dtype=torch.long))
    x0 = (tb["sqrt_recip_alphas_cumprod"][t] * xi - tb["sqrt_recipm1_alphas_cumprod"][t] * eps).clamp(-1, 1)
    mean = tb["posterior_mean_coef1"][t] * x0 + tb["posterior_mean_coef2"][t] * xi     # (pinned after an iteration, not before the first)
    e_mu = float((mu[idx:idx + 1] - mean).abs().max())
    xo, xf = mean.clone(), mean.clone()
    rows = []
    first = None
    for k in range(20):
        own = O.guide_decisions(xo, gp, groups[r])
        d = gc.first_set_difference(hip_sets[k], own)
        if d is not None and first is None:
            first = (k,) + d
        xo = O.apply_hard_conditioning(xo + O.guide_grad(xo, gp, groups[r], clip_mode="always"), hcr)
        xf = O.apply_hard_conditioning(xf + O.guide_grad_forced(xf, gp, groups[r], hip_sets[k]), hcr)
        do, df = (gchain[k, idx] - xo[0]).abs(), (gchain[k, idx] - xf[0]).abs()
        rows.append((float(do.max()), float(df.max()), int(df.max(-1)[0].argmax())))
    worst = max(rows, key=lambda v: v[1])
    if worst[1] > 3e-5 or first is not None:
        print(f"traj {idx}: |mu diff| {e_mu:.1e}; first set difference {first}; per iteration max|HIP - oracle| / max|HIP - forced| (t of the latter):")
        print("   " + " ".join(f"{k}:{a:.0e}/{b:.0e}@{tt}" for k, (a, b, tt) in enumerate(rows)))
        # at the iteration where the forced run first leaves the kernel by > 3e-5: which term?
        kbad = next((k for k, v in enumerate(rows) if v[1] > 3e-5), None)
        if kbad is not None:
            tt = rows[kbad][2]
            w = tr[kbad, idx, tt]
            print(f"   iteration {kbad} t={tt}: flags {int(w[1]):#x} cell {int(w[0])} masks {[hex(int(v)) for v in w[2:10]]} active {int(w[10])}")
print("done")
