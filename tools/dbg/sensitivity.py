"""How sensitive is the REFERENCE ALGORITHM ITSELF (oracle restatement, bit-identical UNet arithmetic to the reference
on CPU) to fp32-rounding-sized perturbations?  Perturb the UNet output by a relative 1e-6 (the size of a different
summation order) at every step and compare final trajectories."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from math import ceil
import torch
import cases
from cases import rel_l2
from oracle import mmd_oracle as O
from mmd_amd import synth

def run(case, pert, seed=0):
    sd = O.state_dict_to_torch(synth.synth_unet_state_dict(0))
    tb = O.schedule_tables(case["T"])
    gp = cases.guide_params(case["map"], case.get("cutoff", 0.05))
    xT, steps = cases.sample_inputs(case)
    gen = torch.Generator().manual_seed(seed)
    orig = O.unet_forward
    def noisy(sd_, x, t, n_levels=3):
        e = orig(sd_, x, t, n_levels)
        return e * (1 + pert * torch.randn(e.shape, generator=gen)) if pert else e
    O.unet_forward = noisy
    try:
        return O.p_sample_loop(sd, tb, xT, cases.hard_conds_for(case["start"], case["goal"]), case["T"], steps,
                               guide=lambda x: O.guide_grad(x, gp, case["cons"]), n_guide_steps=20,
                               t_start_guide=ceil(0.5 * case["T"]), noise_std_extra=0.5, n_diffusion_steps_without_noise=1)
    finally:
        O.unet_forward = orig

for name in sys.argv[1:] or cases.SAMPLE_CASES:
    case = cases.sample_case(name)
    base = run(case, 0.0)
    for pert in (1e-7, 1e-6):
        p = run(case, pert)
        T = case["T"]
        rows = [T // 2, T // 2 + 2, T // 2 + 6, T + 1]
        print(name, "pert", pert, {r: f"{rel_l2(p[r], base[r]):.2e}" for r in rows}, flush=True)
