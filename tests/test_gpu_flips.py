"""-m gpu: the guide's DISCRETE decisions.  A guided ddpm_sample_fn step is 20 norm-clipped gradient iterations
over hinge costs (sample_functions.py:89-107, guides.py:180-259): not continuous in its input wherever a constraint point enters or
leaves its radius (cost_functions.py:305-312), a collision hinge switches (distance_fields.py:110-135) or the nearest SDF cell
changes (grid_map_sdf.py:84-114).  The measurement hook mmd_debug_ddpm_step_trace (include/mmd_amd_debug.h) makes the step kernel
write those decisions per guide iteration and support point; the oracle exposes its own (guide_decisions) and can be run on GIVEN
ones (guide_grad_forced).  Proven here:
  * the trace instantiation reproduces the production step bit for bit, whatever launch shape production picks;
  * on EVERY trajectory of a batch -- also those whose step differs from the oracle by more than the north-star 1e-3 -- the oracle
    run on the kernel's decisions agrees with the kernel up to the fp32 rounding of the two evaluations (each measured against the same
    iterations in float64, decisions frozen: the 20 norm-clipped steps amplify rounding 5 .. 350 x): the arithmetic is the
    reference's, what can differ is a branch;
  * where the step does differ by more than 1e-3 the first differing decision is located (iteration, kind, support point)."""
from math import ceil

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mmd_amd import synth                # noqa: E402
from oracle import mmd_oracle as O       # noqa: E402
import cases                             # noqa: E402
from cases import H, D, rel_l2           # noqa: E402


def _highways(B, T=25, R=2):
    import gpu_common as gc
    starts, goals, soft3, hard = cases.highways_case()
    paths = synth.straight_line_paths(starts, goals, H)
    groups = [[cases.soft_group(paths, r), hard] for r in range(R)]
    guide = gc.hip_guide("EnvHighways2D", groups, n_robots=R)
    hc = {0: torch.stack([cases.hard_conds_for(starts[r], goals[r])[0] for r in range(R)]),
          H - 1: torch.stack([cases.hard_conds_for(starts[r], goals[r])[H - 1] for r in range(R)])}
    return gc.hip_model(T), guide, groups, hc, starts, goals


@pytest.mark.parametrize("B", [8, 320])        # 16 trajectories: the cooperative step kernel; 640: the one-wave kernel
def test_trace_instantiation_equals_production_step_bitwise(B):
    import gpu_common as gc
    model, guide, groups, hc, _, _ = _highways(B)
    x = torch.from_numpy(synth.synth_noise(300, (2 * B, H, D))) * 0.5
    nz = torch.from_numpy(synth.synth_noise(301, (2 * B, H, D)))
    y = x.clone().cuda()
    model.sample_step(y, hc, 9, guide=guide, n_guide_steps=20, t_start_guide=13, noise_std_extra_schedule_fn=lambda t: 0.5,
                      n_robots=2, noise=nz.cuda())
    yt, mu, gchain, tr = gc.hip_step_with_trace(model, x, hc, 9, guide, 13, 2, nz)
    assert torch.equal(yt, y.cpu())
    assert gchain.shape == (20, 2 * B, H, D) and tr.shape == (20, 2 * B, H, 12)
    # the chain's last state + the step's noise = the result (sample_functions.py:86), rows 0 / H-1 pinned
    sigma = float(torch.exp(0.5 * O.schedule_tables(25)["posterior_log_variance_clipped"][9]))
    assert rel_l2(gchain[-1][:, 1:-1] + sigma * nz[:, 1:-1] * 0.5, yt[:, 1:-1]) < 1e-6
    assert int(tr[..., 1].max()) > 0 and int(((tr[..., 1] >> 20) & 15).max()) == 2      # two constraint groups per robot


def test_oracle_on_the_kernels_decisions_equals_the_kernel_on_every_trajectory():
    """16 + 16 trajectories of two Highways robots (soft constraints of nine other robots + one hard vertex constraint), guided
    steps i = 12 (the first) and i = 3: per trajectory the plain oracle error, the oracle-on-kernel-decisions error and the fp32
    rounding of both evaluations along the kernel's decision path (attribute_guided_step).  For ALL of them the oracle on the kernel's
    decisions agrees with the kernel up to that rounding; where the plain error is larger the first differing decision is named."""
    import gpu_common as gc
    import parity_log
    B = 16
    model, guide, groups, hc, starts, goals = _highways(B)
    sd, tb, gp = O.state_dict_to_torch(synth.synth_unet_state_dict(0)), O.schedule_tables(25), cases.guide_params("EnvHighways2D")
    n_diff, worst_ratio, bad = 0, 0.0, []
    for i, seed in ((12, 310), (3, 312)):
        x = torch.from_numpy(synth.synth_noise(seed, (2 * B, H, D))) * 0.5
        for r in range(2):
            x[r * B:(r + 1) * B, 0], x[r * B:(r + 1) * B, -1] = hc[0][r], hc[H - 1][r]
        nz = torch.from_numpy(synth.synth_noise(seed + 1, (2 * B, H, D)))
        y, mu, gchain, tr = gc.hip_step_with_trace(model, x, hc, i, guide, 13, 2, nz)
        for idx in range(2 * B):
            r = idx // B
            hcr = cases.hard_conds_for(starts[r], goals[r])
            slots = [O.slot_table(g).shape[0] for g in groups[r]]
            a = gc.attribute_guided_step(y[idx:idx + 1], mu[idx:idx + 1], gchain[-1, idx:idx + 1], gc.decode_trace(tr[:, idx], slots),
                                         x[idx:idx + 1], nz[idx:idx + 1], i, 13, sd, tb, gp, groups[r], hcr)
            first, bound = a["first"], gc.rounding_bound(a)
            n_diff += first is not None
            parity_log.record("oracle_on_kernel_decisions", f"i{i}_traj{idx}", i, a["ferr"], bound=bound, plain_err=a["err"],
                              fp32_rounding_oracle=a["d_o32"], fp32_rounding_kernel=a["d_hip"],
                              first_difference=None if first is None else f"iteration {first[0]}: {first[1]} at t={first[2]}")
            row = f"i={i} traj={idx}: err {a['err']:.2e} forced {a['ferr']:.2e} bound {bound:.2e} fp32 rounding oracle {a['d_o32']:.2e} kernel {a['d_hip']:.2e} first {first}"
            print("   " + row)
            if not a["ferr"] < bound:
                bad.append("forced error over the rounding bound: " + row)
            # the kernel's own rounding along its path is of the size class of the fp32 oracle's: two different fp32 evaluations (rsq
            # instead of sqrt + divide, contracted fmas) put through an amplification that is itself heavy-tailed -- measured ratio
            # <= 13 over these 64 trajectory-steps; anything beyond 1e-4 AND 30 x would be an arithmetic defect
            worst_ratio = max(worst_ratio, a["d_hip"] / max(a["d_o32"], 2e-6))
            if not a["d_hip"] < max(1e-4, 30.0 * max(a["d_o32"], 2e-6)):
                bad.append("kernel rounding >> oracle rounding: " + row)
            # identical decisions at every iteration: the plain error is arithmetic only, i.e. the same bound
            if first is None and not a["err"] < bound:
                bad.append("identical decisions, error over the rounding bound: " + row)
    print(f"{n_diff} of {4 * B} trajectory-steps take a different decision somewhere in their 20 iterations; kernel / oracle fp32 rounding <= {worst_ratio:.1f}")
    assert not bad, "\n".join(bad)
