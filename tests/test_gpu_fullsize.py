"""-m gpu: BASELINE.json's full sizes (32 robots x 64 samples, H=64, T=100) through size-independent properties,
plus a spot check of the (well-conditioned) prior against the oracle on a random subset of the 2048 trajectories."""
from math import ceil

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mmd_amd import _lib, synth          # noqa: E402
from oracle import mmd_oracle as O       # noqa: E402
import cases                             # noqa: E402
from cases import H, D, rel_l2           # noqa: E402

R, B, T = 32, 64, 100


@pytest.fixture(scope="module")
def headline():
    import gpu_common
    from mmd_amd.multi_robot import MultiRobotSampler
    model = gpu_common.hip_model(T)
    starts, goals = synth.start_goal_circle(R, 0.8)
    s = MultiRobotSampler(model, starts, goals, env_id="EnvEmpty2D", n_samples=B)
    paths = torch.from_numpy(synth.straight_line_paths(starts, goals, H)).cuda()
    return model, s, starts, goals, paths


def test_headline_round_properties(headline):
    model, s, starts, goals, paths = headline
    s.set_other_paths(paths)
    a = s.sample(seed=11)
    assert a.shape == (R * B, H, D) and torch.isfinite(a).all()
    # determinism per seed, sensitivity to the seed
    assert torch.equal(a, s.sample(seed=11)) and not torch.equal(a, s.sample(seed=12))
    # hard conditioning: rows 0 / H-1 are exactly the normalised start / goal of the robot that owns the sample
    hs = s.hard_conds[0].repeat_interleave(B, 0)
    hg = s.hard_conds[H - 1].repeat_interleave(B, 0)
    assert torch.equal(a[:, 0], hs) and torch.equal(a[:, -1], hg)
    # the sampler stays in the normalised box up to the guide's step size
    assert float(a.abs().max()) < 1.5
    # best-path pick + the all-pairs constraint table round trip
    bp = s.best_paths(a)
    assert bp.shape == (R, H, 2) and torch.allclose(bp[:, 0].cpu(), torch.from_numpy(starts), atol=1e-5)


def test_headline_batched_equals_sharded(headline):
    """The property the multi-GPU sharding relies on, at full size: sampling robots [8,16) alone (as rank 1 of 4 would)
    gives bit-identical trajectories to the same robots inside the 32-robot batch (Philox noise is keyed by the global
    trajectory index only when the shard offsets are equal, so inject the noise)."""
    model, s, starts, goals, paths = headline
    from mmd_amd.multi_robot import MultiRobotSampler
    Ts = 25                                   # short schedule keeps the injected-noise tensor small
    import gpu_common
    m25 = gpu_common.hip_model(Ts)
    full = MultiRobotSampler(m25, starts, goals, env_id="EnvEmpty2D", n_samples=B)
    part = MultiRobotSampler(m25, starts, goals, env_id="EnvEmpty2D", n_samples=B, rank=1, world_size=4)
    full.set_other_paths(paths)
    part.set_other_paths(paths)
    xT = torch.from_numpy(synth.synth_noise(80, (R * B, H, D))).cuda()
    steps = torch.from_numpy(synth.synth_noise(81, (Ts + 1, R * B, H, D))).cuda()
    a = full.sample(x_init=xT, step_noise=steps)
    sl = slice(part.robot0 * B, (part.robot0 + part.n_local) * B)
    b = part.sample(x_init=xT[sl].contiguous(), step_noise=steps[:, sl].contiguous())
    assert torch.equal(a[sl], b)


@pytest.mark.parametrize("world,rank", [(2, 1), (4, 1), (4, 3), (8, 0), (8, 5)])
def test_headline_sharded_equals_unsharded_with_inkernel_noise(headline, world, rank):
    """per-robot outputs bitwise identical for G = 1, 2, 4, 8 -- in the PRODUCTION noise path.  The in-kernel
    Philox draws are keyed by (seed, draw, GLOBAL trajectory index), so rank `rank` of `world`, sampling only its robots
    with the same seed, reproduces exactly the rows those robots have in the one-GPU 32-robot run (x_T and every step's
    noise), and different robots never share noise."""
    model, s, starts, goals, paths = headline
    from mmd_amd.multi_robot import MultiRobotSampler
    import gpu_common
    m25 = gpu_common.hip_model(25)
    full = MultiRobotSampler(m25, starts, goals, env_id="EnvEmpty2D", n_samples=B)
    part = MultiRobotSampler(m25, starts, goals, env_id="EnvEmpty2D", n_samples=B, rank=rank, world_size=world)
    full.set_other_paths(paths)
    part.set_other_paths(paths)
    a = full.sample(seed=2024)
    b = part.sample(seed=2024)
    sl = slice(part.robot0 * B, (part.robot0 + part.n_local) * B)
    assert torch.equal(a[sl], b), (world, rank)
    # robots do not share noise: the first two robots' batches differ already in their initial draws
    x0 = m25.p_sample_loop((2 * B, H, D), {}, 0, seed=2024)
    assert not torch.equal(x0[:B], x0[B:])
    # and the shard's x_T is the matching slice of the global x_T
    xg = m25.p_sample_loop((R * B, H, D), {}, 0, seed=2024)
    xs = m25.p_sample_loop((part.n_local * B, H, D), {}, 0, seed=2024, traj_index_base=part.robot0 * B)
    assert torch.equal(xg[sl], xs)


def test_planners_with_the_same_seed_draw_independent_noise():
    """N planners built with the default seed must not sample identical trajectories (the reference seeds once
    and every call advances one global RNG)."""
    from mmd_amd.planners import MPD
    kw = dict(model_id="EnvEmpty2D-RobotPlanarDisk", planner_alg="diffusion_prior", n_samples=4, device="cuda",
              model_state_dict=synth.synth_unet_state_dict(0), model_args=dict(n_diffusion_steps=25), seed=18)
    s, g = torch.tensor([-0.5, 0.0]), torch.tensor([0.5, 0.0])
    p1, p2 = MPD(start_state_pos=s, goal_state_pos=g, **kw), MPD(start_state_pos=s, goal_state_pos=g, **kw)
    o1, o2, o1b = p1(s, g).trajs_iters[-1], p2(s, g).trajs_iters[-1], p1(s, g).trajs_iters[-1]
    assert not torch.equal(o1, o2) and not torch.equal(o1, o1b)


def test_headline_prior_spot_check_vs_oracle(headline):
    """planner_alg 'diffusion_prior' at full size: the unguided chain is well conditioned (sens 7e-7), so 6 of the 2048
    trajectories are compared with the oracle end to end at the north-star tolerance."""
    model, s, starts, goals, paths = headline
    from mmd_amd.diffusion_model import ddpm_sample_fn
    xT = torch.from_numpy(synth.synth_noise(82, (R * B, H, D)))
    steps = torch.from_numpy(synth.synth_noise(83, (T + 1, R * B, H, D)))
    out = model.run_inference(None, s.hard_conds, n_samples=B, n_robots=R, horizon=H, return_chain=False,
                              sample_fn=ddpm_sample_fn, guide=None, noise_std_extra_schedule_fn=lambda t: 0.5,
                              n_diffusion_steps_without_noise=1, warm_start_path_b=xT.cuda(), step_noise=steps.cuda()).cpu()
    sd = O.state_dict_to_torch(synth.synth_unet_state_dict(0))
    tb = O.schedule_tables(T)
    rng = np.random.Generator(np.random.PCG64(7))
    for idx in rng.choice(R * B, size=6, replace=False):
        r = int(idx) // B
        hc = cases.hard_conds_for(starts[r], goals[r])
        ref = O.p_sample_loop(sd, tb, xT[idx:idx + 1], hc, T, steps[:, idx:idx + 1], guide=None, noise_std_extra=0.5,
                              n_diffusion_steps_without_noise=1)[-1]
        assert rel_l2(out[idx:idx + 1], ref) < 1e-3, (int(idx), rel_l2(out[idx:idx + 1], ref))


def test_headline_guided_step_and_guide_vs_oracle(headline):
    """The guided kernel at the headline size -- the uniform-radius (qx, qy)-only LDS staging with 31 slots x 32 robots in one
    launch, which the small parity cases (B = 4 .. 16, one robot) never reach -- against the oracle on 8 randomly chosen
    trajectories from >= 4 robots (1953 soft-constraint points each): one guide evaluation (max-abs 2e-6, g5's bound), the
    20-iteration guide_gradient_steps (2e-4, the bound of test_guide_20_steps_vs_oracle: 20 norm-clipped iterations) and one
    teacher-forced guided DDPM step at loop index i = 49 (UNet + posterior mean + 20 guide iterations + noise; the
    north-star 1e-3, or 1.5 x the oracle's own response to a rounding-sized perturbation of eps where that is larger).  ref: sample_functions.py:40-107, guides.py:180-226."""
    import parity_log
    model, s, starts, goals, paths = headline
    s.set_other_paths(paths)
    gp = cases.guide_params("EnvEmpty2D")
    paths_np = synth.straight_line_paths(starts, goals, H)
    rng = np.random.Generator(np.random.PCG64(21))
    robots = rng.choice(R, size=4, replace=False)
    picks = [(int(r), int(b)) for r in robots for b in rng.choice(B, size=2, replace=False)]
    assert len(picks) == 8 and len({r for r, _ in picks}) >= 4
    # a mid-chain looking state: half-amplitude noise with the robots' start / goal rows
    x = torch.from_numpy(synth.synth_noise(90, (R * B, H, D))) * 0.5
    x[:, 0] = s.hard_conds[0].cpu().repeat_interleave(B, 0)
    x[:, -1] = s.hard_conds[H - 1].cpu().repeat_interleave(B, 0)
    hard = torch.stack((s.hard_conds[0], s.hard_conds[H - 1]), dim=1).contiguous()     # [R, 2, D]
    # (a) one guide evaluation, (b) 20 guide iterations with hard conditioning
    g1 = s.guide(x.cuda()).cpu()
    y20 = x.clone().cuda()
    s.guide.guide_steps(y20, hard, _lib.HARD_ROWS_START_GOAL, 20)
    y20 = y20.cpu()
    # (c) one guided DDPM step from x at i = 49 with injected noise
    noise = torch.from_numpy(synth.synth_noise(91, (R * B, H, D)))
    ys = x.clone().cuda()
    model.sample_step(ys, s.hard_conds, 49, guide=s.guide, n_guide_steps=20, t_start_guide=ceil(0.5 * T),
                      noise_std_extra_schedule_fn=lambda t: 0.5, n_robots=R, noise=noise.cuda())
    ys = ys.cpu()
    sd = O.state_dict_to_torch(synth.synth_unet_state_dict(0))
    tb = O.schedule_tables(T)
    import gpu_common
    judge = gpu_common.GuidedStepJudge(model, s.guide, x, s.hard_conds, 49, ceil(0.5 * T), R, noise, ys)
    for r, b in picks:
        idx = r * B + b
        grp = cases.soft_group(paths_np, r)
        assert grp.q.shape[0] == 31 * 63
        hc = cases.hard_conds_for(starts[r], goals[r])
        guide = lambda z, grp=grp: O.guide_grad(z, gp, [grp], clip_mode="always")      # noqa: E731
        xi = x[idx:idx + 1]
        e1 = float((g1[idx:idx + 1] - guide(xi)).abs().max())
        parity_log.record("fullsize_guide_single_eval", f"robot{r}_sample{b}", None, e1, bound=2e-6)
        assert e1 < 2e-6, (r, b, e1)
        z = xi.clone()
        for _ in range(20):
            z = O.apply_hard_conditioning(z + guide(z), hc)
        e20 = rel_l2(y20[idx:idx + 1], z)
        parity_log.record("fullsize_guide_20_steps", f"robot{r}_sample{b}", None, e20, bound=2e-4)
        assert e20 < 2e-4, (r, b, e20)
        # 20 norm-clipped iterations with hinge constraints are not continuous in eps: a trajectory that sits on a switching
        # surface moves by more than 1e-3 under a rounding-sized change of the UNet output.  Such a step must be SHOWN to be a
        # branch flip (the kernel's decision trace differs from the oracle's and the oracle on the kernel's decisions agrees with
        # the kernel to 1e-4: gpu_common.GuidedStepJudge); 1.5 x the oracle's own response to a relative 2e-6 perturbation of eps
        # is the fallback yardstick only.  Everywhere else the north-star 1e-3 stands.
        judge.check("fullsize_guided_step_teacher_forced", f"robot{r}_sample{b}", idx, sd, tb, gp, [grp], hc, 1000 + idx)


def test_guide_zero_weights_is_identity(headline):
    """Idempotence-type property: with every gradient weight at 0 the guide leaves x untouched (only hard conditioning)."""
    import gpu_common
    from mmd_amd.guides import GuideManagerTrajectoriesWithVelocity
    g = GuideManagerTrajectoriesWithVelocity(gpu_common.dataset(), env_id="EnvHighways2D", n_robots=4,
                                             weight_grad_cost_collision=0.0, weight_grad_cost_smoothness=0.0)
    x = (torch.from_numpy(synth.synth_noise(84, (4 * B, H, D))) * 0.7).cuda()
    y = x.clone()
    g.guide_steps(y, torch.zeros(4, 2, D, device="cuda"), 0, 20)
    assert torch.equal(x, y)


def test_large_batch_and_ragged_tail():
    """8192 trajectories (4 waves of workgroups) and a batch that is not a multiple of the 4-sample workgroup tile."""
    import gpu_common
    model = gpu_common.hip_model(T)
    x = torch.from_numpy(synth.synth_noise(85, (8192 + 3, H, D))).cuda()
    big = model.model(x, 17)
    assert torch.isfinite(big).all()
    small = model.model(x[4096:4096 + 7].contiguous(), 17)
    assert torch.equal(big[4096:4096 + 7], small)          # per-sample results do not depend on the batch they sit in
