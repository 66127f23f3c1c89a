"""Helpers for the -m gpu parity tests: build the HIP-side objects for a golden case."""
from math import ceil

import numpy as np
import torch

from mmd_amd import synth
from mmd_amd.constraints import CostConstraint
from mmd_amd.diffusion_model import GaussianDiffusionModel, ddpm_sample_fn
from mmd_amd.guides import GuideManagerTrajectoriesWithVelocity
from mmd_amd.normalization import TrajectoryDatasetFacade
from mmd_amd.temporal_unet import TemporalUnet

_MODELS = {}


def hip_model(T, weights_seed=0):
    """weights_seed: a synth seed, or "g19" = the trained state dict of golden g19 (cases.named_state_dict)."""
    key = (T, weights_seed)
    if key not in _MODELS:
        import cases
        unet = TemporalUnet(state_dim=4, n_support_points=64, unet_input_dim=32, dim_mults=(1, 2, 4))
        unet.load_state_dict(cases.named_state_dict(weights_seed))
        _MODELS[key] = GaussianDiffusionModel(model=unet, variance_schedule="exponential", n_diffusion_steps=T,
                                              predict_epsilon=True)
    return _MODELS[key]


def dataset():
    return TrajectoryDatasetFacade(synth.NORM_MINS, synth.NORM_MAXS)


def to_cost_constraint(grp):
    """oracle ConstraintGroup -> product CostConstraint holder + weight."""
    return CostConstraint(None, 64, q_l=[q for q in grp.q], traj_range_l=grp.t_range.numpy().tolist(),
                          radius_l=grp.radius.numpy().tolist(), is_soft=grp.weight < 0.1), grp.weight


def hip_guide(map_name, cons_per_robot, cutoff=0.05, n_robots=1, robot_env_ids=None):
    g = GuideManagerTrajectoriesWithVelocity(dataset(), env_id=map_name, obstacle_cutoff_margin=cutoff,
                                             n_robots=n_robots, robot_env_ids=robot_env_ids, device="cuda")
    for r, groups in enumerate(cons_per_robot):
        pairs = [to_cost_constraint(grp) for grp in groups]
        g.add_extra_costs([p[0] for p in pairs], [p[1] for p in pairs], robot=r)
    return g


def hip_run_inference(case, xT, steps, weights_seed=0):
    import cases
    model = hip_model(case["T"], weights_seed)
    guide = hip_guide(case["map"], [case["cons"]], case.get("cutoff", 0.05)) if case.get("use_guide", True) else None
    hc = cases.hard_conds_for(case["start"], case["goal"])
    return model.run_inference(None, hc, n_samples=case["B"], horizon=64, return_chain=True, sample_fn=ddpm_sample_fn,
                               guide=guide, n_guide_steps=20, t_start_guide=ceil(0.5 * case["T"]),
                               noise_std_extra_schedule_fn=lambda x: 0.5, n_diffusion_steps_without_noise=1,
                               warm_start_path_b=xT.cuda(), step_noise=steps.cuda())


def teacher_forced_guided_step(test, tag, s, T, starts, goals, map_name, picks, i, seeds, paths_np=None):
    """ONE guided ddpm_sample_fn step (sample_functions.py:40-107: UNet + posterior mean + 20 guide iterations + noise + hard
    conditioning) of MultiRobotSampler `s` on ALL its local trajectories, started from a mid-chain looking state, against the
    oracle on the trajectories `picks` = [(global robot, sample)].  Bound per trajectory: the north-star 1e-3; a trajectory
    beyond it must be a BRANCH FLIP -- the kernel's decision trace differs from the oracle's at some guide iteration and the
    oracle run on the kernel's decisions agrees with the kernel up to fp32 rounding (GuidedStepJudge); anything else fails.  paths_np: all robots' paths for the inter-robot
    soft constraints (None: no inter-robot term).  Returns the worst error / bound ratio."""
    import cases
    import parity_log
    from oracle import mmd_oracle as O
    H, D, B = 64, 4, s.n_samples
    n = s.n_local * B
    model = s.model
    x = torch.from_numpy(synth.synth_noise(seeds[0], (n, H, D))) * 0.5
    x[:, 0] = s.hard_conds[0].cpu().repeat_interleave(B, 0)
    x[:, -1] = s.hard_conds[H - 1].cpu().repeat_interleave(B, 0)
    noise = torch.from_numpy(synth.synth_noise(seeds[1], (n, H, D)))
    ys = x.clone().cuda()
    model.sample_step(ys, s.hard_conds, i, guide=s.guide, n_guide_steps=20, t_start_guide=ceil(0.5 * T),
                      noise_std_extra_schedule_fn=lambda t: 0.5, n_robots=s.n_local, noise=noise.cuda())
    ys = ys.cpu()
    assert torch.isfinite(ys).all()
    sd = O.state_dict_to_torch(synth.synth_unet_state_dict(0))
    tb = O.schedule_tables(T)
    gp = cases.guide_params(map_name)
    judge = GuidedStepJudge(model, s.guide, x, s.hard_conds, i, ceil(0.5 * T), s.n_local, noise, ys)
    worst = 0.0
    for r, b in picks:
        idx = (r - s.robot0) * B + b
        groups = [cases.soft_group(paths_np, r)] if paths_np is not None else []
        hc = cases.hard_conds_for(starts[r], goals[r])
        verdict, err = judge.check(test, f"{tag}_robot{r}_sample{b}", idx, sd, tb, gp, groups, hc, 1000 + idx)
        if verdict == "within":
            worst = max(worst, err / 1e-3)
    return worst


# ---- branch-flip attribution of a guided step (include/mmd_amd_debug.h: mmd_debug_ddpm_step_trace) -----------
def hip_step_with_trace(model, x, hard_conds, i, guide, t_start_guide, n_robots=1, noise=None, n_guide_steps=20,
                        noise_fn=lambda t: 0.5):
    """model.sample_step through the measurement hook: returns (y, mu, gchain [n_it, n, H, 4], trace [n_it, n, H, 12] int64), all
    on the CPU; x is left untouched."""
    import ctypes as C
    from mmd_amd import _lib
    y = x.clone().cuda().contiguous()
    n, H, D = y.shape
    hard, mask = model._hard_tensor(hard_conds, n_robots, H, y.device, D)
    s = model._sampler_desc(n_guide_steps, t_start_guide, noise_fn, mask)
    gd = guide.desc()
    ws = model.model.workspace(n, y.device, sampler=True)
    mu = torch.empty_like(y)
    gchain = torch.empty((n_guide_steps, n, H, D), dtype=torch.float32, device=y.device)
    trace = torch.zeros((n_guide_steps, n, H, _lib.TRACE_WORDS), dtype=torch.int32, device=y.device)
    nz = noise.cuda().contiguous() if noise is not None else None
    _lib.launch("mmd_debug_ddpm_step_trace", y, model.model.handle(model.n_diffusion_steps, y.device), C.byref(s), C.byref(gd),
                y.data_ptr(), hard.data_ptr(), n_robots, n // n_robots, int(i), nz.data_ptr() if nz is not None else None,
                C.c_uint64(0), C.c_uint32(int(i) & 0xFFFFFFFF), ws.data_ptr(), ws.numel(), mu.data_ptr(), gchain.data_ptr(),
                trace.data_ptr())
    tr = trace.cpu().to(torch.int64) & 0xFFFFFFFF
    return y.cpu(), mu.cpu(), gchain.cpu(), tr


def decode_trace(tr, slots):
    """tr [n_it, H, 12] (one trajectory) -> one oracle GuideSets per iteration; slots = slots per constraint group."""
    from oracle import mmd_oracle as O
    out = []
    for w in tr:
        f = w[:, 1]
        bit = lambda k: ((f >> k) & 1).bool()                                                # noqa: E731
        cons = []
        for g, S in enumerate(slots):
            assert g < 4 and S <= 64, "the trace holds 4 groups x 64 slots"
            lo, hi = w[:, 2 + 2 * g], w[:, 3 + 2 * g]
            sidx = torch.arange(S)
            word = torch.where(sidx[None, :] < 32, lo[:, None], hi[:, None])
            cons.append((((word >> (sidx % 32)[None, :]) & 1).bool())[None])                 # [1,H,S]
        clip = torch.stack([bit(8), bit(9), bit(10)] + [bit(11 + g) for g in range(len(slots))], dim=-1)[None]
        out.append(O.GuideSets(cell=w[None, :, 0], obj_active=bit(0)[None], obj_win=((f >> 1) & 7)[None], ws_active=bit(4)[None],
                               ws_arg=((f >> 5) & 3)[None], clip=clip,
                               unnorm_clip=torch.stack([bit(16 + d) for d in range(4)], dim=-1)[None], cons=cons))
    return out


def first_set_difference(a, b, H=64):
    """(kind, support point, detail) of the first DISCRETE decision in which two GuideSets differ on an interior support point
    (rows 0 / H-1 carry no gradient; the continuous switches -- clips -- are not flips), or None."""
    inner = slice(1, H - 1)
    checks = [("obj_active", a.obj_active, b.obj_active), ("ws_active", a.ws_active, b.ws_active)]
    both = a.obj_active & b.obj_active
    checks.append(("sdf_cell", torch.where(both, a.cell, 0), torch.where(both, b.cell, 0)))
    checks.append(("obj_field", torch.where(both, a.obj_win, 0), torch.where(both, b.obj_win, 0)))
    bw = a.ws_active & b.ws_active
    checks.append(("ws_arg", torch.where(bw, a.ws_arg, 0), torch.where(bw, b.ws_arg, 0)))
    for g, (ma, mb) in enumerate(zip(a.cons, b.cons)):
        S = min(ma.shape[-1], mb.shape[-1])
        checks.append((f"constraint_group{g}_active_set", ma[..., :S], mb[..., :S]))
    for kind, u, v in checks:
        d = (u != v)[0, inner]
        if d.any():
            t = int(torch.nonzero(d.reshape(d.shape[0], -1).any(-1))[0]) + 1
            return kind, t, int(d.sum())
    return None


def attribute_guided_step(y_hip, mu_hip, gstate_hip, hip_sets, xi, nz, i, tsg, sd, tb, gp, groups, hc):
    """One trajectory's guided step, kernel against oracle, taken apart.  y_hip / mu_hip / gstate_hip [1,H,D]: the kernel's result,
    the posterior mean its guide iterations start from and its state after the last iteration; hip_sets: its decoded decisions.
      err     kernel vs the oracle (its own decisions)
      first   (iteration, kind, support point, count) of the first discrete decision in which the two traces differ, or None
      ferr    kernel vs the oracle run on the KERNEL's decisions: what is left is arithmetic
      d_o32 / d_hip   fp32 rounding along the kernel's decision path, amplified by the 20 norm-clipped steps: the oracle's guide
              iterations in float32, and the kernel's, against the same iterations in float64 -- all three from the kernel's
              posterior mean, decisions frozen.  The yardstick for `ferr`: two fp32 evaluations of one smooth map cannot agree
              better than each agrees with exact arithmetic."""
    import cases
    from oracle import mmd_oracle as O
    own_sets = []

    def own_guide(z):
        own_sets.append(O.guide_decisions(z, gp, groups, clip_mode="always"))
        return O.guide_grad(z, gp, groups, clip_mode="always")

    def step(guide):
        return O.apply_hard_conditioning(
            O.ddpm_sample_step(sd, tb, xi.clone(), hc, i, guide=guide, n_guide_steps=20, t_start_guide=tsg, noise=nz,
                               noise_std_extra=0.5), hc)
    ref = step(own_guide)
    first = None
    for k, (a, b) in enumerate(zip(hip_sets, own_sets)):
        d = first_set_difference(a, b)
        if d is not None:
            first = (k,) + d
            break
    it = iter(hip_sets)
    forced = step(lambda z: O.guide_grad_forced(z, gp, groups, next(it)))
    hc64 = {k: v.double() for k, v in hc.items()}
    z32, z64 = mu_hip.clone(), mu_hip.double()
    for sets in hip_sets:
        z32 = O.apply_hard_conditioning(z32 + O.guide_grad_forced(z32, gp, groups, sets), hc)
        z64 = O.apply_hard_conditioning(z64 + O.guide_grad_forced(z64, gp, groups, sets), hc64)
    return dict(err=cases.rel_l2(y_hip, ref), ferr=cases.rel_l2(y_hip, forced), first=first, d_o32=cases.rel_l2(z32, z64),
                d_hip=cases.rel_l2(gstate_hip, z64), ref=ref)


def rounding_bound(a):
    """What two fp32 evaluations of the same decision path may differ by: 3 x (their two distances from exact arithmetic), at least
    1e-4 and never more than half the north-star tolerance."""
    return min(5e-4, max(1e-4, 3.0 * (a["d_hip"] + a["d_o32"])))


class GuidedStepJudge:
    """Verdict on ONE teacher-forced guided step of a HIP batch against the oracle, per trajectory:
      err < 1e-3 (the north-star tolerance)                                   -> 'within'
      else the kernel's own decision trace is read back (bit-identical re-run of the step through the measurement hook,
      include/mmd_amd_debug.h) and the step is taken apart (attribute_guided_step):
        the traces differ at some guide iteration and the oracle on the KERNEL's decisions agrees with the kernel up to fp32
        rounding along that path                                              -> 'flip@<iteration>:<kind>:t<support point>'
        the traces are identical and the difference is within the fp32 rounding of the two evaluations (the 20 norm-clipped
        steps amplify rounding 5 .. 350 x with every decision frozen)          -> 'fp32'
      else                                                                     -> AssertionError."""

    def __init__(self, model, guide, x, hard_conds, i, t_start_guide, n_robots, noise, y_hip):
        self.model, self.guide, self.x, self.hard_conds, self.i = model, guide, x, hard_conds, i
        self.tsg, self.n_robots, self.noise, self.y_hip = t_start_guide, n_robots, noise, y_hip
        self._tr = None

    def trace(self):
        if self._tr is None:
            y, mu, gchain, tr = hip_step_with_trace(self.model, self.x, self.hard_conds, self.i, self.guide, self.tsg,
                                                    self.n_robots, self.noise)
            assert torch.equal(y, self.y_hip), "the trace instantiation must reproduce the production step bit for bit"
            self._tr = (mu, gchain, tr)
        return self._tr

    def check(self, test, tag, idx, sd, tb, gp, groups, hc, sens_seed, pert=2e-6, n_sens=8, lin=1.0):
        import cases
        import parity_log
        from oracle import mmd_oracle as O
        xi, nz, i = self.x[idx:idx + 1], self.noise[idx:idx + 1], self.i

        def step(p=None):
            return O.apply_hard_conditioning(
                O.ddpm_sample_step(sd, tb, xi.clone(), hc, i, guide=lambda z: O.guide_grad(z, gp, groups, clip_mode="always"),
                                   n_guide_steps=20, t_start_guide=self.tsg, noise=nz, noise_std_extra=0.5, eps_rel_perturb=p), hc)
        ref = step()
        err = cases.rel_l2(self.y_hip[idx:idx + 1], ref)
        if err < 1e-3:
            parity_log.record(test, tag, i, err, bound=1e-3)
            return "within", err
        # ---- over the tolerance: whose decisions differ, and is that all that differs?
        mu, gchain, tr = self.trace()
        slots = [O.slot_table(g).shape[0] for g in groups]
        a = attribute_guided_step(self.y_hip[idx:idx + 1], mu[idx:idx + 1], gchain[-1, idx:idx + 1], decode_trace(tr[:, idx], slots),
                                  xi, nz, i, self.tsg, sd, tb, gp, groups, hc)
        extra = dict(forced_err=a["ferr"], fp32_rounding_oracle=a["d_o32"], fp32_rounding_kernel=a["d_hip"], first_difference=str(a["first"]))
        if a["first"] is not None and a["ferr"] < rounding_bound(a):
            verdict = f"flip@{a['first'][0]}:{a['first'][1]}:t{a['first'][2]}"
            parity_log.record(test, tag, i, err, bound=1e-3, flip=verdict, **extra)
            return verdict, err
        if a["first"] is None and err < rounding_bound(a):
            parity_log.record(test, tag, i, err, bound=rounding_bound(a), fp32="identical decisions: fp32 rounding of both evaluations", **extra)
            return "fp32", err
        # neither: a step beyond the tolerance that is not an attributed flip and not fp32 rounding along identical decisions is a
        # parity FAILURE (the sensitivity yardstick of rounds 2-5 is gone)
        parity_log.record(test, tag, i, err, bound=1e-3, unexplained=True, **extra)
        raise AssertionError((test, tag, "guided step beyond 1e-3 without an attributed flip / fp32 verdict", err, a["ferr"], a["first"],
                              a["d_hip"], a["d_o32"]))
