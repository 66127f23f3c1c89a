"""-m gpu: the HIP path (through the C ABI of libmmd_amd.so) against the oracle and the reference's golden vectors.

Tolerance (BASELINE.json north_star): final trajectories within 1e-3 relative L2 of the reference sampler (fp32)."""
import os
from math import ceil

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mmd_amd import _lib, synth      # noqa: E402
from oracle import mmd_oracle as O   # noqa: E402
import cases                         # noqa: E402
import parity_log                    # noqa: E402
from cases import GOLDEN, H, D, rel_l2, chaos_bounds   # noqa: E402

TOL_FINAL = 1e-3          # BASELINE.json north_star: within 1e-3 relative L2 of the reference sampler
TOL_STEP_GUIDED = 1e-3    # one teacher-forced guided DDPM step (20 norm-clipped guide iterations); measured <= 3e-4
TOL_STEP_PLAIN = 2e-5     # one teacher-forced unguided step
SENS_FACTOR = 1.5         # end-to-end rows of a chaotic (guided) chain: err < max(TOL_FINAL, SENS_FACTOR * lin * sens)


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from mmd_amd import _lib
    _lib.load()


def _gc():
    import gpu_common
    return gpu_common


@pytest.mark.parametrize("n", [1, 4, 13, 64])
def test_unet_forward_vs_oracle(n):
    model = _gc().hip_model(100)
    sd = O.state_dict_to_torch(synth.synth_unet_state_dict(0))
    x = torch.from_numpy(synth.synth_noise(40 + n, (n, H, D)))
    for t in (0, 37, 99):
        ref = O.unet_forward(sd, x, torch.full((n,), t, dtype=torch.long))
        out = model.model(x.cuda(), t).cpu()
        assert torch.isfinite(out).all()
        assert rel_l2(out, ref) < 2e-5, (n, t, rel_l2(out, ref))


def test_unet_forward_per_sample_timesteps():
    """TemporalUnet.forward(x, t[B]) with DIFFERENT timesteps per sample (temporal_unet.py:121; the training loss draws them per sample;
    VERDICT r5 missing #4: rounds 1-5 silently used time[0]): against the oracle's per-sample forward, bitwise the per-timestep
    launches, and a uniform tensor / an int stay one launch with the same bits; a time tensor of the wrong length is refused."""
    model = _gc().hip_model(100)
    sd = O.state_dict_to_torch(synth.synth_unet_state_dict(0))
    n = 11
    x = torch.from_numpy(synth.synth_noise(61, (n, H, D)))
    t = torch.tensor([3, 99, 3, 0, 37, 99, 3, 50, 0, 37, 12])
    out = model.model(x.cuda(), t.cuda()).cpu()
    ref = O.unet_forward(sd, x, t)
    assert torch.isfinite(out).all() and rel_l2(out, ref) < 2e-5, rel_l2(out, ref)
    for tv in t.unique().tolist():
        rows = torch.nonzero(t == tv).reshape(-1)
        assert torch.equal(out[rows], model.model(x[rows].contiguous().cuda(), tv).cpu()), tv
    u = model.model(x.cuda(), torch.full((n,), 37)).cpu()
    assert torch.equal(u, model.model(x.cuda(), 37).cpu()) and torch.equal(u, model.model(x.cuda(), torch.tensor(37)).cpu())
    with pytest.raises(ValueError):
        model.model(x.cuda(), torch.tensor([1, 2, 3]))


@pytest.mark.parametrize("scale", [1e-3, 8.0])
def test_unet_forward_input_range(scale):
    """The convs run as fp16 two-piece splits under dynamic per-sample input scales: tiny and large inputs (x_T draws reach
    |x| ~ 4, the clamp keeps the rest in [-1, 1]) stay at fp32-grade agreement with the oracle."""
    model = _gc().hip_model(100)
    sd = O.state_dict_to_torch(synth.synth_unet_state_dict(0))
    x = torch.from_numpy(synth.synth_noise(77, (8, H, D))) * scale
    ref = O.unet_forward(sd, x, torch.full((8,), 63, dtype=torch.long))
    out = model.model(x.cuda(), 63).cpu()
    assert torch.isfinite(out).all() and rel_l2(out, ref) < 2e-5, rel_l2(out, ref)


def test_unet_forward_batch_independence():
    """A trajectory's eps does not depend on which workgroup / wave slot it lands in, nor on which of the three kernels runs it:
    2051 trajectories (unet_kernel<4>: 513 workgroups of four, the last one ragged) against the same rows evaluated in small
    batches (unet_kernel<1>, one trajectory per workgroup, launches of up to 256; unet_kernel<2>, two per workgroup, up to 512)
    -- bit-identical."""
    model = _gc().hip_model(100)
    x = torch.from_numpy(synth.synth_noise(91, (2051, H, D))).cuda()
    big = model.model(x, 17)
    for rows in ([0, 1, 2, 3], [5, 1030, 2046], [2047, 2048, 2049, 2050], [2050]):
        small = model.model(x[rows].contiguous(), 17)
        assert torch.equal(big[rows], small), rows
    # either side of the launch-size thresholds between the kernels (256, 512), full and ragged last workgroups
    for n in (255, 256, 257, 258, 300, 511, 512, 513, 514):
        assert torch.equal(model.model(x[:n].contiguous(), 17), big[:n]), n


def test_unet_forward_is_batch_independent():
    """A trajectory's eps does not depend on which launch / workgroup it is in (the property the sharded sampler rests on):
    the forward of a sub-batch, of a ragged batch and of a large batch agree bit for bit row by row (3- and 64-row batches run
    unet_kernel<1>, the 401-row one unet_kernel<2>, the 1026- and 2050-row ones unet_kernel<4>), and meet the oracle bound."""
    model = _gc().hip_model(100)
    sd = O.state_dict_to_torch(synth.synth_unet_state_dict(0))
    x = torch.from_numpy(synth.synth_noise(300, (2050, H, D))).cuda()
    full = model.model(x, 41)
    assert torch.isfinite(full).all()
    for lo, n in ((0, 3), (5, 64), (700, 401), (1024, 1026), (2047, 3)):
        part = model.model(x[lo:lo + n].contiguous(), 41)
        assert torch.equal(part, full[lo:lo + n]), (lo, n)
    ref = O.unet_forward(sd, x[:64].cpu(), torch.full((64,), 41, dtype=torch.long))
    assert rel_l2(full[:64].cpu(), ref) < 2e-5


def test_unet_forward_accuracy_against_fp64():
    """The kernel computes fp32 arithmetic as a two-piece fp16 split on the fp16 matrix pipe (f16x2): against the
    oracle run in float64 its error is of the size of the fp32 reference's own rounding error (measured ratio <= 1.1, bound 4).
    Two weight sets, inputs scaled by 1e-3 .. 8, and a weight set whose conv kernels span five orders of magnitude (every piece
    of the split carries signal)."""
    from mmd_amd.diffusion_model import GaussianDiffusionModel
    from mmd_amd.temporal_unet import TemporalUnet
    t = torch.full((16,), 41, dtype=torch.long)
    wide = synth.synth_unet_state_dict(2)
    rng = np.random.Generator(np.random.PCG64(9))
    for k, v in wide.items():
        if k.endswith(".block.0.weight"):                     # the k=5 convs
            wide[k] = (v * np.exp(rng.uniform(-6.0, 2.0, size=v.shape))).astype(np.float32)
    for name, sd_np in (("seed0", synth.synth_unet_state_dict(0)), ("seed1", synth.synth_unet_state_dict(1)), ("wide", wide)):
        unet = TemporalUnet(state_dim=4, n_support_points=64, unet_input_dim=32, dim_mults=(1, 2, 4))
        unet.load_state_dict(sd_np)
        model = GaussianDiffusionModel(model=unet, variance_schedule="exponential", n_diffusion_steps=100, predict_epsilon=True)
        sd = O.state_dict_to_torch(sd_np)
        sd64 = {k: v.double() for k, v in sd.items()}
        for scale in (1.0, 1e-3, 8.0):
            x = torch.from_numpy(synth.synth_noise(300, (16, H, D))) * scale
            ref64 = O.unet_forward(sd64, x.double(), t)
            err_ref32 = float((O.unet_forward(sd, x, t).double() - ref64).norm() / ref64.norm())
            err_hip = float((model.model(x.cuda(), 41).cpu().double() - ref64).norm() / ref64.norm())
            parity_log.record("unet_forward_vs_fp64", f"{name} scale={scale:g}", None, err_hip, sens=err_ref32,
                              bound=max(4e-6, 4 * err_ref32), note="sens = the fp32 reference forward against the same float64 forward")
            assert err_hip < max(4e-6, 4 * err_ref32), (name, scale, err_hip, err_ref32)


def test_unet_forward_golden():
    g = np.load(os.path.join(GOLDEN, "g2_unet.npz"))
    model = _gc().hip_model(100)
    x = torch.from_numpy(synth.synth_noise(int(g["x_seed"]), (4, H, D))).cuda()
    for t in g["ts"]:
        out = model.model(x, torch.full((4,), int(t), dtype=torch.long, device="cuda")).cpu()
        assert rel_l2(out, g[f"eps_t{t}"]) < 2e-5


def test_guide_single_eval_golden():
    g = np.load(os.path.join(GOLDEN, "g5_guide.npz"))
    _, _, soft, hard = cases.highways_case()
    guide = _gc().hip_guide("EnvHighways2D", [[soft, hard]])
    x = (torch.from_numpy(synth.synth_noise(7, (8, H, D))) * 0.6).cuda()
    assert float((guide(x).cpu() - torch.from_numpy(g["highways_B8"])).abs().max()) < 2e-6
    x2 = (torch.from_numpy(synth.synth_noise(8, (8, H, D))) * 1.1).cuda()
    assert float((guide(x2).cpu() - torch.from_numpy(g["highways_B8_wide"])).abs().max()) < 2e-6
    starts, goals = synth.start_goal_circle(32, 0.8)
    grp = cases.soft_group(synth.straight_line_paths(starts, goals, H), 0)
    guide = _gc().hip_guide("EnvEmpty2D", [[grp]])
    x3 = (torch.from_numpy(synth.synth_noise(9, (4, H, D))) * 0.5).cuda()
    assert float((guide(x3).cpu() - torch.from_numpy(g["empty32_B4"])).abs().max()) < 2e-6


def test_reference_options_golden():
    """Options of the reference that MPD / MPDEnsemble leave at their defaults, against the reference itself (g16): gradient
    clipping by value and switched off (guides.py:228-259), scale_grad_by_std (sample_functions.py:100-101: the guide gradient
    times the posterior variance -- a well-conditioned chain, so every row is held to the north-star tolerance) and
    GaussianDiffusionModel(predict_epsilon=False) (diffusion_model_base.py:131-141)."""
    from mmd_amd.diffusion_model import GaussianDiffusionModel, ddpm_sample_fn
    from mmd_amd.temporal_unet import TemporalUnet
    g = np.load(os.path.join(GOLDEN, "g16_options.npz"))
    T, B, s_x, s_n = (int(v) for v in g["meta"])
    starts, goals, soft, hard = cases.highways_case()
    x = (torch.from_numpy(synth.synth_noise(7, (8, H, D))) * 0.6).cuda()
    gc = _gc()
    pairs = [gc.to_cost_constraint(grp) for grp in (soft, hard)]

    def guide_with(**kw):
        from mmd_amd.guides import GuideManagerTrajectoriesWithVelocity
        gd = GuideManagerTrajectoriesWithVelocity(gc.dataset(), env_id="EnvHighways2D", device="cuda", **kw)
        gd.add_extra_costs([p[0] for p in pairs], [p[1] for p in pairs])
        return gd
    e_val = float((guide_with(clip_grad=True, clip_grad_rule="value", max_grad_value=float(g["max_grad_value"]))(x).cpu()
                   - torch.from_numpy(g["guide_clip_value"])).abs().max())
    ref_off = torch.from_numpy(g["guide_clip_off"])
    e_off = float((guide_with(clip_grad=False)(x).cpu() - ref_off).abs().max() / ref_off.abs().max())
    parity_log.record("reference_options", "guide_clip_by_value_maxabs", None, e_val, bound=2e-6)
    parity_log.record("reference_options", "guide_no_clip_rel_maxabs", None, e_off, bound=2e-6)
    assert e_val < 2e-6 and e_off < 2e-6, (e_val, e_off)
    xT = torch.from_numpy(synth.synth_noise(s_x, (B, H, D)))
    steps = torch.from_numpy(synth.synth_noise(s_n, (T + 1, B, H, D)))
    hc = cases.hard_conds_for(starts[3], goals[3])
    model = gc.hip_model(T)
    chain = model.run_inference(None, hc, n_samples=B, horizon=H, return_chain=True, sample_fn=ddpm_sample_fn,
                                guide=guide_with(), n_guide_steps=20, t_start_guide=ceil(0.5 * T),
                                noise_std_extra_schedule_fn=lambda t: 0.5, n_diffusion_steps_without_noise=1,
                                warm_start_path_b=xT.cuda(), step_noise=steps.cuda(), scale_grad_by_std=True).cpu()
    ref = torch.from_numpy(g["chain_scale_grad_by_std"])
    e_std = max(rel_l2(chain[k], ref[k]) for k in range(T + 2))
    parity_log.record("reference_options", "chain_scale_grad_by_std", None, e_std, bound=TOL_FINAL)
    assert e_std < TOL_FINAL, e_std
    unet = TemporalUnet(state_dim=4, n_support_points=64, unet_input_dim=32, dim_mults=(1, 2, 4))
    unet.load_state_dict(synth.synth_unet_state_dict(0))
    m0 = GaussianDiffusionModel(model=unet, variance_schedule="exponential", n_diffusion_steps=T, predict_epsilon=False)
    chain0 = m0.run_inference(None, hc, n_samples=B, horizon=H, return_chain=True, sample_fn=ddpm_sample_fn, guide=None,
                              noise_std_extra_schedule_fn=lambda t: 0.5, n_diffusion_steps_without_noise=1,
                              warm_start_path_b=xT.cuda(), step_noise=steps.cuda()).cpu()
    ref0 = torch.from_numpy(g["chain_predict_x0"])
    e_x0 = max(rel_l2(chain0[k], ref0[k]) for k in range(T + 2))
    parity_log.record("reference_options", "chain_predict_x0", None, e_x0, bound=TOL_FINAL)
    assert e_x0 < TOL_FINAL, e_x0


def _raw_guide_grad(guide, x, patch):
    """guide(x) through the C ABI with a patched descriptor (isolates one cost term)."""
    import ctypes as C
    from mmd_amd import _lib
    d = guide.desc()
    patch(d)
    y = x.contiguous().clone()
    hard = torch.zeros(1, 2, D, device="cuda")
    _lib.check(_lib.load().mmd_guide_steps(C.byref(d), y.data_ptr(), hard.data_ptr(), 0, 1, x.shape[0], 1, None,
                                           _lib.current_stream_ptr()))
    return (y - x).cpu()


def test_guide_per_term_golden_g4():
    """Every cost term of the guide on its own against the reference's clipped per-term gradients (g4): fixed-object SDF,
    workspace boundaries, GP prior, soft and hard constraint groups -- each isolated by zeroing the other weights /
    moving the workspace walls out of reach, so an error in one term cannot hide behind another's clip or weight."""
    g = np.load(os.path.join(GOLDEN, "g4_guide_terms.npz"))
    _, _, soft, hard = cases.highways_case()
    x = (torch.from_numpy(synth.synth_noise(int(g["x_seed"]), (8, H, D))) * float(g["x_scale"])).cuda()
    gc = _gc()
    far = 1e6

    def only_obj(d):
        d.weight_collision, d.weight_smoothness = 1.0, 0.0
        d.ws_min[:], d.ws_max[:] = [-far, -far], [far, far]

    def only_ws(d):
        d.weight_collision, d.weight_smoothness, d.n_grids = 1.0, 0.0, 0

    def only_gp(d):
        d.weight_collision, d.weight_smoothness = 0.0, 1.0

    def no_base(d):
        d.weight_collision, d.weight_smoothness = 0.0, 0.0

    plain = gc.hip_guide("EnvHighways2D", [[]])
    checks = [("obj", plain, only_obj), ("ws", plain, only_ws), ("gp", plain, only_gp)]
    soft1 = type(soft)(q=soft.q, t_range=soft.t_range, radius=soft.radius, weight=1.0)
    hard0 = type(hard)(q=hard.q, t_range=hard.t_range, radius=hard.radius, weight=0.0)
    hard1 = type(hard)(q=hard.q, t_range=hard.t_range, radius=hard.radius, weight=1.0)
    checks.append(("cons_soft", gc.hip_guide("EnvHighways2D", [[soft1, hard0]]), no_base))
    checks.append(("cons_hard", gc.hip_guide("EnvHighways2D", [[hard1]]), no_base))
    for name, guide, patch in checks:
        got = -_raw_guide_grad(guide, x, patch)
        err = float((got - torch.from_numpy(g[f"term_{name}"])).abs().max())
        parity_log.record("guide_per_term_g4", name, None, err, bound=2e-6, note="max abs")
        assert err < 2e-6, (name, err)
        assert float(torch.from_numpy(g[f"term_{name}"]).abs().max()) > 1e-3, name      # the term is exercised


def test_guide_and_postprocess_with_two_sdf_grids():
    """n_grids > 1 (a map's fixed-object grid + an extra ObjectField grid, mpd.py:215-233 / env_base.py:76-89): the guide
    takes max_k relu(margin - sdf_k) with the arg-max grid's gradient, the occupancy check ORs the grids -- both kernels'
    k >= 1 loops against the oracle with the Highways and the Conveyor grids stacked as two objects of one map."""
    import ctypes as C
    from mmd_amd import _lib, postprocess as post
    from mmd_amd.environments import sdf_grid_texture
    tex = torch.from_numpy(np.stack([sdf_grid_texture("EnvHighways2D"), sdf_grid_texture("EnvConveyor2D")])[None]).cuda().contiguous()
    guide = _gc().hip_guide("EnvHighways2D", [[]])
    gp = cases.guide_params("EnvHighways2D")
    gp.sdf_grids = [cases.sdf_grid("EnvHighways2D"), cases.sdf_grid("EnvConveyor2D")]
    x = (torch.from_numpy(synth.synth_noise(60, (8, H, D))) * 0.6)

    def two_grids(d):
        d.n_grids, d.n_maps, d.sdf_grids_dev = 2, 1, tex.data_ptr()

    got = _raw_guide_grad(guide, x.cuda(), two_grids)
    ref = O.guide_grad(x, gp, [], clip_mode="always")
    assert float((got - ref).abs().max()) < 2e-6
    one = _raw_guide_grad(guide, x.cuda(), lambda d: None)
    assert float((got - one).abs().max()) > 1e-3                         # the second grid matters on this batch
    # occupancy over both grids
    d = guide.desc()
    two_grids(d)
    pts = torch.from_numpy(np.random.Generator(np.random.PCG64(61)).uniform(-1, 1, size=(4096, 2)).astype(np.float32))
    occ = post.compute_collision(pts.cuda(), d, margin=0.05).cpu()
    assert torch.equal(occ, O.compute_collision(pts, gp, 0.05)) and occ.any() and not occ.all()


def test_guide_point_on_constraint_centre_is_finite():
    """A support point that coincides exactly with a constraint centre: torch.norm's backward gives a zero gradient there;
    the kernel must not produce 0 * inf = NaN."""
    x = torch.zeros(4, H, D)
    x[..., 0] = torch.linspace(-0.5, 0.5, H)[None]
    gp = cases.guide_params("EnvEmpty2D")
    xu = O.unnormalize(x, gp.norm_mins, gp.norm_maxs, clip_mode="always")
    centre = [float(xu[0, 20, 0]), float(xu[0, 20, 1])]               # exactly where support point 20 sits
    grp = cases.hard_group([[0.0, 0.0], centre], [[30, 34], [20, 21]])
    guide = _gc().hip_guide("EnvEmpty2D", [[grp]])
    out = guide(x.cuda()).cpu()
    assert torch.isfinite(out).all()
    ref = O.guide_grad(x, gp, [grp], clip_mode="always")
    ref_autograd = O.guide_grad_dense_autograd(x, gp, [grp])          # torch.norm backward: zero sub-gradient at d = 0
    assert torch.isfinite(ref_autograd).all() and float((ref - ref_autograd).abs().max()) < 2e-6
    assert float((out - ref).abs().max()) < 2e-6


def test_guide_steps_chain_output():
    """mmd_guide_steps with a chain buffer (the post-diffusion guide steps of 'diffusion_prior_then_guide' in ONE launch)
    == n single-step launches."""
    starts, goals, soft, hard = cases.highways_case()
    hc = cases.hard_conds_for(starts[3], goals[3])
    guide = _gc().hip_guide("EnvHighways2D", [[soft, hard]])
    x = O.apply_hard_conditioning(torch.from_numpy(synth.synth_noise(53, (8, H, D))) * 0.5, hc).cuda()
    hardt = torch.stack([hc[0], hc[H - 1]])[None].cuda().contiguous()
    y = x.clone()
    chain = torch.empty((7, 8, H, D), device="cuda")
    guide.guide_steps(y, hardt, _lib.HARD_ROWS_START_GOAL, 7, chain=chain)
    z = x.clone()
    for k in range(7):
        guide.guide_steps(z, hardt, _lib.HARD_ROWS_START_GOAL, 1)
        assert torch.equal(chain[k], z), k
    assert torch.equal(y, z)


def test_q_sample_ensemble_seed_covers_every_tile():
    """DiffusionsEnsemble.run_local_inference noises a [B, K*64, 4] seed with models[0].q_sample
    (diffusion_ensemble.py:279-281): every point of every tile must be noised (the C ABI counts blocks of 64 points)."""
    model = _gc().hip_model(25)
    B, K = 4, 2
    x0 = torch.from_numpy(synth.synth_noise(54, (B, K * H, D))) * 0.4
    noise = torch.from_numpy(synth.synth_noise(55, (B, K * H, D)))
    out = model.q_sample(x0.cuda(), 3, noise=noise.cuda()).cpu()
    ref = O.q_sample(O.schedule_tables(25), x0, 3, noise)
    assert out.shape == ref.shape and rel_l2(out, ref) < 1e-6
    # Philox path: every tile gets noise, reproducible per global index
    a = model.q_sample(x0.cuda(), 3).cpu()
    assert torch.isfinite(a).all() and float((a[:, H:] - x0[:, H:]).abs().mean()) > 1e-3
    with pytest.raises(ValueError):
        model.q_sample(x0[:, :100].cuda(), 3)


def test_noise_std_extra_schedule_is_evaluated_per_step():
    """noise_std_extra_schedule_fn(t) is called with t on every step (sample_functions.py:83-86): a non-constant schedule
    must reach the kernel per step, checked against the oracle on the unguided prior."""
    from mmd_amd.diffusion_model import ddpm_sample_fn
    T, B = 25, 4
    model = _gc().hip_model(T)
    starts, goals = synth.start_goal_circle(6, 0.8)
    hc = cases.hard_conds_for(starts[0], goals[0])
    xT = torch.from_numpy(synth.synth_noise(56, (B, H, D)))
    steps = torch.from_numpy(synth.synth_noise(57, (T + 1, B, H, D)))
    fn = lambda t: 0.25 + 0.03 * float(t)          # noqa: E731
    out = model.run_inference(None, hc, n_samples=B, horizon=H, return_chain=False, sample_fn=ddpm_sample_fn, guide=None,
                              noise_std_extra_schedule_fn=fn, n_diffusion_steps_without_noise=1,
                              warm_start_path_b=xT.cuda(), step_noise=steps.cuda()).cpu()
    sd = O.state_dict_to_torch(synth.synth_unet_state_dict(0))
    tb = O.schedule_tables(T)
    x = O.apply_hard_conditioning(xT.clone(), hc)
    for k, i in enumerate(reversed(range(-1, T))):
        x = O.ddpm_sample_step(sd, tb, x, hc, i, guide=None, noise=steps[k], noise_std_extra=fn(max(i, 0)))
        x = O.apply_hard_conditioning(x, hc)
    assert rel_l2(out, x) < 1e-4, rel_l2(out, x)


def test_workspace_is_exact_for_every_stream_count():
    """the chunked loop must stay inside mmd_sampler_workspace_bytes.  The workspace is allocated with
    exactly that many bytes inside a guarded buffer; the guard words survive n_streams = 1, 2, 3."""
    import ctypes as C
    from mmd_amd import _lib
    from mmd_amd.diffusion_model import ddpm_sample_fn   # noqa: F401
    lib = _lib.load()
    T, B, R = 25, 8, 5
    model = _gc().hip_model(T)
    n = R * B
    handle = model.model.handle(T)
    nbytes = lib.mmd_sampler_workspace_bytes(handle, n)
    assert nbytes == lib.mmd_unet_workspace_bytes(handle, n) + n * H * D * 4
    guard = 4096
    buf = torch.full((nbytes + 2 * guard,), 0xA5, dtype=torch.uint8, device="cuda")
    ws_ptr = buf.data_ptr() + guard
    starts, goals = synth.start_goal_circle(R, 0.8)
    hard = torch.stack([torch.stack([cases.hard_conds_for(starts[r], goals[r])[0],
                                     cases.hard_conds_for(starts[r], goals[r])[H - 1]]) for r in range(R)]).cuda().contiguous()
    for ns in (1, 2, 3):
        s = model._sampler_desc(20, 13, lambda t: 0.5, 3, ns)
        x = torch.empty((n, H, D), device="cuda")
        _lib.check(lib.mmd_p_sample_loop(handle, C.byref(s), None, x.data_ptr(), hard.data_ptr(), R, B, T, 1, 1, None,
                                         C.c_uint64(5), None, C.c_void_p(ws_ptr), nbytes, _lib.current_stream_ptr()))
        torch.cuda.synchronize()
        assert bool((buf[:guard] == 0xA5).all()) and bool((buf[guard + nbytes:] == 0xA5).all()), ns
        assert torch.isfinite(x).all()


def test_guide_20_steps_vs_oracle():
    starts, goals, soft, hard = cases.highways_case()
    gp = cases.guide_params("EnvHighways2D")
    hc = cases.hard_conds_for(starts[3], goals[3])
    x = torch.from_numpy(synth.synth_noise(50, (8, H, D))) * 0.5
    ref = O.apply_hard_conditioning(x.clone(), hc)
    for _ in range(20):
        ref = ref + O.guide_grad(ref, gp, [soft, hard], clip_mode="always")
        ref = O.apply_hard_conditioning(ref, hc)
    guide = _gc().hip_guide("EnvHighways2D", [[soft, hard]])
    y = O.apply_hard_conditioning(x.clone(), hc).cuda()
    hardt = torch.stack([hc[0], hc[H - 1]])[None].cuda().contiguous()
    guide.guide_steps(y, hardt, _lib.HARD_ROWS_START_GOAL, 20)
    assert rel_l2(y.cpu(), ref) < 2e-4   # 20 chained steps: fp32 rounding + nearest-cell / hinge-threshold flips


@pytest.mark.parametrize("n_all", [10, 100])
def test_guide_cooperative_kernel_equals_one_wave_kernel_bitwise(n_all):
    """Launches of <= 512 trajectories run the guided step with FOUR waves per trajectory (ddpm_guide_coop_kernel: wave k sums the
    slots k, k + 4, ... of every constraint group, the partials meet in LDS), larger ones with one wave per trajectory; the slot sum
    is defined as that four-accumulator tree in both, so a robot's rows must not depend on the launch size: 20 guide iterations on
    2 robots x 8 samples (16 trajectories: cooperative) == the same rows inside a 2 x 512 batch (1024: one wave each), bit for
    bit.  Robot 0 carries two groups (soft slots from n_all - 1 paths + a hard vertex group: group boundaries inside the table),
    robot 1 one; n_all = 100: 99 slots, the general (qx, qy, R, R|R|) staging overflows the 60 KiB of LDS into the L2 table.
    ref: cost_functions.py:297-326, guides.py:201-225."""
    starts, goals = synth.start_goal_circle(n_all, 0.8)
    paths = synth.straight_line_paths(starts, goals, H)
    hardg = cases.hard_group([[0.1, 0.2], [-0.2, 0.1]], [[20, 27], [30, 41]])
    cons = [[cases.soft_group(paths, 0), hardg], [cases.soft_group(paths, 1)]]
    hard = torch.stack([torch.stack([cases.hard_conds_for(starts[r], goals[r])[k] for k in (0, H - 1)]) for r in (0, 1)]).cuda().contiguous()
    small = torch.from_numpy(synth.synth_noise(60, (2, 8, H, D))) * 0.5
    big = torch.from_numpy(synth.synth_noise(61, (2, 512, H, D))) * 0.5
    big[:, :8] = small
    g = _gc().hip_guide("EnvHighways2D", cons, n_robots=2)
    ys = small.reshape(16, H, D).clone().cuda()
    yb = big.reshape(1024, H, D).clone().cuda()
    g.guide_steps(ys, hard, _lib.HARD_ROWS_START_GOAL, 20)
    g.guide_steps(yb, hard, _lib.HARD_ROWS_START_GOAL, 20)
    assert torch.isfinite(ys).all()
    assert torch.equal(ys.view(2, 8, H, D), yb.view(2, 512, H, D)[:, :8])
    # ... and against the oracle (robot 0, both groups)
    gp = cases.guide_params("EnvHighways2D")
    hc = cases.hard_conds_for(starts[0], goals[0])
    ref = O.apply_hard_conditioning(small[0].clone(), hc)
    for _ in range(20):
        ref = O.apply_hard_conditioning(ref + O.guide_grad(ref, gp, cons[0], clip_mode="always"), hc)
    start = O.apply_hard_conditioning(small[0].clone(), hc).cuda()
    g1 = _gc().hip_guide("EnvHighways2D", [cons[0]])
    g1.guide_steps(start, hard[:1].contiguous(), _lib.HARD_ROWS_START_GOAL, 20)
    assert rel_l2(start.cpu(), ref) < 2e-4


def test_guide_four_constraint_groups_per_robot():
    """A robot with FOUR constraint groups (soft slots from the other robots, two hard vertex groups with different weights, a second soft
    set) next to a robot with one: the kernels keep the first two groups' bounds and weights in registers for the 20 iterations and read
    the table for the others -- both paths against the oracle, and the cooperative (16 trajectories) and one-wave (2 x 512) kernels
    bitwise against each other.  ref: cost_functions.py:297-326 (one CostConstraint = one term with its own clip and weight)."""
    starts, goals = synth.start_goal_circle(10, 0.45)
    paths = synth.straight_line_paths(starts, goals, H)
    groups0 = [cases.soft_group(paths, 0), cases.hard_group([[0.1, 0.2], [-0.2, 0.1]], [[20, 27], [30, 41]]),
               cases.hard_group([[0.0, 0.0]], [[5, 60]], weight=0.11), cases.soft_group(paths[::-1].copy(), 3, weight=5e-2)]
    cons = [groups0, [cases.soft_group(paths, 1)]]
    hard = torch.stack([torch.stack([cases.hard_conds_for(starts[r], goals[r])[k] for k in (0, H - 1)]) for r in (0, 1)]).cuda().contiguous()
    small = torch.from_numpy(synth.synth_noise(62, (2, 8, H, D))) * 0.5
    big = torch.from_numpy(synth.synth_noise(63, (2, 512, H, D))) * 0.5
    for r in (0, 1):                                          # (the loop pins rows 0 / H - 1 AFTER every iteration: start from pinned states)
        small[r] = O.apply_hard_conditioning(small[r], cases.hard_conds_for(starts[r], goals[r]))
    big[:, :8] = small
    g = _gc().hip_guide("EnvHighways2D", cons, n_robots=2)
    ys, yb = small.reshape(16, H, D).clone().cuda(), big.reshape(1024, H, D).clone().cuda()
    g.guide_steps(ys, hard, _lib.HARD_ROWS_START_GOAL, 20)
    g.guide_steps(yb, hard, _lib.HARD_ROWS_START_GOAL, 20)
    assert torch.isfinite(ys).all() and torch.equal(ys.view(2, 8, H, D), yb.view(2, 512, H, D)[:, :8])
    gp = cases.guide_params("EnvHighways2D")
    for r in (0, 1):
        hc = cases.hard_conds_for(starts[r], goals[r])
        ref = small[r].clone()
        for _ in range(20):
            ref = O.apply_hard_conditioning(ref + O.guide_grad(ref, gp, cons[r], clip_mode="always"), hc)
        assert rel_l2(ys.view(2, 8, H, D)[r].cpu(), ref) < 2e-4, r
    # one evaluation, term by term exact to rounding: every group contributes (dropping any one of robot 0's groups changes the result)
    x0 = small[0].clone()
    full = _gc().hip_guide("EnvHighways2D", [groups0])(x0.cuda()).cpu()
    assert (full - O.guide_grad(x0, gp, groups0, clip_mode="always")).abs().max() < 2e-6
    for k in range(4):
        part = _gc().hip_guide("EnvHighways2D", [groups0[:k] + groups0[k + 1:]])(x0.cuda()).cpu()
        assert not torch.equal(part, full), k


def test_soft_constraints_from_paths_kernel():
    """device-built all-pairs ELL == host-packed ELL for every local robot."""
    from mmd_amd.constraints import soft_constraints_from_paths
    starts, goals = synth.start_goal_circle(6, 0.8)
    paths = synth.straight_line_paths(starts, goals, H)
    ell, gso, gw, rgo, radius = soft_constraints_from_paths(torch.from_numpy(paths).cuda(), 2, 3)
    assert ell.shape == (3 * 5, H, 4) and gso.tolist() == [0, 5, 10, 15] and rgo.tolist() == [0, 1, 2, 3]
    x = (torch.from_numpy(synth.synth_noise(51, (3 * 4, H, D))) * 0.5).cuda()
    g_dev = _gc().hip_guide("EnvEmpty2D", [[], [], []], n_robots=3)
    g_dev.set_packed_constraints((ell, gso, gw, rgo))                 # general staging: (qx, qy, R, R|R|) on chip
    g_cmp = _gc().hip_guide("EnvEmpty2D", [[], [], []], n_robots=3)
    g_cmp.set_packed_constraints((ell, gso, gw, rgo, radius))         # uniform radius: only (qx, qy) on chip
    g_host = _gc().hip_guide("EnvEmpty2D", [[cases.soft_group(paths, r)] for r in (2, 3, 4)], n_robots=3)
    assert torch.equal(g_dev(x), g_host(x))
    assert torch.equal(g_cmp(x), g_host(x))
    gp = cases.guide_params("EnvEmpty2D")
    for k, r in enumerate((2, 3, 4)):
        ref = O.guide_grad(x[k * 4:(k + 1) * 4].cpu(), gp, [cases.soft_group(paths, r)], clip_mode="always")
        assert float((g_dev(x)[k * 4:(k + 1) * 4].cpu() - ref).abs().max()) < 2e-6


@pytest.mark.parametrize("n_all", [100, 300])
def test_large_constraint_tables(n_all):
    """99 / 299 slots per robot: the 8-wave workgroups stage the table in LDS (all of it when only (qx, qy) is kept, the
    first 288 / 144 slots otherwise) and read the rest from L2; both stagings agree with each other and with the oracle."""
    from mmd_amd.constraints import soft_constraints_from_paths
    starts, goals = synth.start_goal_circle(n_all, 0.8)
    paths = synth.straight_line_paths(starts, goals, H)
    cons = soft_constraints_from_paths(torch.from_numpy(paths).cuda(), 7, 2)
    x = (torch.from_numpy(synth.synth_noise(52, (2 * 8, H, D))) * 0.5).cuda()
    g_gen = _gc().hip_guide("EnvEmpty2D", [[], []], n_robots=2)
    g_gen.set_packed_constraints(cons[:4])
    g_cmp = _gc().hip_guide("EnvEmpty2D", [[], []], n_robots=2)
    g_cmp.set_packed_constraints(cons)
    out = g_cmp(x)
    if n_all - 1 <= 144:
        assert torch.equal(out, g_gen(x))          # both fully on chip: the same sums in the same order
    else:
        assert float((out - g_gen(x)).abs().max()) < 1e-6   # LDS / L2 split at 288 vs 144 slots: another sum order
    gp = cases.guide_params("EnvEmpty2D")
    ref = O.guide_grad(x[:8].cpu(), gp, [cases.soft_group(paths, 7)], clip_mode="always")
    assert float((out[:8].cpu() - ref).abs().max()) < 2e-6


@pytest.mark.parametrize("name", cases.SAMPLE_CASES)
def test_single_step_teacher_forced_golden(name):
    """Start from the REFERENCE's chain row r, run ONE ddpm step with the same injected noise, compare with the
    reference's row r+1.  This is the per-step parity statement; it does not suffer from the chaotic amplification
    of the guided loop (see test_run_inference_golden)."""
    from mmd_amd.diffusion_model import ddpm_sample_fn   # noqa: F401
    g = np.load(os.path.join(GOLDEN, f"g6_sample_{name}.npz"))
    case = cases.sample_case(name)
    T = case["T"]
    _, steps = cases.sample_inputs(case)
    model = _gc().hip_model(T)
    guide = _gc().hip_guide(case["map"], [case["cons"]]) if case.get("use_guide", True) else None
    hc = cases.hard_conds_for(case["start"], case["goal"])
    rows = [int(r) for r in g["rows"]]
    ref = torch.from_numpy(g["chain_rows"])
    full = os.path.join(GOLDEN, f"g6_full_{name}.npz")
    if os.path.exists(full):                              # every row of the reference's chain is stored: all T + 1 steps
        ref = torch.from_numpy(np.load(full)["chain"])
        rows = list(range(ref.shape[0]))
    n_pairs = 0
    for k, r in enumerate(rows[:-1]):
        if rows[k + 1] != r + 1:
            continue
        i = T - 1 - r                                     # loop index that maps chain[r] -> chain[r+1]
        x = ref[k].clone().cuda()
        model.sample_step(x, hc, i, guide=guide, n_guide_steps=20, t_start_guide=ceil(0.5 * T),
                          noise_std_extra_schedule_fn=lambda t: 0.5, noise=steps[r].cuda())
        err = rel_l2(x.cpu(), ref[k + 1])
        guided = guide is not None and i < ceil(0.5 * T)
        # unguided steps are smooth: 2e-5.  A guided step is 20 bang-bang (norm-clipped) gradient iterations whose
        # direction flips on rounding-size differences: the north-star 1e-3 (measured: 1e-6 ... 3e-4).
        bound = TOL_STEP_GUIDED if guided else TOL_STEP_PLAIN
        parity_log.record("single_step_teacher_forced", name, r, err, bound=bound, note="guided" if guided else "unguided")
        assert err < bound, (name, r, i, err)
        n_pairs += 1
    assert n_pairs >= (T + 1 if os.path.exists(full) else 5)


@pytest.mark.parametrize("name", cases.SAMPLE_CASES)
def test_run_inference_golden(name):
    """End-to-end chain against the reference.  The guided sampler is CHAOTIC: the genuine reference, run twice on
    CPU with its UNet output perturbed by a relative 1e-6 (a different fp32 summation order), differs from itself by
    `sens` (max over MMD_SENS_DRAWS = 24 perturbation draws per row, stored in the fixture by tools/make_golden.py:
    1e-1..3e-1 rel. L2 on the constraint cases, 7e-7 for the unguided prior).  So the bound per row is
    max(1e-3, 1.5 * LIN * sens) (cases.chaos_bounds: LIN = 3, a constant from the forward's fp64-bounded deviation; the kernel's
    measured deviation on the well-conditioned rows before guidance starts, in units of that 1e-6 perturbation, is asserted
    below it: 1.1 .. 2.3): the north-star 1e-3 wherever the reference
    itself is that reproducible, and "no further from the reference than the reference is from itself" elsewhere.  The sharp per-step statement is
    test_single_step_teacher_forced_golden; every measured error lands in r04_parity.json."""
    g = np.load(os.path.join(GOLDEN, f"g6_sample_{name}.npz"))
    case = cases.sample_case(name)
    xT, steps = cases.sample_inputs(case)
    chain = _gc().hip_run_inference(case, xT, steps).cpu()
    assert chain.shape == (case["T"] + 2, case["B"], H, D)
    assert torch.isfinite(chain).all()
    ref = torch.from_numpy(g["chain_rows"])
    failures = []
    errs = [rel_l2(chain[int(r)], ref[k]) for k, r in enumerate(g["rows"])]
    tsg = ceil(0.5 * case["T"])                              # chain row k is the state after k steps: guided from row T - tsg + 1 on
    n_unguided = sum(1 for r in g["rows"] if int(r) <= case["T"] - tsg)
    lin, bounds = chaos_bounds(errs, [float(v) for v in g["sens"]], n_unguided)
    assert lin < cases.LIN, (name, lin)                      # the kernel's per-step deviation: a few 1e-6
    for k, r in enumerate(g["rows"]):
        err, bound = errs[k], bounds[k]
        parity_log.record("run_inference_golden", name, r, err, sens=float(g["sens"][k]), bound=bound,
                          note=f"lin = {lin:.2f}")
        if not err < bound:
            failures.append((name, int(r), err, float(g["sens"][k]), bound))
    assert not failures, failures
    if not case.get("use_guide", True):
        assert rel_l2(chain[-1], ref[-1]) < TOL_FINAL


@pytest.mark.parametrize("name", cases.DDIM_CASES)
def test_ddim_sample_golden(name):
    """conditional_sample(ddim=True) (diffusion_model_base.py:213-290) against every chain row of the reference (g11).
    The DDIM chain is deterministic and takes ONE guide step per sampling step: well conditioned (sens 4e-7)."""
    g = np.load(os.path.join(GOLDEN, f"g11_ddim_{name}.npz"))
    case = cases.ddim_case(name)
    T, B = case["T"], case["B"]
    model = _gc().hip_model(T)
    guide = _gc().hip_guide(case["map"], [case["cons"]], n_robots=1) if case.get("use_guide", True) else None
    hc = cases.hard_conds_for(case["start"], case["goal"])
    xT = torch.from_numpy(synth.synth_noise(case["seed"], (B, H, D)))
    x, chain = model.ddim_sample((B, H, D), hc, n_diffusion_steps=T, return_chain=True, guide=guide,
                                 t_start_guide=ceil(0.5 * T), n_guide_steps=20, x_init=xT)
    ref = torch.from_numpy(g["chain"])
    chain = chain.transpose(0, 1).cpu()
    assert chain.shape == ref.shape
    for r in range(ref.shape[0]):
        parity_log.record("ddim_sample_golden", name, r, rel_l2(chain[r], ref[r]), sens=float(g["sens"]), bound=1e-4)
        assert rel_l2(chain[r], ref[r]) < 1e-4, (name, r, rel_l2(chain[r], ref[r]))
    assert torch.equal(x.cpu(), chain[-1])
    with pytest.raises(ValueError):
        model.conditional_sample(hc, T, batch_size=B, ddim=True, warm_start_path_b=xT)


def test_hard_rows_and_ddim_x0_golden():
    """Hard conditions on rows other than 0 / H-1 (apply_hard_conditioning takes any {row: state}, sample_functions.py:8-14): every
    guided DDPM step of the reference's chain with four pinned rows, teacher-forced (1e-3 each), the chain end to end (chaos bound
    + the pinned rows bit-exact in every row), the fused unguided step and the step API giving the same bits, a guided DDIM chain
    with the four rows, and DDIM with GaussianDiffusionModel(predict_epsilon=False) (diffusion_model_base.py:114-124, :248) --
    against the reference (g18)."""
    from mmd_amd.diffusion_model import GaussianDiffusionModel, ddpm_sample_fn
    from mmd_amd.temporal_unet import TemporalUnet
    g = np.load(os.path.join(GOLDEN, "g18_hard_rows_ddim_x0.npz"))
    starts, goals, soft, hard = cases.highways_case()
    hc2 = cases.hard_conds_for(starts[3], goals[3])
    hc4 = dict(hc2)
    for row, v in zip(g["via_rows"], g["via_states"]):
        hc4[int(row)] = O.normalize(torch.from_numpy(v), cases.MINS, cases.MAXS)
    gc = _gc()
    guide = gc.hip_guide("EnvHighways2D", [[soft, hard]])
    T, B, s_x, s_n = (int(v) for v in g["ddpm_meta"])
    model = gc.hip_model(T)
    xT = torch.from_numpy(synth.synth_noise(s_x, (B, H, D)))
    steps = torch.from_numpy(synth.synth_noise(s_n, (T + 1, B, H, D)))
    ref = torch.from_numpy(g["ddpm_chain"])
    hcd = {k: v.cuda() for k, v in hc4.items()}
    sd_o, tb_o, gp_o = O.state_dict_to_torch(synth.synth_unet_state_dict(0)), O.schedule_tables(T), cases.guide_params("EnvHighways2D")
    n_ill = 0
    for k in range(T + 1):
        i = T - 1 - k if k < T else 0
        y = ref[k].clone().cuda()
        model.sample_step(y, hcd, i, guide=guide, n_guide_steps=20, t_start_guide=ceil(0.5 * T),
                          noise_std_extra_schedule_fn=lambda t: 0.5, noise=(steps[k] if k < T else torch.zeros_like(steps[k])).cuda())
        err = rel_l2(y.cpu(), ref[k + 1])
        bound, sens = 1e-3, None
        if err >= bound:
            # One trajectory of this step (i = 6) amplifies a rounding-sized change of eps 500 .. 800 times (20 norm-clipped iterations
            # over hinge constraints next to a pinned via state): the reference-shaped oracle itself lands 1.6e-3 away from the golden
            # row on the GPU box's CPU and bit-exact on the CPU that generated it.  Yardstick = cases.chaos_bounds' rule for one step:
            # the oracle's OWN response to relative 1e-6 perturbations of eps (16 draws), times LIN (the kernel's forward is within 3
            # such units of the fp32 reference), times 1.5 for the tail.
            step = lambda pert=None: O.apply_hard_conditioning(O.ddpm_sample_step(                   # noqa: E731
                sd_o, tb_o, ref[k].clone(), hc4, i, guide=lambda z: O.guide_grad(z, gp_o, [soft, hard]), n_guide_steps=20,
                t_start_guide=ceil(0.5 * T), noise=steps[k] if k < T else torch.zeros_like(steps[k]), noise_std_extra=0.5,
                eps_rel_perturb=pert), hc4)
            base = step()
            gen = torch.Generator().manual_seed(3000 + k)
            sens = max(rel_l2(step(1e-6 * torch.randn(ref[k].shape, generator=gen)), base) for _ in range(16))
            bound = max(bound, 1.5 * cases.LIN * sens)
            n_ill += 1
            # ... and WHO is beyond 1e-3: per trajectory against the oracle evaluated on THIS host, the kernel's step is within the
            # tolerance or a shown branch flip (the kernel's decision trace differs from the oracle's, and the oracle on the kernel's
            # decisions agrees with the kernel).  Round 5 measured all four within 1e-3: the party that moved is the golden row --
            # the reference run on the CPU that generated the fixture takes another branch than the same arithmetic on this host.
            nz = steps[k] if k < T else torch.zeros_like(steps[k])
            jd = gc.GuidedStepJudge(model, guide, ref[k].clone(), hcd, i, ceil(0.5 * T), 1, nz, y.cpu())
            verdicts = [jd.check("hard_rows_step_attribution", f"row{k + 1}_traj{j}", j, sd_o, tb_o, gp_o, [soft, hard], hc4, 3000 + j)[0]
                        for j in range(B)]
            assert "sens" not in verdicts, verdicts
        parity_log.record("hard_rows_teacher_forced_step", f"row{k + 1}", i, err, sens=sens, bound=bound)
        assert err < bound, (k, i, err, sens)
    assert n_ill <= 1, n_ill                       # (25 of the 26 steps hold the plain 1e-3, measured <= 1.2e-5)
    chain = model.run_inference(None, hc4, n_samples=B, horizon=H, return_chain=True, sample_fn=ddpm_sample_fn, guide=guide,
                                n_guide_steps=20, t_start_guide=ceil(0.5 * T), noise_std_extra_schedule_fn=lambda t: 0.5,
                                n_diffusion_steps_without_noise=1, warm_start_path_b=xT.cuda(), step_noise=steps.cuda()).cpu()
    for row, val in hc4.items():
        assert torch.equal(chain[:, :, row], val.expand(T + 2, B, D)), row          # every chain row, x_T included
    errs = [rel_l2(chain[k], ref[k]) for k in range(T + 2)]
    n_unguided = T - ceil(0.5 * T) + 1
    lin, bounds = cases.chaos_bounds(errs, list(g["ddpm_sens"]), n_unguided)
    for k in range(T + 2):
        parity_log.record("hard_rows_chain", f"row{k}", None, errs[k], sens=float(g["ddpm_sens"][k]), bound=bounds[k])
        assert errs[k] < bounds[k], (k, errs[k], bounds[k])
    assert max(errs[:n_unguided]) < 1e-3 and lin < cases.LIN
    # the unguided steps of that run went through the step fused into the UNet launch: same bits as the two-launch step API
    y = chain[3].clone().cuda()
    model.sample_step(y, hcd, T - 4, guide=guide, n_guide_steps=20, t_start_guide=ceil(0.5 * T),
                      noise_std_extra_schedule_fn=lambda t: 0.5, noise=steps[3].cuda())
    assert torch.equal(y.cpu(), chain[4])
    # DDIM, four pinned rows, guided
    T2, B2, s2 = (int(v) for v in g["ddim_meta"])
    x, ch = gc.hip_model(T2).ddim_sample((B2, H, D), hc4, n_diffusion_steps=T2, return_chain=True, guide=guide,
                                         t_start_guide=ceil(0.5 * T2), n_guide_steps=20,
                                         x_init=torch.from_numpy(synth.synth_noise(s2, (B2, H, D))))
    ch, ref2 = ch.transpose(0, 1).cpu(), torch.from_numpy(g["ddim_chain"])
    assert ch.shape == ref2.shape
    e2 = max(rel_l2(ch[r], ref2[r]) for r in range(ref2.shape[0]))
    parity_log.record("hard_rows_ddim", "worst_row", None, e2, bound=1e-4)
    assert e2 < 1e-4, e2
    # DDIM of an x0-predicting model
    T3, B3, s3 = (int(v) for v in g["ddim_x0_meta"])
    unet = TemporalUnet(state_dim=4, n_support_points=64, unet_input_dim=32, dim_mults=(1, 2, 4))
    unet.load_state_dict(synth.synth_unet_state_dict(0))
    m0 = GaussianDiffusionModel(model=unet, variance_schedule="exponential", n_diffusion_steps=T3, predict_epsilon=False)
    x, ch0 = m0.ddim_sample((B3, H, D), hc2, n_diffusion_steps=T3, return_chain=True,
                            x_init=torch.from_numpy(synth.synth_noise(s3, (B3, H, D))))
    ch0, ref3 = ch0.transpose(0, 1).cpu(), torch.from_numpy(g["ddim_x0_chain"])
    e3 = max(rel_l2(ch0[r], ref3[r]) for r in range(ref3.shape[0]))
    parity_log.record("ddim_predict_x0", "worst_row", None, e3, bound=1e-4)
    assert e3 < 1e-4, e3


def test_run_local_inference_golden():
    g = np.load(os.path.join(GOLDEN, "g7_local.npz"))
    T, B = 50, 8
    starts, goals, soft, hard = cases.highways_case()
    model = _gc().hip_model(T)
    guide = _gc().hip_guide("EnvHighways2D", [[soft, hard]])
    a = np.linspace(0, 1, H, dtype=np.float32)[None, :, None]
    pos = starts[3][None, None] * (1 - a) + goals[3][None, None] * a
    seed = np.concatenate([np.repeat(pos, B, 0), np.zeros((B, H, 2), np.float32)], -1)
    seed = torch.from_numpy((seed + 0.02 * synth.synth_noise(23, (B, H, D))).astype(np.float32)).cuda()
    qn = torch.from_numpy(synth.synth_noise(24, (B, H, D))).cuda()
    steps = torch.from_numpy(synth.synth_noise(25, (4, B, H, D))).cuda()
    from mmd_amd.diffusion_model import ddpm_sample_fn
    chain = model.run_local_inference(seed, 3, 3, None, cases.hard_conds_for(starts[3], goals[3]), n_samples=B,
                                      horizon=H, return_chain=True, sample_fn=ddpm_sample_fn, guide=guide,
                                      n_guide_steps=20, t_start_guide=25, noise_std_extra_schedule_fn=lambda x: 0.5,
                                      n_diffusion_steps_without_noise=1, q_noise=qn, step_noise=steps).cpu()
    ref = torch.from_numpy(g["chain"])
    assert chain.shape == ref.shape
    # row 0 = the forward-noised seed (smooth); rows 1..4 are four CHAINED guided steps (chaotic, see above): bound by
    # the reference's own sensitivity to a 1e-6 UNet perturbation
    for r in range(5):
        err = rel_l2(chain[r], ref[r])
        bound = 2e-5 if r == 0 else max(TOL_FINAL, SENS_FACTOR * float(g["sens"][r]))
        parity_log.record("run_local_inference_golden", "highways_T50", r, err, sens=float(g["sens"][r]), bound=bound)
        assert err < bound, (r, err, g["sens"])
    # teacher-forced: each single step from the reference's own state
    hc = cases.hard_conds_for(starts[3], goals[3])
    for r in range(4):
        x = ref[r].clone().cuda()
        model.sample_step(x, hc, 2 - r, guide=guide, n_guide_steps=20, t_start_guide=25,
                          noise_std_extra_schedule_fn=lambda t: 0.5, noise=steps[r])
        parity_log.record("run_local_inference_teacher_forced", "highways_T50", r, rel_l2(x.cpu(), ref[r + 1]),
                          bound=TOL_STEP_GUIDED)
        assert rel_l2(x.cpu(), ref[r + 1]) < TOL_STEP_GUIDED, (r, rel_l2(x.cpu(), ref[r + 1]))


def test_multi_robot_batch_equals_per_robot():
    """Robots are independent given their constraints: one batched call over 3 robots must reproduce, bit for bit,
    the three single-robot calls (the property the multi-GPU sharding relies on)."""
    from mmd_amd.diffusion_model import ddpm_sample_fn
    T, B, R = 25, 4, 3
    starts, goals = synth.start_goal_circle(6, 0.8)
    paths = synth.straight_line_paths(starts, goals, H)
    model = _gc().hip_model(T)
    xT = torch.from_numpy(synth.synth_noise(60, (R * B, H, D))).cuda()
    steps = torch.from_numpy(synth.synth_noise(61, (T + 1, R * B, H, D))).cuda()
    hc_all = {0: torch.stack([cases.hard_conds_for(starts[r], goals[r])[0] for r in range(R)]),
              H - 1: torch.stack([cases.hard_conds_for(starts[r], goals[r])[H - 1] for r in range(R)])}
    kw = dict(horizon=H, return_chain=False, sample_fn=ddpm_sample_fn, n_guide_steps=20, t_start_guide=13,
              noise_std_extra_schedule_fn=lambda x: 0.5, n_diffusion_steps_without_noise=1)
    guide = _gc().hip_guide("EnvEmpty2D", [[cases.soft_group(paths, r)] for r in range(R)], n_robots=R)
    out = model.run_inference(None, hc_all, n_samples=B, n_robots=R, guide=guide, warm_start_path_b=xT,
                              step_noise=steps, **kw)
    for r in range(R):
        g1 = _gc().hip_guide("EnvEmpty2D", [[cases.soft_group(paths, r)]])
        o1 = model.run_inference(None, cases.hard_conds_for(starts[r], goals[r]), n_samples=B, guide=g1,
                                 warm_start_path_b=xT[r * B:(r + 1) * B],
                                 step_noise=steps[:, r * B:(r + 1) * B].contiguous(), **kw)
        assert torch.equal(out[r * B:(r + 1) * B], o1), r


def test_stream_chunking_is_bit_identical():
    """mmd_p_sample_loop split over 2 / 3 concurrent HIP streams == the single-stream run, bit for bit (Philox path,
    so the in-kernel noise indexing is covered too)."""
    from mmd_amd.diffusion_model import ddpm_sample_fn
    T, B, R = 25, 8, 5
    starts, goals = synth.start_goal_circle(R, 0.8)
    paths = synth.straight_line_paths(starts, goals, H)
    model = _gc().hip_model(T)
    hc_all = {0: torch.stack([cases.hard_conds_for(starts[r], goals[r])[0] for r in range(R)]),
              H - 1: torch.stack([cases.hard_conds_for(starts[r], goals[r])[H - 1] for r in range(R)])}
    guide = _gc().hip_guide("EnvHighways2D", [[cases.soft_group(paths, r)] for r in range(R)], n_robots=R)
    kw = dict(horizon=H, return_chain=True, sample_fn=ddpm_sample_fn, n_guide_steps=20, t_start_guide=13,
              noise_std_extra_schedule_fn=lambda x: 0.5, n_diffusion_steps_without_noise=1, guide=guide, seed=77,
              n_samples=B, n_robots=R)
    ref = model.run_inference(None, hc_all, n_streams=1, **kw)
    for ns in (2, 3):
        out = model.run_inference(None, hc_all, n_streams=ns, **kw)
        assert torch.equal(out, ref), ns


def test_fused_unguided_steps_equal_the_step_api_bitwise():
    """mmd_p_sample_loop applies the steps WITHOUT guidance inside the UNet launch (FusedStep, unet.hip); mmd_ddpm_step keeps the
    UNet launch + ddpm_guide_kernel form.  Both run the same explicit-fma helpers (guide_dev.h), so the whole chain of a guided
    call -- 12 fused steps, then 14 guided ones -- must equal the step-by-step replay BIT FOR BIT (injected noise: the two entry
    points number their Philox draws differently); B = 5 also covers a workgroup with invalid waves.
    ref: sample_functions.py:40-86."""
    from mmd_amd.diffusion_model import ddpm_sample_fn
    T, B, R = 25, 5, 3
    starts, goals = synth.start_goal_circle(R, 0.8)
    paths = synth.straight_line_paths(starts, goals, H)
    model = _gc().hip_model(T)
    hc_all = {0: torch.stack([cases.hard_conds_for(starts[r], goals[r])[0] for r in range(R)]),
              H - 1: torch.stack([cases.hard_conds_for(starts[r], goals[r])[H - 1] for r in range(R)])}
    guide = _gc().hip_guide("EnvHighways2D", [[cases.soft_group(paths, r)] for r in range(R)], n_robots=R)
    xT = torch.from_numpy(synth.synth_noise(80, (R * B, H, D))).cuda()
    steps = torch.from_numpy(synth.synth_noise(81, (T + 1, R * B, H, D))).cuda()
    kw = dict(horizon=H, return_chain=True, sample_fn=ddpm_sample_fn, n_guide_steps=20, t_start_guide=13,
              noise_std_extra_schedule_fn=lambda x: 0.5, n_diffusion_steps_without_noise=1, guide=guide, n_samples=B, n_robots=R,
              warm_start_path_b=xT.clone())
    chain = model.run_inference(None, hc_all, step_noise=steps, **kw)                  # [T + 2, R B, H, D]
    x = chain[0].clone()
    assert torch.equal(x[:, 1:-1], xT[:, 1:-1])
    for k, i in enumerate(reversed(range(-1, T))):
        model.sample_step(x, hc_all, i, guide=guide, n_guide_steps=20, t_start_guide=13, noise_std_extra_schedule_fn=lambda t: 0.5,
                          n_robots=R, noise=steps[k])
        assert torch.equal(x, chain[k + 1]), (k, i, float((x - chain[k + 1]).abs().max()))


def test_philox_noise_statistics():
    """Production path: in-kernel Philox draws (no injected noise) are N(0,1) and reproducible per seed."""
    model = _gc().hip_model(25)
    hc = {}
    kw = dict(hard_conds=hc, n_diffusion_steps=0, n_diffusion_steps_without_noise=0, return_chain=False)
    a = model.p_sample_loop((4096, H, D), seed=123, **kw)
    b = model.p_sample_loop((4096, H, D), seed=123, **kw)
    c = model.p_sample_loop((4096, H, D), seed=124, **kw)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert abs(float(a.mean())) < 5e-3 and abs(float(a.std()) - 1.0) < 5e-3
    assert abs(float((a ** 4).mean()) - 3.0) < 0.05


@pytest.mark.parametrize("n_robots,B", [(48, 4), (48, 8), (160, 16)])
def test_guide_large_constraint_tables(n_robots, B):
    """More constraint slots than the LDS staging holds (4-wave workgroups: 40; 16-wave workgroups: 144): the overflow
    is read from the L2-resident table.  47 / 159 other robots + a hard group, both workgroup shapes, vs the oracle."""
    starts, goals = synth.start_goal_circle(n_robots, 0.8)
    paths = synth.straight_line_paths(starts, goals, H)
    soft = cases.soft_group(paths, 1)
    hard = cases.hard_group([[0.3, 0.1], [-0.2, 0.4]], [[10, 30], [25, 40]])
    assert soft.q.shape[0] == (n_robots - 1) * 63
    guide = _gc().hip_guide("EnvHighways2D", [[soft, hard]])
    gp = cases.guide_params("EnvHighways2D")
    x = torch.from_numpy(synth.synth_noise(95, (B, H, D))) * 0.5
    ref = O.guide_grad(x, gp, [soft, hard], clip_mode="always")
    out = guide(x.cuda()).cpu()
    assert float((out - ref).abs().max()) < 3e-6
    y = x.clone().cuda()
    guide.guide_steps(y, torch.zeros(1, 2, D, device="cuda"), 0, 5)
    r = x.clone()
    for _ in range(5):
        r = r + O.guide_grad(r, gp, [soft, hard], clip_mode="always")
    assert rel_l2(y.cpu(), r) < 1e-4


def test_calls_follow_the_tensors_device_not_torchs_current_device():
    """one TemporalUnet / guide may serve several GPUs -- every C call runs under a device guard on the GPU that owns
    its tensors, on THAT device's current stream, with the handle / workspace resolved for it.  With a second GPU: a forward and a
    guided step on cuda:1 tensors while cuda:0 is current equal the cuda:0 results bit for bit.  On a one-GPU box the same
    property is exercised through a non-default current stream (the launch must land on the stream torch reports for the
    tensors' device, and results must match the default-stream call)."""
    model = _gc().hip_model(25)
    x = torch.from_numpy(synth.synth_noise(70, (6, H, D)))
    ref = model.model(x.cuda(), 7).cpu()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        y = model.model(x.cuda(), 7)
    st.synchronize()
    assert torch.equal(y.cpu(), ref)
    if torch.cuda.device_count() < 2:
        pytest.skip("the cross-device leg needs 2 GPUs (the one-GPU leg above -- a non-default current stream -- has run and passed)")
    x1 = x.to("cuda:1")
    torch.cuda.set_device(0)
    y1 = model.model(x1, 7)                                  # tensors on cuda:1, current device cuda:0
    torch.cuda.synchronize(1)
    assert y1.device.index == 1 and torch.equal(y1.cpu(), ref)


def test_persistent_run_equals_launch_per_step():
    """mmd_sampler_desc.flags & MMD_SAMPLER_PERSIST (opt-in; a field of the call's descriptor, so both modes run in this process): the
    leading run of unguided steps in ONE launch per <= 64 steps (unet_persist_kernel<2> / <4>: a workgroup iterates the steps of its
    own trajectories) gives the launch-per-step results bit for bit -- every chain row of a T = 100 guided call (50 leading unguided
    steps) and a prior-only call (101 steps: two runs), Philox and injected noise.  MMD_SAMPLER_NO_FUSED_STEP (unguided steps as
    separate step-kernel launches) likewise."""
    import gpu_common
    from mmd_amd import _lib
    from mmd_amd.diffusion_model import ddpm_sample_fn
    T = 100
    model = gpu_common.hip_model(T)
    starts, goals = synth.start_goal_circle(6, 0.8)
    paths = synth.straight_line_paths(starts, goals, H)
    res = {}
    try:
        for flags in (0, _lib.SAMPLER_PERSIST, _lib.SAMPLER_NO_FUSED_STEP):
            model.sampler_flags = flags
            out = {}
            for tag, R, B in (("one_per_workgroup", 2, 8), ("two_per_workgroup", 3, 100), ("four_per_workgroup", 3, 200)):
                guide = gpu_common.hip_guide("EnvHighways2D", [[cases.soft_group(paths, r)] for r in range(R)], n_robots=R)
                hc = {0: torch.stack([cases.hard_conds_for(starts[r], goals[r])[0] for r in range(R)]),
                      H - 1: torch.stack([cases.hard_conds_for(starts[r], goals[r])[H - 1] for r in range(R)])}
                # guided from i = 49 on (50 leading unguided steps), in-kernel Philox noise, the whole chain
                out[tag] = model.run_inference(None, hc, n_samples=B, n_robots=R, horizon=H, return_chain=True, sample_fn=ddpm_sample_fn,
                                               guide=guide, n_guide_steps=20, t_start_guide=50, noise_std_extra_schedule_fn=lambda t: 0.5,
                                               n_diffusion_steps_without_noise=1, seed=91).cpu()
                # prior only with injected noise: 101 unguided steps = a run of 64 + a run of 37
                xT = torch.from_numpy(synth.synth_noise(130, (R * B, H, D)))
                st = torch.from_numpy(synth.synth_noise(131, (T + 1, R * B, H, D))) if R * B <= 64 else None
                out[tag + "_prior"] = model.run_inference(None, hc, n_samples=B, n_robots=R, horizon=H, return_chain=st is not None,
                                                          sample_fn=ddpm_sample_fn, guide=None, noise_std_extra_schedule_fn=lambda t: 0.5,
                                                          n_diffusion_steps_without_noise=1, warm_start_path_b=xT.cuda(),
                                                          step_noise=None if st is None else st.cuda(), seed=92).cpu()
            res[flags] = out
    finally:
        model.sampler_flags = 0
    assert len(res[0]) == 6
    for flags in (_lib.SAMPLER_PERSIST, _lib.SAMPLER_NO_FUSED_STEP):
        for k in res[0]:
            assert torch.isfinite(res[0][k]).all() and torch.equal(res[0][k], res[flags][k]), (flags, k)
