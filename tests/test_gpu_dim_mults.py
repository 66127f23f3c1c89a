"""-m gpu: TemporalUnet configurations other than the fused kernel's -- first of all the reference's UNET_DIM_MULTS[1] = (1, 2, 4, 8)
(mmd/models/__init__.py:8-11, selected by a checkpoint's args.yaml at mpd.py:158) -- which run layer by layer (csrc/unet_layers.hip),
against the reference (g17) and against the fused kernel on the configuration both serve."""
import os
from math import ceil

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mmd_amd import synth                # noqa: E402
import cases                             # noqa: E402
import parity_log                        # noqa: E402
from cases import GOLDEN, H, D, rel_l2   # noqa: E402

NETS = (("d32_1248", 32, (1, 2, 4, 8)), ("d16_12", 16, (1, 2)), ("d8_1", 8, (1,)), ("d64_124", 64, (1, 2, 4)))
TOL_STEP = 1e-3


def _unet(uid, dm, seed=0, layered=False, **options):
    from mmd_amd.temporal_unet import TemporalUnet
    u = TemporalUnet(state_dim=4, n_support_points=H, unet_input_dim=uid, dim_mults=dm, layered=layered, **options)
    u.load_state_dict(synth.synth_unet_state_dict(seed, unet_input_dim=uid, dim_mults=dm))
    return u


def _model(T, dm=(1, 2, 4, 8)):
    from mmd_amd.diffusion_model import GaussianDiffusionModel
    return GaussianDiffusionModel(model=_unet(32, dm), variance_schedule="exponential", n_diffusion_steps=T, predict_epsilon=True)


@pytest.mark.parametrize("tag,uid,dm", NETS, ids=[n[0] for n in NETS])
def test_layered_unet_forward_golden(tag, uid, dm):
    g = np.load(os.path.join(GOLDEN, "g17_unet_dim_mults.npz"))
    unet = _unet(uid, dm)
    x = torch.from_numpy(synth.synth_noise(int(g["x_seed"]), (4, H, D))).cuda()
    for t in g["ts"]:
        out = unet(x, torch.full((4,), int(t), dtype=torch.long, device="cuda")).cpu()
        err = rel_l2(out, g[f"{tag}.eps_t{t}"])
        parity_log.record("layered_unet_forward_golden", f"{tag}_t{int(t)}", None, err, bound=2e-5)
        assert err < 2e-5, (tag, int(t), err)


def test_layered_path_equals_fused_kernel_on_option0():
    """The two implementations share no device code: the fp32 layer-by-layer kernels forced onto the fused kernel's own
    configuration (TemporalUnet(layered=True) -> mmd_unet_options.flags & MMD_UNET_LAYERED) must give the fused kernel's output within its
    documented distance from the fp32 reference (2.4e-6 rel-L2, test_unet_forward_accuracy_against_fp64), at a batch that is not
    a multiple of the fused kernel's workgroup size."""
    n = 37
    x = (torch.from_numpy(synth.synth_noise(901, (n, H, D))) * 0.7).cuda()
    fused = _unet(32, (1, 2, 4), layered=False)
    layered = _unet(32, (1, 2, 4), layered=True)              # (a creation-time option of the device model, fixed per object)
    layered.handle(25, "cuda")
    fused.handle(25, "cuda")
    assert layered.handle(device="cuda").value != fused.handle(device="cuda").value
    for t in (0, 11, 24):
        err = rel_l2(layered(x, t).cpu(), fused(x, t).cpu())
        parity_log.record("layered_vs_fused", f"t{t}", None, err, bound=6e-6)
        assert err < 6e-6, (t, err)
    from mmd_amd import _lib
    lib = _lib.load()
    assert lib.mmd_unet_workspace_bytes(layered.handle(device="cuda"), n) > lib.mmd_unet_workspace_bytes(fused.handle(device="cuda"), n)


@pytest.mark.parametrize("dm", [(1, 2, 4), (1, 2, 4, 8)], ids=["option0", "option1"])
def test_layered_forward_bits_do_not_depend_on_the_batch(monkeypatch, dm):
    """The launch shape of the layer kernels follows the batch size (matrix-pipe kernel: slices of 16 .. 128 GEMM columns per workgroup, 1 .. 8
    items per workgroup; vector-ALU kernels: 1, 2 or 4 output channels per thread); every sum is DEFINED as a fixed tree (the K order of the
    GEMM, the GroupNorm statistics' balanced tree over channels and rows), so a trajectory's eps must not change by a bit with the size of
    the batch it sits in -- n = 1, 6, 13 (partial items at every level), 200, 800 and 4099 (the widest slices, several items per workgroup, a
    partial last item) -- and for option 0 every size must agree with the fused kernel."""
    layered = _unet(32, dm, layered=True)
    layered.handle(25, "cuda")
    x = (torch.from_numpy(synth.synth_noise(902, (4099, H, D))) * 0.7).cuda()
    huge = layered(x, 7)
    assert torch.isfinite(huge).all()
    big = layered(x[:800].contiguous(), 7)
    assert torch.equal(big, huge[:800])
    assert torch.equal(layered(x[:200].contiguous(), 7), big[:200])
    assert torch.equal(layered(x[600:608].contiguous(), 7), big[600:608])
    assert torch.equal(layered(x[4093:4099].contiguous(), 7), huge[4093:4099])
    assert torch.equal(layered(x[5:6].contiguous(), 7), huge[5:6])                 # a batch of one
    assert torch.equal(layered(x[100:113].contiguous(), 7), huge[100:113])
    x = x[:800].contiguous()
    if dm == (1, 2, 4):
        fused = _unet(32, dm, layered=False)
        fused.handle(25, "cuda")
        err = rel_l2(big.cpu(), fused(x, 7).cpu())
        parity_log.record("layered_vs_fused", "n800_t7", None, err, bound=6e-6)
        assert err < 6e-6, err


def _g17_chain_case():
    g = np.load(os.path.join(GOLDEN, "g17_unet_dim_mults.npz"))
    T, B, s_x, s_n = (int(v) for v in g["meta"])
    starts, goals, soft, hard = cases.highways_case()
    return g, T, B, s_x, s_n, cases.hard_conds_for(starts[3], goals[3]), soft, hard


def test_option1_guided_steps_teacher_forced_vs_reference():
    """Every ddpm_sample_fn step of the reference's guided Highways chain through the FOUR-level network (g17: T = 25, B = 4; 13
    steps with 20 guide iterations each), started from the reference's own previous row: one step is well conditioned, so each is
    held to the north-star tolerance against the reference itself."""
    import gpu_common
    g, T, B, s_x, s_n, hc, soft, hard = _g17_chain_case()
    model = _model(T)
    guide = gpu_common.hip_guide("EnvHighways2D", [[soft, hard]])
    steps = torch.from_numpy(synth.synth_noise(s_n, (T + 1, B, H, D)))
    ref = torch.from_numpy(g["chain"])
    hcd = {k: v.cuda() for k, v in hc.items()}
    worst = 0.0
    for k in range(T + 1):
        i = T - 1 - k if k < T else 0
        y = ref[k].clone().cuda()
        model.sample_step(y, hcd, i, guide=guide, n_guide_steps=20, t_start_guide=ceil(0.5 * T),
                          noise_std_extra_schedule_fn=lambda t: 0.5, noise=(steps[k] if k < T else torch.zeros_like(steps[k])).cuda())
        err = rel_l2(y.cpu(), ref[k + 1])
        parity_log.record("option1_teacher_forced_step", f"row{k + 1}", i, err, bound=TOL_STEP)
        worst = max(worst, err)
        assert err < TOL_STEP, (k, i, err)
    assert worst < TOL_STEP


def test_option1_chain_vs_reference():
    """The same chain end to end (run_inference with injected noise).  Its map noise -> trajectory is chaotic once guidance
    starts (the reference's own response to 1e-6 relative perturbations reaches 0.2 on the last row): rows are held to
    cases.chaos_bounds, the unguided rows to 1e-3."""
    import gpu_common
    from mmd_amd.diffusion_model import ddpm_sample_fn
    g, T, B, s_x, s_n, hc, soft, hard = _g17_chain_case()
    model = _model(T)
    guide = gpu_common.hip_guide("EnvHighways2D", [[soft, hard]])
    xT = torch.from_numpy(synth.synth_noise(s_x, (B, H, D)))
    steps = torch.from_numpy(synth.synth_noise(s_n, (T + 1, B, H, D)))
    chain = model.run_inference(None, hc, n_samples=B, horizon=H, return_chain=True, sample_fn=ddpm_sample_fn, guide=guide,
                                n_guide_steps=20, t_start_guide=ceil(0.5 * T), noise_std_extra_schedule_fn=lambda t: 0.5,
                                n_diffusion_steps_without_noise=1, warm_start_path_b=xT.cuda(), step_noise=steps.cuda()).cpu()
    ref = torch.from_numpy(g["chain"])
    errs = [rel_l2(chain[k], ref[k]) for k in range(T + 2)]
    n_unguided = T - ceil(0.5 * T) + 1                       # rows 0 .. : x_T and the steps t = T-1 .. t_start_guide
    lin, bounds = cases.chaos_bounds(errs, list(g["sens"]), n_unguided)
    for k in range(T + 2):
        parity_log.record("option1_chain", f"row{k}", None, errs[k], sens=float(g["sens"][k]), bound=bounds[k])
        assert errs[k] < bounds[k], (k, errs[k], bounds[k])
    assert max(errs[:n_unguided]) < 1e-3 and lin < cases.LIN, (errs[:n_unguided], lin)


def test_option1_multi_robot_sampler_shards_bitwise_and_ignores_stream_chunks():
    """MultiRobotSampler over a four-level model: the layer-by-layer path keeps its activations in the one workspace, so the
    sampler must not split the robots over concurrent streams (n_streams = 2 gives the n_streams = 1 bits), and a rank's shard
    equals its rows of the unsharded run."""
    from mmd_amd.multi_robot import MultiRobotSampler
    T, R, B = 25, 6, 4
    starts, goals = synth.start_goal_circle(R, 0.6)
    paths = torch.from_numpy(synth.straight_line_paths(starts, goals, H)).cuda()
    model = _model(T)

    def run(**kw):
        s = MultiRobotSampler(model, starts, goals, env_id="EnvEmpty2D", n_samples=B, **kw)
        s.set_other_paths(paths)
        return s.sample(seed=5).cpu()
    whole = run(n_streams=1)
    assert torch.isfinite(whole).all() and whole.shape == (R * B, H, D)
    assert torch.equal(whole, run(n_streams=2))
    assert torch.equal(whole[2 * B:4 * B], run(rank=1, world_size=3))


def test_option1_planner_from_args_yaml(tmp_path):
    """mpd.py:117-177 with args.yaml's unet_dim_mults_option = 1: MPD builds the four-level TemporalUnet from the checkpoint and
    plans through it."""
    import yaml
    from mmd_amd.planners import MPD
    from mmd_amd.schedules import diffusion_buffers
    model_id = "EnvEmpty2D-RobotPlanarDisk"
    mdir = tmp_path / "models" / model_id
    (mdir / "checkpoints").mkdir(parents=True)
    yaml.safe_dump(dict(variance_schedule="exponential", n_diffusion_steps=25, predict_epsilon=True, unet_input_dim=32,
                        unet_dim_mults_option=1, use_ema=True, dataset_subdir=model_id, include_velocity=True),
                   open(mdir / "args.yaml", "w"))
    sd = {"model." + k: torch.from_numpy(v) for k, v in synth.synth_unet_state_dict(0, dim_mults=(1, 2, 4, 8)).items()}
    sd.update(diffusion_buffers(25))
    torch.save(sd, mdir / "checkpoints" / "ema_model_current_state_dict.pth")
    p = MPD(model_id=model_id, planner_alg="mmd", start_state_pos=torch.tensor([-0.5, 0.1]), goal_state_pos=torch.tensor([0.5, -0.1]),
            n_samples=8, trained_models_dir=str(tmp_path / "models"), device="cuda")
    assert p.model.model.dim_mults == (1, 2, 4, 8)
    res = p(torch.tensor([-0.5, 0.1]), torch.tensor([0.5, -0.1]))
    tf = res.trajs_iters[-1]
    assert tuple(tf.shape) == (8, H, D) and torch.isfinite(res.trajs_iters).all()
    assert torch.allclose(tf[:, 0, :2].cpu(), torch.tensor([-0.5, 0.1]).expand(8, 2), atol=1e-5)


def test_option1_ddim_and_ensemble_paths_vs_oracle():
    """The other sampler entry points over a four-level model: ddim_sample (mmd_ddim_sample) against the oracle's DDIM on the same
    weights, every chain row (the DDIM chain is deterministic and well conditioned: 1e-4), and the unguided p_sample_loop through the
    step API == through the loop (the layer-by-layer path has no fused step: both are UNet + step kernel launches)."""
    from oracle import mmd_oracle as O
    from mmd_amd.diffusion_model import ddpm_sample_fn
    T, B = 50, 4
    model = _model(T)
    starts, goals, soft, hard = cases.highways_case()
    hc = cases.hard_conds_for(starts[3], goals[3])
    xT = torch.from_numpy(synth.synth_noise(77, (B, H, D)))
    x, ch = model.ddim_sample((B, H, D), hc, n_diffusion_steps=T, return_chain=True, x_init=xT)
    sd = O.state_dict_to_torch(synth.synth_unet_state_dict(0, dim_mults=(1, 2, 4, 8)))
    ref = O.ddim_sample(sd, O.schedule_tables(T), xT, hc, T)
    ch = ch.transpose(0, 1).cpu()
    assert ch.shape == ref.shape
    err = max(rel_l2(ch[r], ref[r]) for r in range(ref.shape[0]))
    parity_log.record("option1_ddim_vs_oracle", "worst_row", None, err, bound=1e-4)
    assert err < 1e-4, err
    steps = torch.from_numpy(synth.synth_noise(78, (T + 1, B, H, D))).cuda()
    chain = model.run_inference(None, hc, n_samples=B, horizon=H, return_chain=True, sample_fn=ddpm_sample_fn, guide=None,
                                noise_std_extra_schedule_fn=lambda t: 0.5, n_diffusion_steps_without_noise=1,
                                warm_start_path_b=xT.cuda(), step_noise=steps)
    y = chain[0].clone()
    hcd = {k: v.cuda() for k, v in hc.items()}
    for k in range(3):
        model.sample_step(y, hcd, T - 1 - k, guide=None, noise_std_extra_schedule_fn=lambda t: 0.5, noise=steps[k])
        assert torch.equal(y, chain[k + 1]), k


@pytest.mark.parametrize("uid", [24, 40, 48, 56])
def test_layered_unet_any_group_width_vs_oracle(uid):
    """unet_input_dim 40 / 56 give GroupNorm groups of 5 / 7 (x 2^level) channels, which do not cut into the 16 .. 64
    register tiles a workgroup's thread group takes (the launch then had 320 / 448 threads and failed); a thread now loops over its
    thread group's tiles.  Every accepted width x 3 and 4 levels against the oracle's forward (pinned by g2 / g17), at the three
    batch sizes that select 1, 2 and 4 output channels per thread -- the bits must not depend on the batch."""
    from oracle import mmd_oracle as O
    for dm in ((1, 2, 4), (1, 2, 4, 8)):
        unet = _unet(uid, dm)
        sd = O.state_dict_to_torch(synth.synth_unet_state_dict(0, unet_input_dim=uid, dim_mults=dm))
        x = torch.from_numpy(synth.synth_noise(905, (800, H, D))) * 0.7
        big = unet(x.cuda(), 11)
        assert torch.isfinite(big).all()
        ref = O.unet_forward(sd, x[:6], torch.full((6,), 11, dtype=torch.long))
        err = rel_l2(big[:6].cpu(), ref)
        parity_log.record("layered_unet_group_widths", f"uid{uid}_levels{len(dm)}", None, err, bound=2e-5)
        assert err < 2e-5, (uid, dm, err)
        assert torch.equal(unet(x[:200].contiguous().cuda(), 11), big[:200])
        assert torch.equal(unet(x[600:608].contiguous().cuda(), 11), big[600:608])


def test_layered_unet_widest_network_widest_slices_vs_oracle():
    """unet_input_dim 64 with four levels: 512-channel layers (GroupNorm groups of 64 channels: the statistics of a group span two waves of
    the matrix-pipe kernel), four input chunks at the deepest up block, and -- at 3100 trajectories -- 128-column slices with several items
    per workgroup.  Against the oracle's forward on six trajectories, and bitwise against the same trajectories in a batch of six."""
    from oracle import mmd_oracle as O
    uid, dm = 64, (1, 2, 4, 8)
    unet = _unet(uid, dm)
    sd = O.state_dict_to_torch(synth.synth_unet_state_dict(0, unet_input_dim=uid, dim_mults=dm))
    x = torch.from_numpy(synth.synth_noise(906, (3100, H, D))) * 0.7
    big = unet(x.cuda(), 11)
    assert torch.isfinite(big).all()
    ref = O.unet_forward(sd, x[1000:1006], torch.full((6,), 11, dtype=torch.long))
    err = rel_l2(big[1000:1006].cpu(), ref)
    parity_log.record("layered_unet_widest", "uid64_levels4_n3100", None, err, bound=2e-5)
    assert err < 2e-5, err
    assert torch.equal(unet(x[1000:1006].contiguous().cuda(), 11), big[1000:1006])


def test_one_launch_residual_blocks_equal_the_two_launch_form():
    """mconv_kernel KIND 4 runs a ResidualTemporalBlock of <= 64 channels as ONE launch (the hidden tensor becomes the second conv's LDS slab
    instead of a tensor in memory); the arithmetic is that of the two Conv1dBlock launches, so eps must be bitwise the same.
    mmd_unet_options.rtb_fused is a creation-time option of the device model (TemporalUnet(rtb_fused=...)); the vector-ALU form of
    every layer (MMD_UNET_LAYERED_VALU) is held against the matrix-pipe form here too, within the fp32 reference's own distance."""
    x = (torch.from_numpy(synth.synth_noise(907, (1300, H, D))) * 0.7).cuda()
    for dm in ((1, 2, 4, 8), (1, 2, 4)):
        outs = []
        for fused, T in ((0, 25), (64, 26), (128, 27)):
            u = _unet(32, dm, layered=True, rtb_fused=fused)
            u.handle(T, "cuda")
            outs.append(u(x, 7))
            outs.append(u(x[:5].contiguous(), 7))
        assert torch.isfinite(outs[0]).all()
        assert torch.equal(outs[0], outs[2]) and torch.equal(outs[0], outs[4])
        assert torch.equal(outs[1], outs[3]) and torch.equal(outs[1], outs[5]) and torch.equal(outs[1], outs[0][:5])
        valu = _unet(32, dm, layered=True, layered_valu=True)
        narrow = _unet(32, dm, layered=True, mconv_max_cs=32)
        assert rel_l2(valu(x, 7).cpu(), outs[0].cpu()) < 6e-6
        assert rel_l2(narrow(x, 7).cpu(), outs[0].cpu()) < 6e-6


def test_option1_sampling_loop_captures_into_a_hip_graph():
    """The layer-by-layer path inside a stream capture: the 26-step guided loop of an option-1 model (39 launches per forward, their grids
    sized from a cached occupancy query) as ONE hipGraph, replayed, against the eager loop."""
    import gpu_common
    from mmd_amd import _lib, ops
    from mmd_amd.diffusion_model import ddpm_sample_fn
    T, B, R = 25, 8, 2
    model = _model(T)
    starts, goals = synth.start_goal_circle(6, 0.8)
    paths = synth.straight_line_paths(starts, goals, H)
    guide = gpu_common.hip_guide("EnvHighways2D", [[cases.soft_group(paths, r)] for r in range(R)], n_robots=R)
    hc = {0: torch.stack([cases.hard_conds_for(starts[r], goals[r])[0] for r in range(R)]),
          H - 1: torch.stack([cases.hard_conds_for(starts[r], goals[r])[H - 1] for r in range(R)])}
    hard = torch.stack([hc[0], hc[H - 1]], dim=1).cuda().contiguous()
    tm, tg = ops.register(model), ops.register(guide)
    sg = _lib.signed64(_lib.HARD_ROWS_START_GOAL)
    ref = model.run_inference(None, hc, n_samples=B, n_robots=R, horizon=H, return_chain=True, sample_fn=ddpm_sample_fn,
                              guide=guide, n_guide_steps=20, t_start_guide=13, noise_std_extra_schedule_fn=lambda t: 0.5,
                              n_diffusion_steps_without_noise=1, seed=77)
    xg = torch.empty((R * B, H, D), device="cuda")
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        cg = torch.ops.mmd_amd.p_sample_loop(xg, hard, sg, tm, tg, R, T, 1, True, None, 77, 20, 13, 0.5, 0, True)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.isfinite(ref).all() and torch.equal(cg, ref) and torch.equal(xg, ref[-1])
