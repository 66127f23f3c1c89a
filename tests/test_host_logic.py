"""CPU: host-side logic of the product package (no device compute): schedule tables, SDF textures, normaliser,
constraint time-bucketing (the host half of the C ABI), parameter spec, synthetic-input determinism."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from mmd_amd import _lib, synth

SG = _lib.signed64(_lib.HARD_ROWS_START_GOAL)
from mmd_amd.constraints import CostConstraint
from mmd_amd.environments import sdf_grid_texture
from mmd_amd.normalization import LimitsNormalizer, TrajectoryDatasetFacade
from mmd_amd.schedules import SCHEDULE_KEYS, diffusion_buffers
from cases import GOLDEN, H, D


def test_schedule_tables_bit_exact_vs_reference():
    g = np.load(os.path.join(GOLDEN, "g1_schedules.npz"))
    for T in (25, 50, 100):
        tb = diffusion_buffers(T)
        for k in SCHEDULE_KEYS:
            assert np.array_equal(tb[k].numpy(), g[f"T{T}.{k}"]), (T, k)
    for T in (25, 100):                                  # variance_schedule='cosine' (helpers.py:16-27)
        tb = diffusion_buffers(T, "cosine")
        for k in SCHEDULE_KEYS:
            assert np.array_equal(tb[k].numpy(), g[f"cosine.T{T}.{k}"]), ("cosine", T, k)
    with pytest.raises(NotImplementedError):
        diffusion_buffers(25, "linear")


@pytest.mark.parametrize("env_id", ["EnvEmpty2D", "EnvHighways2D", "EnvConveyor2D", "EnvDropRegion2D"])
def test_sdf_texture_vs_reference(env_id):
    g = np.load(os.path.join(GOLDEN, "g3_sdf.npz"))
    tex = sdf_grid_texture(env_id)
    assert tex.shape == (400, 400, 4) and tex.dtype == np.float32
    idx = g[f"{env_id}.idx"]
    assert np.array_equal(tex[idx[:, 0], idx[:, 1], 0], g[f"{env_id}.sdf"])
    assert np.array_equal(tex[idx[:, 0], idx[:, 1], 1:3], g[f"{env_id}.grad"])
    assert not tex[..., 3].any()


def test_normalizer_roundtrip_and_clip():
    nz = LimitsNormalizer(synth.NORM_MINS, synth.NORM_MAXS)
    x = torch.from_numpy(synth.synth_noise(3, (5, H, 4))) * 0.3
    assert torch.allclose(nz.unnormalize(nz.normalize(x)), x, atol=1e-6)
    wide = torch.tensor([[1.5, -2.0, 0.0, 0.5]])
    assert torch.equal(nz.unnormalize(wide), torch.tensor([[1.0, -1.0, 0.0, 0.75]]))      # data-dependent clip fires
    ds = TrajectoryDatasetFacade(synth.NORM_MINS, synth.NORM_MAXS)
    hc = ds.get_hard_conditions(torch.tensor([[0.8, 0.0], [-0.8, 0.0]]), normalize=True)
    assert set(hc) == {0, H - 1} and torch.allclose(hc[0], torch.tensor([0.8, 0.0, 0.0, 0.0]))


def _pack(groups):
    lib = _lib.load()
    G = len(groups)
    n_pts = (C.c_int32 * G)(*[g.qs.shape[0] for g in groups])
    arrs = [[np.ascontiguousarray(getattr(g, a), dtype=np.float32) for g in groups] for a in ("qs", "traj_ranges", "radii")]
    ptrs = [(C.c_void_p * G)(*[a.ctypes.data for a in col]) for col in arrs]
    slots = (C.c_int32 * G)()
    _lib.check(lib.mmd_pack_constraints(G, n_pts, ptrs[0], ptrs[1], ptrs[2], H, None, 0, slots))
    total = int(sum(slots))
    ell = np.zeros((total, H, 4), np.float32)
    _lib.check(lib.mmd_pack_constraints(G, n_pts, ptrs[0], ptrs[1], ptrs[2], H, ell.ctypes.data, total, slots))
    return ell, list(slots)


def test_pack_constraints_time_buckets():
    soft = CostConstraint(None, H, q_l=[torch.tensor([0.1 * j, t / 64.0]) for j in range(3) for t in range(1, H)],
                          traj_range_l=[(t, t + 1) for j in range(3) for t in range(1, H)],
                          radius_l=[0.12] * (3 * 63), is_soft=True)
    hard = CostConstraint(None, H, q_l=[torch.tensor([0.5, 0.5]), torch.tensor([-0.5, 0.25])],
                          traj_range_l=[(20, 27), (25, 70)], radius_l=[0.12, 0.2])
    empty_range = CostConstraint(None, H, q_l=[torch.tensor([0.0, 0.0])], traj_range_l=[(5, 5)], radius_l=[0.1])
    ell, slots = _pack([soft, hard, empty_range])
    assert slots == [3, 2, 0]                                   # 3 robots per step; ranges overlap on t=25,26; [5,5) empty
    s = ell[:3]
    assert (s[:, 0, 2] < 0).all() and (s[:, 1:, 2] == np.float32(0.12)).all()          # no point at t=0
    assert np.allclose(sorted(s[:, 10, 0]), [0.0, 0.1, 0.2]) and np.allclose(s[:, 10, 1], 10 / 64.0)
    h = ell[3:5]
    active = (h[..., 2] >= 0).sum(0)
    assert list(active[18:30]) == [0, 0, 1, 1, 1, 1, 1, 2, 2, 1, 1, 1] and active[63] == 1   # [20,27) and [25,70) clipped to H
    # exclusive end (cost_functions.py:305): t=27 belongs only to the second point
    assert {tuple(np.round(r[:2], 3)) for r in h[:, 27] if r[2] >= 0} == {(-0.5, 0.25)}


def test_pack_constraints_slot_overflow_is_an_error():
    lib = _lib.load()
    g = CostConstraint(None, H, q_l=[torch.zeros(2)] * 2, traj_range_l=[(1, 5), (1, 5)], radius_l=[0.1, 0.1])
    n_pts = (C.c_int32 * 1)(2)
    arrs = [np.ascontiguousarray(a, np.float32) for a in (g.qs, g.traj_ranges, g.radii)]
    ptrs = [(C.c_void_p * 1)(a.ctypes.data) for a in arrs]
    ell = np.zeros((1, H, 4), np.float32)
    slots = (C.c_int32 * 1)()
    assert lib.mmd_pack_constraints(1, n_pts, ptrs[0], ptrs[1], ptrs[2], H, ell.ctypes.data, 1, slots) != 0
    assert b"slots" in lib.mmd_last_error()
    assert lib.mmd_pack_constraints(1, n_pts, ptrs[0], ptrs[1], ptrs[2], 32, ell.ctypes.data, 1, slots) != 0   # wrong H


def test_synthetic_inputs_are_deterministic():
    a, b = synth.synth_unet_state_dict(0), synth.synth_unet_state_dict(0)
    assert list(a) == list(b) and all(np.array_equal(a[k], b[k]) for k in a)
    assert sum(v.size for v in a.values()) == 997124                     # 997 124 parameters
    assert np.array_equal(synth.synth_noise(5, (2, 3)), synth.synth_noise(5, (2, 3)))
    s, g = synth.start_goal_circle(32)
    assert np.allclose(s, -g, atol=1e-6) and np.allclose(np.linalg.norm(s, axis=1), 0.8, atol=1e-6)


def test_no_cpu_fallback_when_library_missing(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_torch_library_ops_are_registered_with_fake_impls():
    """torch.ops.mmd_amd.* exist with the documented schemas and meta (fake) implementations (no GPU needed)."""
    import mmd_amd.ops  # noqa: F401
    for name in ("unet_forward", "guide_steps", "p_sample_loop", "ddim_sample"):
        assert hasattr(torch.ops.mmd_amd, name)
    schema = str(torch.ops.mmd_amd.p_sample_loop.default._schema)
    assert "Tensor(a0!) x" in schema and "Tensor? step_noise" in schema and "-> Tensor" in schema
    x, h = torch.empty(8, H, 4, device="meta"), torch.empty(1, 2, 4, device="meta")
    assert torch.ops.mmd_amd.p_sample_loop(x, h, SG, 0, 0, 1, 25, 1, False, None, 0, 20, 13, 0.5, 0, True).shape == (27, 8, H, 4)
    assert torch.ops.mmd_amd.ddim_sample(x, h, SG, 0, 0, 1, 100, False, 0, 50, 0, True).shape == (22, 8, H, 4)
    assert torch.ops.mmd_amd.unet_forward(x, 3, 25, 0).shape == x.shape
    with pytest.raises(RuntimeError):                       # CPU tensors: no kernel is registered for them
        torch.ops.mmd_amd.unet_forward(torch.zeros(1, H, 4), 0, 25, 0)


def test_philox_stream_ids_are_process_wide():
    """Sampling calls that draw their own noise take consecutive stream ids from ONE process-wide counter (the way every
    reference call advances torch's global RNG): two planners with the same base seed never reuse a stream."""
    from mmd_amd.diffusion_model import next_stream_seed
    a, b, c = next_stream_seed(18), next_stream_seed(18), next_stream_seed(19)
    assert b == a + 1 and c != a and c != b and (a >> 24) == 18 and (c >> 24) == 19


def test_bench_sharding_modes():
    """bench.py --scaling strong shards the metric's own 32-robot instance; weak keeps 32 robots per GPU; the Philox key
    base of a rank is its first global trajectory."""
    from mmd_amd.multi_robot import shard_range
    for world in (1, 2, 4, 8):
        r0, n_local = shard_range(32, world - 1, world)
        assert n_local == 32 // world and r0 + n_local == 32
        assert r0 * 64 == (32 - n_local) * 64                # traj_index_base of the last rank


def test_bench_power_sampler_reads_sysfs(tmp_path):
    """bench.py's PowerSampler: current pp_dpm_sclk level (the starred line, also the 'S:' sleep level), hwmon power1_input in
    microwatts, power1_cap; cards without the files are skipped, the busiest card is the one reported, the first skip_s
    seconds are dropped."""
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    for card, (sclk, uw) in {"card3": ("S: 95Mhz *\n0: 500Mhz\n1: 2400Mhz\n", 40_000_000),
                             "card8": ("S: 95Mhz\n0: 500Mhz\n1: 2055Mhz *\n", 1_300_000_000)}.items():
        hw = tmp_path / card / "device" / "hwmon" / "hwmon4"
        hw.mkdir(parents=True)
        (tmp_path / card / "device" / "pp_dpm_sclk").write_text(sclk)
        (hw / "power1_input").write_text(f"{uw}\n")
        (hw / "power1_cap").write_text("1400000000\n")
    (tmp_path / "card9" / "device").mkdir(parents=True)          # a render node without power files
    w = bench.PowerSampler(period=0.005, drm_root=str(tmp_path))
    assert len(w.cards) == 2
    w.start()
    time.sleep(0.1)
    out = w.finish(skip_s=0.02)
    assert out["sclk_mhz"] == 2055.0 and out["package_watts"] == 1300.0 and out["package_cap_watts"] == 1400.0
    assert out["samples"] >= 3 and "sysfs" in out["source"]
    assert bench.PowerSampler(drm_root=str(tmp_path / "card9")).cards == []      # nothing there: rocm-smi fallback


def test_split_cost_constraints_to_tasks_golden_g13():
    """MPDEnsemble.split_cost_constraints_to_tasks + the per-tile range / transform shift of run_constrained_inference
    (mpd_ensemble.py:431-507, 515-518) against the reference's own output (g13): tile order, hard-then-soft order inside a
    tile, every (q, range, radius, is_soft) table exactly -- a mixed list over 3 tiles with a range that starts on a tile
    boundary and one that straddles it.  The guides are stand-ins that record what add_extra_costs receives."""
    import types
    from mmd_amd.planners import MPDEnsemble
    g = np.load(os.path.join(GOLDEN, "g13_split_constraints.npz"))
    received = {k: [] for k in range(3)}

    def recorder(k):
        return types.SimpleNamespace(add_extra_costs=lambda costs, weights, k=k: received[k].extend(zip(costs, weights)))
    me = types.SimpleNamespace(robot=None, guides={k: recorder(k) for k in range(3)},
                               transforms={0: torch.tensor([0.0, 0.0]), 1: torch.tensor([2.0, 0.0]), 2: torch.tensor([4.0, 0.5])},
                               weight_grad_cost_constraints=2e-1, weight_grad_cost_soft_constraints=2e-2)
    me.infer_task_id_from_q_idx = types.MethodType(MPDEnsemble.infer_task_id_from_q_idx, me)
    me.split_cost_constraints_to_tasks = types.MethodType(MPDEnsemble.split_cost_constraints_to_tasks, me)

    def cc(qs, ranges, radii, soft):
        return CostConstraint(None, H, q_l=[torch.tensor(q, dtype=torch.float32) for q in qs], traj_range_l=ranges,
                              radius_l=radii, is_soft=soft)
    cons = [cc(([0.1, 0.2], [2.3, 0.1]), [(10, 14), (70, 75)], [0.12, 0.10], False),
            cc(([-0.4, 0.3], [1.7, -0.2], [4.4, 0.6]), [(5, 6), (64, 65), (130, 131)], [0.12, 0.12, 0.12], True),
            cc(([3.9, 0.4],), [(128, 140)], [0.2], False),
            cc(([0.9, -0.1], [0.0, 0.0]), [(62, 66), (0, 1)], [0.12, 0.15], True)]
    split = MPDEnsemble.split_cost_constraints_to_tasks(me, cons)
    assert list(split.keys()) == g["task_order"].tolist()
    MPDEnsemble._add_constraints(me, cons)
    for k in g["task_order"].tolist():
        assert len(received[k]) == int(g[f"n_{k}"])
        for j, (c, w) in enumerate(received[k]):
            assert bool(c.is_soft) == bool(g[f"soft_{k}_{j}"])
            assert w == (2e-2 if c.is_soft else 2e-1)
            assert np.array_equal(np.asarray(c.qs, dtype=np.float32), g[f"qs_{k}_{j}"]), (k, j, c.qs)
            assert np.array_equal(np.asarray(c.traj_ranges, dtype=np.float32), g[f"ranges_{k}_{j}"]), (k, j, c.traj_ranges)
            assert np.array_equal(np.asarray(c.radii, dtype=np.float32), g[f"radii_{k}_{j}"]), (k, j)



def test_launch_helper_rejects_host_tensors():
    """_lib.launch pins a call to the device of its tensors: a host tensor has no such device -- loud error, no
    launch."""
    from mmd_amd import _lib
    with pytest.raises(ValueError):
        _lib.launch("mmd_unet_forward", torch.zeros(1, 64, 4))


def test_temporal_unet_accepts_doubling_ladders_only():
    """UNET_DIM_MULTS (mmd/models/__init__.py:8-11) holds (1, 2, 4) and (1, 2, 4, 8); anything that is not a prefix of the doubling
    ladder, or a width the kernels are not written for, is refused at construction -- not at the first forward."""
    from mmd_amd.temporal_unet import TemporalUnet
    from mmd_amd.unet_spec import UNET_DIM_MULTS
    for dm in UNET_DIM_MULTS.values():
        assert TemporalUnet(dim_mults=dm).dim_mults == tuple(dm)
    for bad in ((1, 3), (2, 4), (1, 2, 4, 8, 16), ()):
        with pytest.raises(NotImplementedError):
            TemporalUnet(dim_mults=bad)
    with pytest.raises(NotImplementedError):
        TemporalUnet(unet_input_dim=20)


def test_constraint_points_cross_to_the_host_in_one_copy():
    """CostConstraint takes CBS's list of tiny tensors (cbs.py:468-508): stacked once (no per-point host copies), first two entries of
    every point, float32; mixed lists fall back to the element-wise path with the same result."""
    from mmd_amd.constraints import CostConstraint, _points_xy
    rng = np.random.Generator(np.random.PCG64(3))
    pts = rng.standard_normal((37, 4)).astype(np.float32)
    as_tensors = [torch.from_numpy(p.copy()).double() for p in pts]
    want = pts[:, :2]
    for q_l in (as_tensors, torch.from_numpy(pts), [p for p in pts], [as_tensors[0], pts[1]] + [list(p) for p in pts[2:]],
                [t[None] for t in as_tensors]):
        got = _points_xy(q_l)
        assert got.dtype == np.float32 and got.shape == (37, 2) and got.flags["C_CONTIGUOUS"] and np.array_equal(got, want)
    c = CostConstraint(None, 64, q_l=as_tensors, traj_range_l=[(3, 5)] * 37, radius_l=[0.12] * 37, is_soft=True)
    assert np.array_equal(c.qs, want) and c.traj_ranges.shape == (37, 2) and c.radii.shape == (37,)


def test_stream_seeds_are_unique_across_host_threads():
    """planners.plan_concurrently runs planner calls on several host threads; the global draw counter behind next_stream_seed must hand
    every caller its own value (it is the stand-in for torch's one global RNG stream in the reference)."""
    from concurrent.futures import ThreadPoolExecutor
    from mmd_amd import diffusion_model as dm
    start = dm._GLOBAL_DRAWS
    with ThreadPoolExecutor(max_workers=8) as pool:
        seeds = list(pool.map(lambda _: dm.next_stream_seed(18), range(4000)))
    assert len(set(seeds)) == 4000 and dm._GLOBAL_DRAWS == start + 4000
    assert {s >> 24 for s in seeds} == {18}


@pytest.mark.parametrize("direction", ("fwd", "rev"))
def test_split_cost_constraints_to_tasks_golden_g20(direction):
    """The 3-tile corner-turning instance of golden g20: MPDEnsemble's own routing (split_cost_constraints_to_tasks + the tile shift of
    _add_constraints, mpd_ensemble.py:431-522) of the instance's hard + soft MultiPointConstraints, built the way __call__ builds them,
    against the tables the genuine MPDEnsemble produced -- tile order, hard-then-soft order inside a tile, the shifted (q, range,
    radius, is_soft) rows (a soft range that straddles a tile boundary stays with the tile of its first index), for x AND y tile
    offsets in both directions."""
    import types
    from mmd_amd import synth
    from mmd_amd.constraints import MultiPointConstraint
    from mmd_amd.planners import MPDEnsemble
    g = np.load(os.path.join(GOLDEN, "g20_ensemble3.npz"))
    case = synth.ensemble3_case(direction)
    K = len(case["env_ids"])
    received = {k: [] for k in range(K)}

    def recorder(k):
        return types.SimpleNamespace(add_extra_costs=lambda costs, weights, k=k: received[k].extend(zip(costs, weights)))
    me = types.SimpleNamespace(robot=None, guides={k: recorder(k) for k in range(K)},
                               transforms={k: torch.from_numpy(case["transforms"][k]) for k in range(K)},
                               weight_grad_cost_constraints=2e-1, weight_grad_cost_soft_constraints=2e-2)
    me.infer_task_id_from_q_idx = types.MethodType(MPDEnsemble.infer_task_id_from_q_idx, me)
    me.split_cost_constraints_to_tasks = types.MethodType(MPDEnsemble.split_cost_constraints_to_tasks, me)
    cl = [MultiPointConstraint(q_l=[torch.from_numpy(q) for q in qs], t_range_l=[tuple(int(v) for v in t) for t in tr],
                               radius_l=[float(r) for r in rad], is_soft=soft) for (qs, tr, rad, soft) in case["constraints"]]
    cons = [CostConstraint(None, H, q_l=c.get_q_l(), traj_range_l=c.get_t_range_l(), radius_l=c.radius_l, is_soft=c.is_soft) for c in cl]
    split = MPDEnsemble.split_cost_constraints_to_tasks(me, cons)
    assert list(split.keys()) == g[f"{direction}.task_order"].tolist()
    MPDEnsemble._add_constraints(me, cons)
    n_total = 0
    for k in g[f"{direction}.task_order"].tolist():
        assert len(received[k]) == int(g[f"{direction}.n_{k}"])
        for j, (c, w) in enumerate(received[k]):
            n_total += 1
            assert bool(c.is_soft) == bool(g[f"{direction}.soft_{k}_{j}"]) and w == (2e-2 if c.is_soft else 2e-1)
            assert np.array_equal(np.asarray(c.qs, dtype=np.float32), g[f"{direction}.qs_{k}_{j}"]), (k, j, c.qs)
            assert np.array_equal(np.asarray(c.traj_ranges, dtype=np.float32), g[f"{direction}.ranges_{k}_{j}"]), (k, j)
            assert np.array_equal(np.asarray(c.radii, dtype=np.float32), g[f"{direction}.radii_{k}_{j}"]), (k, j)
    assert n_total == 5 and all(len(received[k]) >= 1 for k in range(K))


def test_ensemble_ddim_fails_the_way_the_reference_does():
    """DiffusionsEnsemble.ddim_sample (diffusion_ensemble.py:109-221) cannot run in the reference: run on the genuine class in the build
    container it raised AttributeError("'DiffusionsEnsemble' object has no attribute 'betas'"), and joint_conditional_sampling(ddim=True)
    / run_inference(ddim=True) raised TypeError("DiffusionsEnsemble.ddim_sample() got multiple values for argument
    'n_diffusion_steps'").  The mirror raises the same exception types (it must not silently sample something else)."""
    import types
    from mmd_amd.diffusion_ensemble import DiffusionsEnsemble
    m = types.SimpleNamespace(n_diffusion_steps=25, predict_epsilon=True, state_dim=4)
    ens = DiffusionsEnsemble({0: m, 1: m}, {0: torch.tensor([0.0, 0.0]), 1: torch.tensor([2.0, 0.0])})
    with pytest.raises(AttributeError, match="betas"):
        ens.ddim_sample((4, H, 4), {0: {}, 1: {}}, n_diffusion_steps=25)
    with pytest.raises(TypeError, match="n_diffusion_steps"):
        ens.joint_conditional_sampling({0: {}, 1: {}}, {(0, 1): (H - 1, 0)}, n_diffusion_steps=25, batch_size=4, ddim=True)
    with pytest.raises(TypeError, match="n_diffusion_steps"):
        ens.run_inference(None, {0: {}, 1: {}}, cross_conds={(0, 1): (H - 1, 0)}, n_samples=4, ddim=True, sample_kwargs={0: {}, 1: {}})
    with pytest.raises(ValueError):
        ens.joint_conditional_sampling({0: {}, 1: {}}, {}, n_diffusion_steps=None)


def test_empty_constraint_is_refused_like_the_reference():
    """CostConstraint(q_l=[]) fails in the reference (torch.stack of an empty list, cost_functions.py:293: RuntimeError)."""
    with pytest.raises(RuntimeError):
        CostConstraint(None, H, q_l=[], traj_range_l=[], radius_l=[])


def test_pack_constraints_randomized_vs_oracle_slot_table():
    """mmd_pack_constraints against the oracle's slot_table (the layout both sides define: a point covers the integer t with
    ceil(t0) <= t < ceil(t1), its slot = its rank among the points covering t, in list order) on 40 random groups: overlapping, empty,
    reversed, fractional and out-of-horizon ranges (negative starts, ends beyond H), 1 to 150 points, per-point radii; a group whose only range lies beyond the horizon owns no slot."""
    from oracle import mmd_oracle as O
    rng = np.random.Generator(np.random.PCG64(77))
    groups = []
    for k in range(40):
        n = int(rng.integers(1, 150))
        q = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
        t0 = rng.uniform(-6, 70, n)
        ln = rng.choice([0.0, 0.5, 1.0, 3.0, 9.5, 40.0, -2.0], n) if n else np.zeros(0)
        if k % 3 == 0:
            t0, ln = np.floor(t0), np.round(ln)
        tr = np.stack([t0, t0 + ln], 1).astype(np.float32).reshape(n, 2)
        r = rng.uniform(0.05, 0.3, n).astype(np.float32)
        groups.append(CostConstraint(None, H, q_l=torch.from_numpy(q), traj_range_l=tr, radius_l=r))
    groups = groups + [CostConstraint(None, H, q_l=np.zeros((1, 2), np.float32), traj_range_l=[(70, 90)], radius_l=[0.1])]
    ell, slots = _pack(groups)
    off = 0
    for g, S in zip(groups, slots):
        grp = O.ConstraintGroup(q=torch.from_numpy(g.qs), t_range=torch.from_numpy(g.traj_ranges), radius=torch.from_numpy(g.radii), weight=1.0)
        tab = O.slot_table(grp).numpy()                                  # [S', H] point index or -1
        S_ref = int((tab >= 0).any(1).sum())
        assert S == S_ref, (S, S_ref)
        blk = ell[off:off + S]
        for s in range(S):
            for t in range(H):
                p = tab[s, t]
                if p < 0:
                    assert blk[s, t, 2] < 0 and blk[s, t, 3] < 0, (s, t)
                else:
                    want = (g.qs[p, 0], g.qs[p, 1], g.radii[p], np.float32(g.radii[p] * abs(g.radii[p])))
                    assert tuple(blk[s, t]) == want, (s, t, tuple(blk[s, t]), want)
        off += S
    assert off == ell.shape[0] or (off == 0 and ell.shape[0] == 0)


def test_planner_call_host_shortcuts_keep_the_reference_semantics():
    """The host-side shortcuts of a planner call (profiles/r06_replan_call.txt) against the forms they replace: the (t0, t1) list
    conversion, the start / goal check (torch.allclose, mpd.py:318-321) and the cached hard-condition tensor (apply_hard_conditioning's
    dict, sample_functions.py:8-14) -- same values, and an in-place change of a stored state is seen."""
    from mmd_amd.constraints import _ranges
    from mmd_amd.diffusion_model import GaussianDiffusionModel
    from mmd_amd.planners import _same_state
    for tr in ([(t, t + 1) for t in range(9)], [(1.5, 2), (3, 4.25)], [[1, 2], [3, 4]], np.array([[1, 2], [3, 4]]),
               [(torch.tensor(1), torch.tensor(2))], torch.tensor([[0.0, 5.0]])):
        assert np.array_equal(_ranges(tr), np.asarray(tr, dtype=np.float32).reshape(-1, 2))
        assert _ranges(tr).dtype == np.float32
    assert _ranges([]).shape == (0, 2)
    stored = torch.tensor([0.3, -0.7])
    for given in (stored.clone(), stored.double(), stored + 5e-6, stored + 1e-3, [0.3, -0.7], np.array([0.3, -0.7]),
                  torch.tensor([[0.3, -0.7]]), torch.tensor([0.3])):
        want = bool(torch.allclose(torch.as_tensor(given).cpu().float(), stored))
        assert _same_state(given, stored) == want, given
    start, goal = torch.tensor([0.1, 0.2, 0.0, 0.0]), torch.tensor([0.5, -0.2, 0.0, 0.0])
    hc = {0: start, H - 1: goal}
    a, mask = GaussianDiffusionModel._hard_tensor(hc, 2, H, "cpu", D)
    assert mask == (1 | 1 << (H - 1)) and torch.equal(a, torch.stack([torch.stack([start, goal])] * 2))
    b, _ = GaussianDiffusionModel._hard_tensor(dict(hc), 2, H, "cpu", D)            # a copy of the dict, the same tensor objects: the cached one
    assert b is a
    c, _ = GaussianDiffusionModel._hard_tensor(hc, 3, H, "cpu", D)
    assert c.shape == (3, 2, D)
    goal[0] = 0.25                                                                  # in place: the version counter moves, the cache misses
    d, _ = GaussianDiffusionModel._hard_tensor(hc, 2, H, "cpu", D)
    assert d is not a and float(d[1, 1, 0]) == 0.25 and float(a[1, 1, 0]) == 0.5
    e, mask_e = GaussianDiffusionModel._hard_tensor({5: [0.0, 1.0, 2.0, 3.0]}, 1, H, "cpu", D)       # not tensors: built, not cached
    assert mask_e == 1 << 5 and e.shape == (1, 1, D)
    with pytest.raises(ValueError):
        GaussianDiffusionModel._hard_tensor({H: start}, 1, H, "cpu", D)
