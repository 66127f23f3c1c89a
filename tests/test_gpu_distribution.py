"""-m gpu: distribution-level parity of the guided sampler.  End to end the guided chain is chaotic (DESIGN.md
section 4), so beside the per-step teacher-forced tests the sampler's OUTPUT DISTRIBUTION is compared with the genuine
reference's: tests/golden/g15_distribution_*.npz hold the reference's final rows over 32 noise seeds (tools/make_golden.py::g15);
the HIP sampler runs the same seeds (same injected x_T and step noise) as ONE batch of 32 B samples and must reproduce, within
Z_MAX = 4 standard errors, the per-support-point mean and covariance of the positions, the free / collision split, the number of
trajectories that violate a constraint and the mean violation count.  ref: guides.py:180-226, tasks.py:236-311,
sample_functions.py:40-107."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cases                             # noqa: E402
import parity_log                        # noqa: E402
from cases import GOLDEN, H, D, rel_l2   # noqa: E402


@pytest.mark.parametrize("name", cases.DIST_CASES)
def test_guided_sampling_distribution_vs_reference(name):
    import gpu_common
    from mmd_amd import postprocess as post
    g = np.load(os.path.join(GOLDEN, f"g15_distribution_{name}.npz"))
    T, B, n_seeds, base = (int(v) for v in g["meta"])
    case = dict(cases.sample_case(name))
    assert case["T"] == T
    n = n_seeds * B
    case["B"] = n
    xT, steps = cases.distribution_inputs(T, B, n_seeds, base)
    final = gpu_common.hip_run_inference(case, xT, steps)[-1]                      # [n, H, D] normalised, on the GPU
    assert final.shape == (n, H, D) and torch.isfinite(final).all()
    ref = torch.from_numpy(g["finals"])
    pos_ref = cases.unnormalize(ref)[..., :2].numpy()
    pos_hip = cases.unnormalize(final.cpu())[..., :2].numpy()
    # (the stored statistics are those of the stored rows)
    m, c = cases.position_stats(pos_ref)
    assert np.allclose(m, g["pos_mean"], atol=1e-6) and np.allclose(c, g["pos_cov"], atol=1e-6)
    z_mean, z_cov = cases.distribution_z(pos_hip, pos_ref)
    # free / collision split through the product's own post-processing (bit-exact vs the reference's on equal inputs: g9 / g12)
    guide = gpu_common.hip_guide(case["map"], [case["cons"]])
    r = post.postprocess_batch(guide, cases.unnormalize(final.cpu()).cuda().contiguous(), n_robots=1, smooth=False)
    free_hip = int(r.free_mask.bool().sum())
    free_ref = int(g["free_mask"].sum())
    viol_hip = cases.violation_counts(pos_hip, case["cons"])
    viol_ref = g["violations"]
    assert np.array_equal(cases.violation_counts(pos_ref, case["cons"]), viol_ref)  # the test's counter == the generator's
    z_free = cases.proportion_z(free_hip, free_ref, n)
    z_viol = cases.proportion_z(int((viol_hip > 0).sum()), int((viol_ref > 0).sum()), n)
    z_pairs = cases.mean_z(viol_hip, viol_ref)
    # matched pairs (same noise): how many trajectories coincide with the reference's at the north-star tolerance
    per = np.array([rel_l2(final[i].cpu(), ref[i]) for i in range(n)])
    for key, val in (("z_mean", z_mean), ("z_cov", z_cov), ("z_free", z_free), ("z_violating", z_viol), ("z_violation_pairs", z_pairs)):
        parity_log.record("distribution_vs_reference", f"{name}_{key}", None, val, bound=cases.Z_MAX)
    parity_log.record("distribution_vs_reference", f"{name}_matched_within_1e-3", None, float((per < 1e-3).mean()),
                      note=f"free {free_hip} / {free_ref} of {n}; violating {int((viol_hip > 0).sum())} / {int((viol_ref > 0).sum())}; "
                           f"pairs {int(viol_hip.sum())} / {int(viol_ref.sum())}; median matched rel-L2 {np.median(per):.2e}")
    parity_log.track(f"end_to_end_matched_within_1e-3.random_weights.{name}", int((per < 1e-3).sum()), int(n),
                     note=f"random-init weights, {n} trajectories over {n_seeds} noise seeds; median rel-L2 {np.median(per):.2e} (the chain is "
                          "chaotic end to end on random weights: the distribution-level z-tests next to this are the parity statement)")
    assert z_mean < cases.Z_MAX and z_cov < cases.Z_MAX, (name, z_mean, z_cov)
    assert z_free < cases.Z_MAX, (name, free_hip, free_ref)
    assert z_viol < cases.Z_MAX and z_pairs < cases.Z_MAX, (name, z_viol, z_pairs)
