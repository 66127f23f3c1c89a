"""Wall time of the re-plans of ONE CBS expansion (cbs.py:390-458): the conflict's two agents re-planned independently, each from its
previous batch (xCBS / xECBS: experience = trajs_final, 3 noising + 3 denoising steps, mmd_params.py) under its hard constraint + the
soft constraints from the other agents' paths -- one call after the other as the reference loops, and packed by planners.plan_batched.
Usage: replan_time.py [n_samples] [n_agents_replanned ...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmd_amd import synth
from mmd_amd.constraints import MultiPointConstraint
from mmd_amd.planners import MPD, PathBatchExperience, plan_batched

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
RS = [int(v) for v in sys.argv[2:]] or [1, 2, 4]
H, N = 64, 10
starts, goals = synth.start_goal_circle(N, 0.45)
paths = synth.straight_line_paths(starts, goals, H)
sd = synth.synth_unet_state_dict(0)
ps = [MPD(model_id="EnvHighways2D-RobotPlanarDisk", planner_alg="mmd", start_state_pos=torch.from_numpy(starts[r]),
          goal_state_pos=torch.from_numpy(goals[r]), n_samples=B, device="cuda", model_state_dict=sd, model_args=dict(n_diffusion_steps=25),
          trained_models_dir="", seed=18 + r) for r in range(max(RS))]


def cons(r):
    soft = MultiPointConstraint(q_l=[torch.from_numpy(paths[j, t]) for j in range(N) if j != r for t in range(1, H)],
                                t_range_l=[(t, t + 1) for j in range(N) if j != r for t in range(1, H)], is_soft=True)
    return [MultiPointConstraint(q_l=[torch.from_numpy(paths[r, 30])], t_range_l=[(28, 33)]), soft]


first = plan_batched([(p, torch.from_numpy(starts[r]), torch.from_numpy(goals[r])) for r, p in enumerate(ps)])
for R in RS:
    calls = [(ps[r], torch.from_numpy(starts[r]), torch.from_numpy(goals[r]), cons(r), PathBatchExperience(first[r].trajs_final)) for r in range(R)]
    res = {}
    for name, fn in (("one after the other", lambda: [c[0](*c[1:]) for c in calls]), ("plan_batched", lambda: plan_batched(calls))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(15):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
            torch.cuda.synchronize()
        res[name] = sorted(ts)[len(ts) // 2]
    a, b = res["one after the other"], res["plan_batched"]
    print(f"B={B}: {R} re-plan(s) from an experience (3 + 3 steps, 1 hard + {(N - 1) * (H - 1)} soft constraint points each): one after the other "
          f"{1e3 * a:.2f} ms, plan_batched {1e3 * b:.2f} ms  ({R * B / a:.0f} -> {R * B / b:.0f} trajectories/s)")
