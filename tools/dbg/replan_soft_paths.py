"""The ECBS re-plan (experience, 3 + 3 steps, 1 hard constraint + the soft constraints from 9 other agents' paths) with the soft constraints as
the reference's list of 567 tiny tensors against MPD.__call__(..., soft_paths=(paths_all, agent)).  Usage: replan_soft_paths.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmd_amd import synth
from mmd_amd.constraints import MultiPointConstraint
from mmd_amd.planners import MPD, PathBatchExperience

H, N, r = 64, 10, 3
starts, goals = synth.start_goal_circle(N, 0.45)
paths = torch.from_numpy(synth.straight_line_paths(starts, goals, H)).cuda()
p = MPD(model_id="EnvHighways2D-RobotPlanarDisk", planner_alg="mmd", start_state_pos=torch.from_numpy(starts[r]),
        goal_state_pos=torch.from_numpy(goals[r]), n_samples=64, device="cuda", model_state_dict=synth.synth_unet_state_dict(0),
        model_args=dict(n_diffusion_steps=25), trained_models_dir="")
s, g = torch.from_numpy(starts[r]), torch.from_numpy(goals[r])
soft = MultiPointConstraint(q_l=[paths[j, t] for j in range(N) if j != r for t in range(1, H)],
                            t_range_l=[(t, t + 1) for j in range(N) if j != r for t in range(1, H)], is_soft=True)
hard = MultiPointConstraint(q_l=[paths[r, 30]], t_range_l=[(28, 33)])
exp = PathBatchExperience(p(s, g).trajs_final)
for name, fn in (("constraint list (567 tensors)", lambda: p(s, g, [hard, soft], exp)), ("soft_paths tensor", lambda: p(s, g, [hard], exp, soft_paths=(paths, r)))):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    print(f"re-plan from an experience, 1 hard + 567 soft constraint points, soft constraints as {name}: {1e3 * sorted(ts)[15]:.3f} ms per call")
