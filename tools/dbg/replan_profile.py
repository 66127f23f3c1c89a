"""cProfile of the host side of ONE re-plan call (an MPD call from an experience under 1 hard + 567 soft constraint points): where the
host time of a 1.2 ms call goes.  Usage: replan_profile.py [n_calls]"""
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmd_amd import synth
from mmd_amd.constraints import MultiPointConstraint
from mmd_amd.planners import MPD, PathBatchExperience

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
H, N, r = 64, 10, 3
starts, goals = synth.start_goal_circle(N, 0.45)
paths = torch.from_numpy(synth.straight_line_paths(starts, goals, H)).cuda()
p = MPD(model_id="EnvHighways2D-RobotPlanarDisk", planner_alg="mmd", start_state_pos=torch.from_numpy(starts[r]),
        goal_state_pos=torch.from_numpy(goals[r]), n_samples=64, device="cuda", model_state_dict=synth.synth_unet_state_dict(0),
        model_args=dict(n_diffusion_steps=25), trained_models_dir="")
s, g = torch.from_numpy(starts[r]), torch.from_numpy(goals[r])
# as cbs.py:468-508 builds them: one tiny DEVICE tensor per point
soft = MultiPointConstraint(q_l=[paths[j, t] for j in range(N) if j != r for t in range(1, H)],
                            t_range_l=[(t, t + 1) for j in range(N) if j != r for t in range(1, H)], is_soft=True)
hard = MultiPointConstraint(q_l=[paths[r, 30]], t_range_l=[(28, 33)])
first = p(s, g)
exp = PathBatchExperience(first.trajs_final)
for _ in range(5):
    p(s, g, [hard, soft], exp)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n_calls):
    p(s, g, [hard, soft], exp)
print(f"{n_calls} re-plan calls: {1e3 * (time.perf_counter() - t0) / n_calls:.3f} ms per call")
pr = cProfile.Profile()
pr.enable()
for _ in range(n_calls):
    p(s, g, [hard, soft], exp)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(35)
