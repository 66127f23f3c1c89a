"""Wall time of ONE MPD planner call as CBS / PP issue them (mpd.py:274-460: constraints -> guided sampling -> post-processing ->
PlannerOutput), split into host enqueue and device time.  Usage: planner_time.py [n_samples] [T]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmd_amd import synth
from mmd_amd.constraints import MultiPointConstraint
from mmd_amd.planners import MPD

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 25
H = 64
starts, goals = synth.start_goal_circle(10, 0.45)
paths = synth.straight_line_paths(starts, goals, H)
p = MPD(model_id="EnvHighways2D-RobotPlanarDisk", planner_alg="mmd", start_state_pos=torch.from_numpy(starts[3]),
        goal_state_pos=torch.from_numpy(goals[3]), n_samples=B, device="cuda", model_state_dict=synth.synth_unet_state_dict(0),
        model_args=dict(n_diffusion_steps=T), trained_models_dir="")
soft = MultiPointConstraint(q_l=[torch.from_numpy(paths[j, t]) for j in range(10) if j != 3 for t in range(1, H)],
                            t_range_l=[(t, t + 1) for j in range(10) if j != 3 for t in range(1, H)])
soft.is_soft = True
hard = MultiPointConstraint(q_l=[torch.tensor([0.1, 0.2])], t_range_l=[(20, 27)])
for cons, name in ((None, "no constraints"), ([soft, hard], "567 soft + 1 hard constraint points")):
    for _ in range(3):
        p(torch.from_numpy(starts[3]), torch.from_numpy(goals[3]), constraints_l=cons)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        out = p(torch.from_numpy(starts[3]), torch.from_numpy(goals[3]), constraints_l=cons)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ts.append((t1 - t0, time.perf_counter() - t0))
    ts.sort(key=lambda v: v[1])
    print(f"B={B} T={T} {name}: call returns after {1e3 * ts[5][0]:.2f} ms, device idle after {1e3 * ts[5][1]:.2f} ms (median of 10); "
          f"t_total reported {1e3 * out.t_total:.2f} ms")
