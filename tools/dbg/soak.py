"""Soak: many planner calls of every kind in one process -- device memory, host RSS and per-call time at the start and at the end."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import psutil
import torch
from mmd_amd import synth
from mmd_amd.constraints import MultiPointConstraint
from mmd_amd.planners import MPD, PathBatchExperience, plan_batched, plan_concurrently

H, N = 64, 10
starts, goals = synth.start_goal_circle(N, 0.45)
paths = torch.from_numpy(synth.straight_line_paths(starts, goals, H)).cuda()
sd = synth.synth_unet_state_dict(0)
ps = [MPD(model_id="EnvHighways2D-RobotPlanarDisk", planner_alg="mmd", start_state_pos=torch.from_numpy(starts[r]), goal_state_pos=torch.from_numpy(goals[r]),
          n_samples=64, device="cuda", model_state_dict=sd, model_args=dict(n_diffusion_steps=25), trained_models_dir="", seed=18 + r) for r in range(4)]
sg = [(torch.from_numpy(starts[r]), torch.from_numpy(goals[r])) for r in range(4)]


def cons(r):
    soft = MultiPointConstraint(q_l=[paths[j, t] for j in range(N) if j != r for t in range(1, H)],
                                t_range_l=[(t, t + 1) for j in range(N) if j != r for t in range(1, H)], is_soft=True)
    return [MultiPointConstraint(q_l=[paths[r, 30]], t_range_l=[(28, 33)]), soft]


proc = psutil.Process()
outs = plan_batched([(p, *sg[r]) for r, p in enumerate(ps)])
exps = [PathBatchExperience(o.trajs_final) for o in outs]


def cycle(k):
    r = k % 4
    ps[r](*sg[r], cons(r), exps[r])                                            # a re-plan, constraint list
    ps[r](*sg[r], cons(r)[:1], exps[r], soft_paths=(paths, r))                 # a re-plan, soft paths tensor
    if k % 10 == 0:
        ps[r](*sg[r], cons(r))                                                 # a fresh plan
        plan_batched([(p, *sg[j], cons(j), exps[j]) for j, p in enumerate(ps)])
        plan_concurrently([(p, *sg[j], cons(j)) for j, p in enumerate(ps[:2])])


def snapshot(tag, dt):
    torch.cuda.synchronize()
    print(f"{tag}: device allocated {torch.cuda.memory_allocated() / 2**20:.1f} MiB reserved {torch.cuda.memory_reserved() / 2**20:.1f} MiB  "
          f"host RSS {proc.memory_info().rss / 2**20:.0f} MiB  {1e3 * dt:.2f} ms per cycle", flush=True)


n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for k in range(50):
    cycle(k)
t0 = time.perf_counter()
for k in range(200):
    cycle(k)
snapshot("after 250 cycles", (time.perf_counter() - t0) / 200)
done = 250
while done < n:
    for k in range(800):
        cycle(k)
    t0 = time.perf_counter()
    for k in range(200):
        cycle(k)
    done += 1000
    snapshot(f"after {done} cycles", (time.perf_counter() - t0) / 200)
