"""The guided-step kernel at the reference's call size (64 trajectories = one planner call): time of 20 guide iterations by term --
map with obstacles (SDF gather) or without, 0 / 9 / 31 constraint slots.  Usage: guide_small.py [n_traj ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import cases
import gpu_common
from mmd_amd import synth

H = 64
ns = [int(a) for a in sys.argv[1:]] or [64, 256]
for n_others in (0, 9, 31):
    N = n_others + 1
    starts, goals = synth.start_goal_circle(max(N, 2), 0.45)
    paths = synth.straight_line_paths(starts, goals, H)
    for env in ("EnvEmpty2D", "EnvHighways2D"):
        groups = [[cases.soft_group(paths, 0)]] if n_others else [[]]
        guide = gpu_common.hip_guide(env, groups)
        hc = cases.hard_conds_for(starts[0], goals[0])
        hard = torch.stack([hc[0], hc[H - 1]])[None].cuda().contiguous()
        for n in ns:
            x = (torch.from_numpy(synth.synth_noise(7, (n, H, 4))) * 0.3).cuda()
            for steps in (20, 200):
                y = x.clone()
                for _ in range(3):
                    guide.guide_steps(y, hard, (1 << 0) | (1 << (H - 1)), steps)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    guide.guide_steps(y, hard, (1 << 0) | (1 << (H - 1)), steps)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / 20 * 1e3
                print(f"{env:14s} {n_others:2d} slots  n={n:4d}  {steps:3d} guide iterations: {us:7.1f} us per launch = {us / steps:5.2f} us per iteration")
