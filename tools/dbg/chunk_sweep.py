"""Round time of a guided sampling round by batch size and stream-chunk count (mmd_sampler_desc.n_streams): where does splitting the
robots into concurrent chunks pay?  Usage: chunk_sweep.py [robots ...]   (x 64 samples each, Empty map, all-pairs soft constraints, T = 100)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmd_amd import synth
from mmd_amd.diffusion_model import GaussianDiffusionModel
from mmd_amd.multi_robot import MultiRobotSampler
from mmd_amd.temporal_unet import TemporalUnet

H, T, B = 64, 100, 64
unet = TemporalUnet()
unet.load_state_dict(synth.synth_unet_state_dict(0))
model = GaussianDiffusionModel(model=unet, variance_schedule="exponential", n_diffusion_steps=T, predict_epsilon=True)
robots = [int(v) for v in sys.argv[1:]] or [6, 8, 9, 10, 12, 16, 20, 24, 28, 32]
for N in robots:
    starts, goals = synth.start_goal_circle(N, 0.8)
    row = []
    for ns in (1, 2, 3, 4):
        if ns > N:
            continue
        s = MultiRobotSampler(model, starts, goals, env_id="EnvEmpty2D", n_samples=B, device="cuda", n_streams=ns)
        paths = torch.from_numpy(synth.straight_line_paths(starts, goals, H)).cuda()
        s.set_other_paths(paths)
        for k in range(2):
            s.sample(seed=100 + k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(5):
            s.sample(seed=200 + k)
        torch.cuda.synchronize()
        row.append(f"{ns} chunk(s) {(time.perf_counter() - t0) / 5 * 1e3:6.2f} ms")
    print(f"{N:3d} robots = {N * B:5d} trajectories: " + "   ".join(row), flush=True)
