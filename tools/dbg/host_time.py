import sys, os, time
sys.path.insert(0, '.')
import torch
from mmd_amd import synth
from mmd_amd.diffusion_model import GaussianDiffusionModel
from mmd_amd.multi_robot import MultiRobotSampler
from mmd_amd.temporal_unet import TemporalUnet
T, B, R = 100, 64, 32
unet = TemporalUnet(); unet.load_state_dict(synth.synth_unet_state_dict(0))
model = GaussianDiffusionModel(model=unet, n_diffusion_steps=T, predict_epsilon=True)
starts, goals = synth.start_goal_circle(R, 0.8)
s = MultiRobotSampler(model, starts, goals, n_samples=B)
paths = torch.from_numpy(synth.straight_line_paths(starts, goals, 64)).cuda()
s.set_other_paths(paths)
for ns in (1, 2, 1, 2):
    s.n_streams = ns
    s.sample(seed=1); torch.cuda.synchronize()
    t0 = time.perf_counter(); s.sample(seed=2); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"streams {ns}: host enqueue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms")
