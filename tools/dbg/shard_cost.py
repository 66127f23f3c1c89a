"""Per-GPU compute of rank 0 of a W-GPU weak-scaling job (32 robots per GPU of one 32W-robot instance), timed in ONE process:
the constraint table from all 32W paths + one guided sampling call for the 32 local robots.  No communication."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmd_amd import synth, _lib
if os.environ.get("MMD_AMD_LIB"):
    _lib.LIB_PATH = os.environ["MMD_AMD_LIB"]
from mmd_amd.diffusion_model import GaussianDiffusionModel
from mmd_amd.multi_robot import MultiRobotSampler
from mmd_amd.temporal_unet import TemporalUnet

H = 64
unet = TemporalUnet(state_dim=4, n_support_points=H, unet_input_dim=32, dim_mults=(1, 2, 4))
unet.load_state_dict(synth.synth_unet_state_dict(0))
model = GaussianDiffusionModel(model=unet, variance_schedule="exponential", n_diffusion_steps=100, predict_epsilon=True)
for W in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    n = 32 * W
    starts, goals = synth.start_goal_circle(n, 0.8)
    s = MultiRobotSampler(model, starts, goals, env_id="EnvEmpty2D", n_samples=64, rank=0, world_size=W, device="cuda")
    paths = torch.from_numpy(synth.straight_line_paths(starts, goals, H)).cuda()
    for _ in range(2):
        s.set_other_paths(paths); s.sample(seed=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(3):
        s.set_other_paths(paths); s.sample(seed=k)
    torch.cuda.synchronize()
    print(f"W={W}: {n} robots, {n - 1} constraint slots per robot: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per round on one GPU")
