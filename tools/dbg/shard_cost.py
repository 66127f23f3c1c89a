"""Per-GPU compute of rank 0 of a W-GPU job, timed in ONE process on one GPU (no communication): the constraint table
from all paths + one guided sampling call (T = 100 + 1) for the local robots + the device-side best-path pick.
  strong: the metric's 32-robot instance, 32 / W robots (2048 / W trajectories) per GPU
  weak:   32 robots per GPU of one 32 W-robot instance (the pairwise term grows with W)
Usage: python tools/dbg/shard_cost.py [strong|weak] [W ...]"""
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
args = sys.argv[1:]
mode = args.pop(0) if args and args[0] in ("strong", "weak") else "strong"
unet = TemporalUnet(state_dim=4, n_support_points=H, unet_input_dim=32, dim_mults=(1, 2, 4))
unet.load_state_dict(synth.synth_unet_state_dict(0))
model = GaussianDiffusionModel(model=unet, variance_schedule="exponential", n_diffusion_steps=100, predict_epsilon=True)
for W in [int(a) for a in args] or [1, 2, 4, 8]:
    n = 32 if mode == "strong" else 32 * W
    starts, goals = synth.start_goal_circle(n, 0.8)
    s = MultiRobotSampler(model, starts, goals, env_id="EnvEmpty2D", n_samples=64, rank=0, world_size=W, device="cuda")
    paths = torch.from_numpy(synth.straight_line_paths(starts, goals, H)).cuda()
    for _ in range(2):
        s.set_other_paths(paths); tr = s.sample(seed=1); s.best_paths(tr, paths)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(3):
        s.set_other_paths(paths); tr = s.sample(seed=k); s.best_paths(tr, paths)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print(f"{mode} W={W}: {n} robots, {s.n_local} robots = {s.n_local * 64} trajectories on this GPU, {n - 1} constraint slots "
          f"per robot: {ms:.1f} ms per round -> {n * 64 / ms * 1e3:.0f} trajectories/s if every rank takes as long", flush=True)
