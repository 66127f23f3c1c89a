"""Per-GPU round time of a W-GPU strong-scaling shard over many rounds, with the shader clock and socket power sampled
(bench.PowerSampler): does a GPU that is mostly idle between small launches run at full clock?
Usage: python tools/dbg/shard_clock.py [W ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from mmd_amd import synth
from mmd_amd.diffusion_model import GaussianDiffusionModel
from mmd_amd.multi_robot import MultiRobotSampler
from mmd_amd.temporal_unet import TemporalUnet

H = 64
unet = TemporalUnet(state_dim=4, n_support_points=H, unet_input_dim=32, dim_mults=(1, 2, 4))
unet.load_state_dict(synth.synth_unet_state_dict(0))
model = GaussianDiffusionModel(model=unet, variance_schedule="exponential", n_diffusion_steps=100, predict_epsilon=True)
for W in [int(a) for a in sys.argv[1:]] or [8, 4, 2, 1]:
    starts, goals = synth.start_goal_circle(32, 0.8)
    s = MultiRobotSampler(model, starts, goals, env_id="EnvEmpty2D", n_samples=64, rank=0, world_size=W, device="cuda")
    paths = torch.from_numpy(synth.straight_line_paths(starts, goals, H)).cuda()
    times = []
    for block in range(6):
        watch = bench.PowerSampler()
        watch.start()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(10):
            s.set_other_paths(paths); tr = s.sample(seed=k); s.best_paths(tr, paths)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        p = watch.finish() or {}
        times.append(f"{ms:.1f} ms @ {p.get('sclk_mhz', 0):.0f} MHz {p.get('package_watts', 0):.0f} W")
    print(f"strong W={W} ({s.n_local * 64} trajectories on this GPU), 6 x 10 rounds: " + " | ".join(times), flush=True)
