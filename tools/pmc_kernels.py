"""The two kernels of a planning round, a few launches each, at the launch size the sampler uses: the body of bench.py's live PMC
passes (rocprofv3 --pmc ... --kernel-trace -- python tools/pmc_kernels.py <workload> <trajectories per launch> <T>).  The UNet
forward runs alone; the guided step kernel runs alone on the same batch with the workload's guide (constraint table from the
straight-line paths).  Prints nothing the passes need: rocprofv3's counter_collection.csv is what bench.py reads."""
import os
import sys
from math import ceil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                    # noqa: E402
import bench                                    # noqa: E402
from mmd_amd import synth                       # noqa: E402
from mmd_amd.diffusion_model import GaussianDiffusionModel   # noqa: E402
from mmd_amd.multi_robot import MultiRobotSampler            # noqa: E402
from mmd_amd.temporal_unet import TemporalUnet              # noqa: E402

wl, n_launch, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = int(os.environ.get("REPS", "6"))
w = bench.WORKLOADS[wl]
B = 64
unet = TemporalUnet()
unet.load_state_dict(synth.synth_unet_state_dict(0))
model = GaussianDiffusionModel(model=unet, variance_schedule="exponential", n_diffusion_steps=T, predict_epsilon=True)
n_robots = w["robots"]
if w.get("ensemble"):
    starts, goals = synth.start_goal_circle(n_robots, 0.8)
else:
    starts, goals = bench.workload_starts_goals(w, n_robots)
rpl = max(1, min(n_robots, n_launch // B))      # robots per launch: the first rpl robots of the instance
s = MultiRobotSampler(model, starts, goals, env_id=w["env"], n_samples=B, rank=0, world_size=n_robots // rpl if n_robots % rpl == 0 else 1,
                      inter_robot=w["inter_robot"])
if n_robots % rpl:
    raise SystemExit("robots per launch must divide the instance")
s.set_other_paths(torch.from_numpy(synth.straight_line_paths(starts, goals, 64)).cuda() if w["inter_robot"] else None)
n = s.n_local * B
x = torch.randn(n, 64, 4, device="cuda") * 0.5
for _ in range(reps):
    unet(x, T // 2)
torch.cuda.synchronize()
y = x.clone()
for _ in range(reps):
    model.sample_step(y, s.hard_conds, ceil(0.5 * T) - 1, guide=s.guide, n_guide_steps=20, t_start_guide=ceil(0.5 * T),
                      noise_std_extra_schedule_fn=lambda t: 0.5, n_robots=s.n_local)
torch.cuda.synchronize()
print("pmc_kernels done", wl, n, flush=True)
