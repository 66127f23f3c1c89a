"""One planning round of the headline workload (32 robots x 64 samples on the Empty map, T = 100 + 1, all-pairs soft constraints) with the
reference's OTHER network shape, UNET_DIM_MULTS[1] = (1, 2, 4, 8): the layer-by-layer TemporalUnet path end to end in the sampler.
MMD_AMD_LAYERED_VALU=1 (read HERE by this tool and passed to TemporalUnet(layered_valu=...) -> mmd_unet_options; the library reads no environment) = the vector-ALU kernels.  Usage: option1_round.py [rounds]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmd_amd import synth
from mmd_amd.diffusion_model import GaussianDiffusionModel
from mmd_amd.multi_robot import MultiRobotSampler
from mmd_amd.temporal_unet import TemporalUnet

H, T, B, N = 64, 100, 64, 32
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for dm in ((1, 2, 4, 8), (1, 2, 4)):
    unet = TemporalUnet(state_dim=4, n_support_points=H, unet_input_dim=32, dim_mults=dm, layered_valu=os.environ.get('MMD_AMD_LAYERED_VALU', '0') == '1')
    unet.load_state_dict(synth.synth_unet_state_dict(0, dim_mults=dm))
    model = GaussianDiffusionModel(model=unet, variance_schedule="exponential", n_diffusion_steps=T, predict_epsilon=True)
    starts, goals = synth.start_goal_circle(N, 0.8)
    s = MultiRobotSampler(model, starts, goals, env_id="EnvEmpty2D", n_samples=B, device="cuda")
    paths = torch.from_numpy(synth.straight_line_paths(starts, goals, H)).cuda()
    for k in range(2):
        trajs, paths = s.plan_round(paths, seed=100 + k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(rounds):
        trajs, paths = s.plan_round(paths, seed=200 + k)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / rounds * 1e3
    assert torch.isfinite(trajs).all()
    print(f"dim_mults {dm}: {ms:8.2f} ms per round of {N * B} trajectories = {N * B / ms:7.2f} k trajectories/s   "
          f"[MMD_AMD_LAYERED_VALU={os.environ.get('MMD_AMD_LAYERED_VALU', '0')}]", flush=True)
