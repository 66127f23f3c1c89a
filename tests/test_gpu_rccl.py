"""The RCCL branch of the per-round all-gather (mmd_amd/multi_robot.py: all_gather_paths) on real hardware.  The GPU boxes
have one device, so the collective runs in a world of ONE rank -- which still loads RCCL, creates the communicator over
the `nccl` backend and executes all_gather_into_tensor on device tensors, i.e. everything but the xGMI transfer itself.
Run in a subprocess: a process group must not leak into the pytest process."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from mmd_amd import synth
from mmd_amd.multi_robot import all_gather_paths, shard_range
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
starts, goals = synth.start_goal_circle(32, 0.8)
paths = torch.from_numpy(synth.straight_line_paths(starts, goals, 64)).cuda()
out = all_gather_paths(paths, 1, force_collective=True)
torch.cuda.synchronize()
assert out.data_ptr() != paths.data_ptr(), "the collective must have run (a fresh output tensor)"
assert torch.equal(out, paths)
assert shard_range(32, 0, 1) == (0, 32)
dist.destroy_process_group()
print("RCCL_OK")
'''


@pytest.mark.gpu
def test_rccl_all_gather_branch_runs_on_device():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", SCRIPT, ROOT], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


# One rank per visible GPU: the planning round's all-gather over RCCL / xGMI between real devices, and every rank's sharded
# sampling call against the rows of the unsharded one.  Runs as soon as the suite lands on a box with >= 2 GPUs (skipped on the
# one-GPU boxes of the pool): rank r owns robots [r n/W, (r + 1) n/W) of an 8 W-robot Empty-map instance.
WORLD_SCRIPT = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
rehearsal = os.environ.get("MMD_RCCL_REHEARSAL") == "1"       # every rank on cuda:0 over gloo: the script's logic on a one-GPU box
dev = torch.device("cuda", 0 if rehearsal else rank)
torch.cuda.set_device(dev)
if rehearsal:
    dist.init_process_group("gloo")
else:
    dist.init_process_group("nccl", device_id=dev)
import gpu_common
from mmd_amd import synth
from mmd_amd.multi_robot import MultiRobotSampler, all_gather_paths, shard_range
N, B, T, H = 8 * world, 8, 25, 64
starts, goals = synth.start_goal_circle(N, 0.8)
paths = torch.from_numpy(synth.straight_line_paths(starts, goals, H)).to(dev)
r0, n_local = shard_range(N, rank, world)
gathered = all_gather_paths(paths[r0:r0 + n_local].contiguous() + 0.001 * rank, world)
expect = paths.clone()
for r in range(world):
    expect[r * n_local:(r + 1) * n_local] += 0.001 * r
assert torch.equal(gathered, expect), "all_gather_into_tensor over RCCL: rank-major rows"
model = gpu_common.hip_model(T)
shard = MultiRobotSampler(model, starts, goals, env_id="EnvEmpty2D", n_samples=B, rank=rank, world_size=world, device=dev)
trajs, best = shard.plan_round(paths[r0:r0 + n_local].contiguous(), seed=5)        # gather -> table -> sample -> pick
full = MultiRobotSampler(model, starts, goals, env_id="EnvEmpty2D", n_samples=B, device=dev)
full.set_other_paths(paths)
ref = full.sample(seed=5)
assert torch.equal(trajs, ref[r0 * B:(r0 + n_local) * B]), "a rank's shard == the rows of the unsharded run, bitwise"
best_all = all_gather_paths(best, world)
assert best_all.shape == (N, H, 2) and torch.isfinite(best_all).all()
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("RCCL_WORLD_OK", world)
'''


def _run_world(world, rehearsal):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    path = os.path.join(ROOT, "gpurun_out", "_rccl_world.py")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(WORLD_SCRIPT)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if rehearsal:
        env["MMD_RCCL_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), path, ROOT], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "RCCL_WORLD_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
def test_rccl_all_gather_world_n():
    import torch
    world = torch.cuda.device_count()
    if world < 2:
        pytest.skip("needs >= 2 GPUs: one rank per GPU over RCCL / xGMI (the one-rank RCCL branch is covered above)")
    _run_world(world, rehearsal=False)


@pytest.mark.gpu
def test_world_script_rehearsal_two_ranks_on_one_gpu():
    """The very script of test_rccl_all_gather_world_n with two ranks sharing cuda:0 over gloo: its sharding / gather / bitwise
    assertions are exercised on the one-GPU boxes too (everything but the RCCL transport between devices)."""
    _run_world(2, rehearsal=True)
