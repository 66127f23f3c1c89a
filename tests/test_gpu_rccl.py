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
