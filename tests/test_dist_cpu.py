"""CPU, world_size 2 over gloo: the N>1 path's host logic -- robot sharding, the per-round all-gather of best paths,
and that the constraint table each rank derives from the gathered paths equals the single-process one."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mmd_amd import synth
from mmd_amd.multi_robot import all_gather_paths, shard_range

H = 64


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_robots, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        starts, goals = synth.start_goal_circle(n_robots, 0.8)
        paths = synth.straight_line_paths(starts, goals, H)
        r0, n_local = shard_range(n_robots, rank, world)
        local = torch.from_numpy(paths[r0:r0 + n_local]) + 0.001 * rank          # rank-specific content
        gathered = all_gather_paths(local, world)
        np.save(os.path.join(out_dir, f"gathered_{rank}.npy"), gathered.numpy())
        # second round with different data: the collective is re-entrant
        gathered2 = all_gather_paths(local * 2, world)
        assert torch.allclose(gathered2, gathered * 2)
    finally:
        dist.destroy_process_group()


def test_shard_range():
    assert shard_range(32, 0, 1) == (0, 32)
    assert [shard_range(64, r, 8) for r in (0, 3, 7)] == [(0, 8), (24, 8), (56, 8)]
    with pytest.raises(ValueError):
        shard_range(10, 0, 4)


def test_all_gather_paths_world2(tmp_path):
    n_robots, world = 6, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_robots, str(tmp_path)), nprocs=world, join=True)
    starts, goals = synth.start_goal_circle(n_robots, 0.8)
    paths = synth.straight_line_paths(starts, goals, H)
    expect = paths.copy()
    expect[3:] += 0.001
    g0, g1 = (np.load(tmp_path / f"gathered_{r}.npy") for r in (0, 1))
    assert np.array_equal(g0, g1), "every rank must see the same gathered paths"
    assert np.allclose(g0, expect) and g0.shape == (n_robots, H, 2)


def test_sharded_constraint_tables_equal_global():
    """What rank r derives for its robots from the gathered paths == the rows of the 1-rank table (host restatement of
    mmd_soft_constraints_from_paths; the device kernel itself is checked in test_gpu_parity)."""
    from oracle import mmd_oracle as O
    n_robots = 6
    starts, goals = synth.start_goal_circle(n_robots, 0.8)
    paths = torch.from_numpy(synth.straight_line_paths(starts, goals, H))
    full = [O.soft_constraints_from_paths(paths, r, 0.12, 2e-2) for r in range(n_robots)]
    for world in (2, 3):
        for rank in range(world):
            r0, n_local = shard_range(n_robots, rank, world)
            for i in range(n_local):
                grp = O.soft_constraints_from_paths(paths, r0 + i, 0.12, 2e-2)
                assert torch.equal(grp.q, full[r0 + i].q) and torch.equal(grp.t_range, full[r0 + i].t_range)
                assert grp.q.shape[0] == (n_robots - 1) * 63
