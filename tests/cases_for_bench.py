"""Oracle-side inputs of bench.py's cpu_baseline leg (the only non-test user of the oracle besides smoke())."""
import torch

from mmd_amd import synth          # noqa: F401  (re-exported for bench.py)
from oracle import mmd_oracle as O
import cases


def oracle_headline_robot(T, n_robots, robot=0):
    """Robot `robot` of the n_robots Empty-map circle instance: weights, schedule, guide params, its soft-constraint
    group (all other robots' straight-line paths), hard conditions."""
    sd = O.state_dict_to_torch(synth.synth_unet_state_dict(0))
    tb = O.schedule_tables(T)
    gp = cases.guide_params("EnvEmpty2D")
    starts, goals = synth.start_goal_circle(n_robots, 0.8)
    grp = cases.soft_group(synth.straight_line_paths(starts, goals, cases.H), robot)
    return sd, tb, gp, grp, cases.hard_conds_for(starts[robot], goals[robot])
