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


def oracle_workload_robot(w, T, robot=0):
    """Robot `robot` of a bench.py workload (bench.WORKLOADS entry): weights, schedule, guide params of its map, its constraint
    groups (the other robots' straight-line paths where the workload has the inter-robot term), hard conditions.  The ensemble
    workload (config4) is represented by ONE of its two tile models (no constraints, MPDEnsemble's 0.01 cutoff margin,
    mpd_ensemble.py:139); bench.py counts a step twice."""
    sd = O.state_dict_to_torch(synth.synth_unet_state_dict(0))
    tb = O.schedule_tables(T)
    gp = cases.guide_params(w["env"], cutoff=0.01 if w.get("ensemble") else 0.05)
    n = w["robots"]
    if w.get("ensemble"):
        return sd, tb, gp, [], cases.hard_conds_for((-0.7, -0.6), (0.7, 0.6))
    f = w["formation"]
    starts, goals = synth.start_goal_circle(n, f[1]) if f[0] == "circle" else synth.start_goal_boundary(n)
    groups = [cases.soft_group(synth.straight_line_paths(starts, goals, cases.H), robot)] if w["inter_robot"] else []
    return sd, tb, gp, groups, cases.hard_conds_for(starts[robot], goals[robot])
