"""VERDICT r5 #3, measured: does replaying the sampling loop of one planner call (B = 64, T = 25: 26 UNet launches + 14 guided-step launches
+ the init kernel, all enqueued by ONE C-ABI call without a host synchronisation) as a captured hipGraph shorten its DEVICE span?
Eager = torch.ops.mmd_amd.p_sample_loop on the current stream; graph = the same op captured once (torch.cuda.graph) and replayed.  HIP events
around both, median of 20.  Usage: planner_graph_ab.py [n_samples] [T]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import cases
import gpu_common
from mmd_amd import _lib, ops, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 25
H, D = 64, 4
model = gpu_common.hip_model(T)
starts, goals = synth.start_goal_circle(10, 0.45)
guide = gpu_common.hip_guide("EnvHighways2D", [[]])
hc = cases.hard_conds_for(starts[3], goals[3])
hard = torch.stack([hc[0], hc[H - 1]], dim=0)[None].cuda().contiguous()
tm, tg = ops.register(model), ops.register(guide)
sg = _lib.signed64(_lib.HARD_ROWS_START_GOAL)
x = torch.empty((B, H, D), device="cuda")


def loop():
    return torch.ops.mmd_amd.p_sample_loop(x, hard, sg, tm, tg, 1, T, 1, True, None, 77, 20, (T + 1) // 2, 0.5, 0, True)


def span(fn, n=20):
    out = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1))
    return sorted(out)[n // 2]


for _ in range(3):
    ref = loop()
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    cg = loop()
graph.replay()
torch.cuda.synchronize()
assert torch.equal(cg, ref)
a, b = span(loop), span(graph.replay)
a2, b2 = span(loop), span(graph.replay)
print(f"B={B} T={T}: device span of the sampling loop  eager {a:.3f} / {a2:.3f} ms   captured hipGraph replay {b:.3f} / {b2:.3f} ms   (bitwise equal results)")
