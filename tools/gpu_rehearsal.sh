#!/bin/bash
# Rehearse the N>1 bench path on ONE GPU (all ranks on cuda:0, gloo) with EXACTLY the command line the driver uses for a
# multi-GPU run -- `python3 bench.py --gpus N --steps K --warmup W`, no launcher: bench.py re-launches itself under
# torch.distributed.run.  Not a measurement: a does-it-run check of the sharded code path (strong headline + weak alongside).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-r05}
for n in 2 4; do
  MMD_BENCH_REHEARSAL=1 timeout 900 python3 bench.py --gpus $n --steps 2 --warmup 1 2>$OUT/${TAG}_rehearsal_$n.err | tee $OUT/${TAG}_rehearsal_gpus$n.json | cut -c1-500
  tail -2 $OUT/${TAG}_rehearsal_$n.err
done
