#!/bin/bash
# Rehearse the N>1 bench path on ONE GPU (all ranks on cuda:0, gloo): not a measurement, a does-it-run check of the
# sharded code path (strong and weak), plus the RCCL single-process sanity of all_gather_into_tensor at world size 1.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
for mode in strong weak; do
  MMD_BENCH_REHEARSAL=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 2 --warmup 1 --scaling $mode 2>$OUT/rehearsal_$mode.err | tee $OUT/rehearsal_$mode.json | cut -c1-700
  tail -3 $OUT/rehearsal_$mode.err
done
MMD_BENCH_REHEARSAL=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29518 \
    bench.py --gpus 4 --steps 1 --warmup 1 2>$OUT/rehearsal_4.err | tee $OUT/rehearsal_4.json | cut -c1-400
