#!/bin/bash
# bench A/B over the sampler's stream chunks: gpu_streams.sh  (writes gpurun_out/streams.txt)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT; rm -f $OUT/streams.txt
run() { echo "streams=$1: $(MMD_AMD_STREAMS=$1 timeout 300 python bench.py --steps 8 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"]), "traj/s", round(d["ms_per_step"],2), "ms/round; unet launch", round(d["roofline"]["launch_ms"]*1e3,1), "us frac", round(d["roofline"]["frac"],3))')" | tee -a $OUT/streams.txt; }
for n in 1 2 3 2 1; do run $n; done
