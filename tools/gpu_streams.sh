#!/bin/bash
# A/B of the sampler's stream chunking on the headline round (bench.py --streams -> mmd_sampler_desc.n_streams)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
: > $OUT/${1:-r03}_stream_chunks.txt
run() { echo "streams=$1: $(timeout 300 python bench.py --streams $1 --steps 8 --warmup 1 --no-cpu-baseline --no-pmc --no-power-probe 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"]), "traj/s", round(d["ms_per_step"],2), "ms/round")')" | tee -a $OUT/${2:-r03}_stream_chunks.txt; }
for s in 1 2 3 2 1; do run $s ${1:-r03}; done
