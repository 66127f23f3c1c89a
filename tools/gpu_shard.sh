#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 300 python tools/dbg/shard_cost.py strong 1 2 4 8 2>&1 | grep "W=" | tee $OUT/shard_cost.txt
timeout 300 python tools/dbg/shard_cost.py weak 1 2 4 8 2>&1 | grep "W=" | tee -a $OUT/shard_cost.txt
