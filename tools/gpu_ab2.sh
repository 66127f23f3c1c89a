#!/bin/bash
# A/B of several builds on one box: forward timings, interleaved twice.  Usage: gpu_ab2.sh libA.so libB.so ...
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT; rm -f $OUT/ab_times.txt
for i in 1 2; do
  for lib in "$@"; do
    MMD_AMD_LIB=$PWD/mmd_amd/lib/$lib REPS=40 timeout 120 python tools/unet_forward_loop.py 2048 1024 2>&1 | grep "n=" | tee -a $OUT/ab_times.txt
  done
done
