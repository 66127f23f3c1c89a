#!/bin/bash
# forward timing of side libraries: gpu_ab.sh lib1.so lib2.so ...   (SIZES, REPS)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT; rm -f $OUT/ab.txt
for lib in "$@"; do
  MMD_AMD_LIB=$PWD/$lib REPS=${REPS:-40} timeout 300 python tools/unet_forward_loop.py ${SIZES:-1024 2048 4096} 2>&1 | grep "n=" | tee -a $OUT/ab.txt
done
