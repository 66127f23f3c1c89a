#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
for n in 2048; do
  MMD_AMD_UNET_KERNEL=big MMD_AMD_LIB=$PWD/mmd_amd/lib/libmmd_amd_trace.so timeout 120 python tools/dbg/trace_phases.py $n > $OUT/trace_$n.txt 2>&1
done
tail -8 $OUT/trace_2048.txt
