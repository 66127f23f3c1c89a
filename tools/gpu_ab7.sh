#!/bin/bash
# unet parity tests on the in-tree library + forward timing against the side library given as $1
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT; rm -f $OUT/ab7.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "unet" 2>&1 | tail -8 | tee $OUT/ab_pytest.log
for lib in $1 mmd_amd/lib/libmmd_amd.so; do
  MMD_AMD_LIB=$PWD/$lib REPS=${REPS:-40} timeout 300 python tools/unet_forward_loop.py ${SIZES:-256 1024 2048 4096} 2>&1 | grep "n=" | tee -a $OUT/ab7.txt
done
