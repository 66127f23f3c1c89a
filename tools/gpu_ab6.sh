#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT; rm -f $OUT/ab6.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "unet" 2>&1 | tail -6 | tee $OUT/ab_pytest.log
for k in small one; do
  MMD_AMD_UNET_KERNEL=$k REPS=40 timeout 120 python tools/unet_forward_loop.py 64 256 512 1024 2>&1 | grep "n=" | sed "s/default lib/$k/" | tee -a $OUT/ab6.txt
done
