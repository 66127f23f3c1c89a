#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT; rm -f $OUT/ab_small.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "unet or sharded or batch" 2>&1 | tail -12 | tee $OUT/ab_pytest.log
for i in 1 2; do
  for k in big small; do
    MMD_AMD_UNET_KERNEL=$k REPS=40 timeout 120 python tools/unet_forward_loop.py 256 512 1024 2048 2>&1 | grep "n=" | sed "s/\[default lib\]/[$k]/" | tee -a $OUT/ab_small.txt
  done
done
