#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT; rm -f $OUT/ab4.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "unet or sharded or batch" 2>&1 | tail -5 | tee $OUT/ab_pytest.log
for i in 1 2 3; do
  for lib in libmmd_amd_prev.so libmmd_amd.so; do
    MMD_AMD_LIB=$PWD/mmd_amd/lib/$lib REPS=40 timeout 120 python tools/unet_forward_loop.py 2048 2>&1 | grep "n=" | tee -a $OUT/ab4.txt
  done
done
