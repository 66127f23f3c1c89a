#!/bin/bash
# A/B of several builds on one box: UNet parity tests on the default build, then forward timings interleaved 3x.
# Usage: gpu_ab2.sh libA.so libB.so ...   (files under mmd_amd/lib/)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT; rm -f $OUT/ab_times.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "unet or sharded or batch" 2>&1 | tail -4 | tee $OUT/ab_pytest.log
for i in 1 2 3; do
  for lib in "$@"; do
    MMD_AMD_LIB=$PWD/mmd_amd/lib/$lib REPS=40 timeout 120 python tools/unet_forward_loop.py ${SIZES:-2048} 2>&1 | grep "n=" | tee -a $OUT/ab_times.txt
  done
done
