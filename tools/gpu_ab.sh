#!/bin/bash
# A/B of two builds of the library on one box: UNet parity tests on the default build, then forward timings of both.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "unet or teacher" 2>&1 | tail -15 | tee $OUT/ab_pytest.log
for i in 1 2; do
  MMD_AMD_LIB=$PWD/mmd_amd/lib/libmmd_amd_base.so timeout 120 python tools/unet_forward_loop.py 2048 1024 512 2>&1 | grep "n=" | tee -a $OUT/ab_times.txt
  timeout 120 python tools/unet_forward_loop.py 2048 1024 512 2>&1 | grep "n=" | tee -a $OUT/ab_times.txt
done
