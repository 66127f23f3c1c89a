export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "unet or forward" > gpurun_out/s22_pytest_unet.log 2>&1; tail -3 gpurun_out/s22_pytest_unet.log
for i in 1 2 3; do
for v in prev r3222 r3322 r3334 r3332; do
MMD_AMD_LIB=$PWD/build_tmp/libmmd_amd_$v.so REPS=40 timeout 300 python tools/unet_forward_loop.py 512 2048 2>&1 | grep unet
done; done > gpurun_out/s22_ab.txt
