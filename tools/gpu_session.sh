export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "unet or forward" > gpurun_out/s20_pytest_unet.log 2>&1; tail -3 gpurun_out/s20_pytest_unet.log
for i in 1 2 3; do
MMD_AMD_LIB=$PWD/build_tmp/libmmd_amd_prev.so REPS=40 timeout 300 python tools/unet_forward_loop.py 256 1024 2048 2>&1 | grep unet
REPS=40 timeout 300 python tools/unet_forward_loop.py 256 1024 2048 2>&1 | grep unet
done > gpurun_out/s20_ab.txt
