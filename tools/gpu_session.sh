export TMPDIR=/tmp
for i in 1 2; do
MMD_AMD_LIB=$PWD/build_tmp/libmmd_amd_head.so REPS=40 timeout 300 python tools/unet_forward_loop.py 256 512 1024 2048 4096 2>&1 | grep unet
REPS=40 timeout 300 python tools/unet_forward_loop.py 256 512 1024 2048 4096 2>&1 | grep unet
done > gpurun_out/s3_ab.txt
