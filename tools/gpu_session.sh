export TMPDIR=/tmp
for i in 1 2 3; do
echo "head   $(REPS=40 timeout 300 python tools/unet_forward_loop.py 2048 2>&1 | grep unet | cut -c1-40)"
echo "skew0  $(MMD_AMD_LIB=$PWD/build_tmp/libmmd_amd_skew.so REPS=40 timeout 300 python tools/unet_forward_loop.py 2048 2>&1 | grep unet | cut -c1-40)"
echo "skew6  $(MMD_AMD_UNET_SKEW=6 MMD_AMD_LIB=$PWD/build_tmp/libmmd_amd_skew.so REPS=40 timeout 300 python tools/unet_forward_loop.py 2048 2>&1 | grep unet | cut -c1-40)"
done > gpurun_out/s24_skew.txt
