export TMPDIR=/tmp
for i in 1 2; do
echo "default $(REPS=40 timeout 300 python tools/unet_forward_loop.py 512 2048 2>&1 | grep unet | cut -c1-40 | tr '\n' ' ')"
for v in f_rategymaxilp f_memoryclause f_nlinealltrue; do
echo "$v $(MMD_AMD_LIB=$PWD/build_tmp/libmmd_amd_$v.so REPS=40 timeout 300 python tools/unet_forward_loop.py 512 2048 2>&1 | grep unet | cut -c1-40 | tr '\n' ' ')"
done; done > gpurun_out/s27_flags.txt; cat gpurun_out/s27_flags.txt
