export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "unet or forward" > gpurun_out/s6_pytest_unet.log 2>&1; tail -3 gpurun_out/s6_pytest_unet.log
for i in 1 2; do
MMD_AMD_LIB=$PWD/build_tmp/libmmd_amd_u1w.so REPS=40 timeout 300 python tools/unet_forward_loop.py 256 1024 2048 2>&1 | grep unet
REPS=40 timeout 300 python tools/unet_forward_loop.py 256 1024 2048 2>&1 | grep unet
done > gpurun_out/s6_ab.txt
timeout 300 python bench.py --steps 10 --warmup 2 > gpurun_out/s6_bench.json 2> gpurun_out/s6_bench.err; cut -c1-200 gpurun_out/s6_bench.json
