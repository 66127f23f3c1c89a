export TMPDIR=/tmp
for n in 1024 2048; do MMD_AMD_LIB=$PWD/build_tmp/libmmd_amd_trace.so timeout 120 python tools/dbg/trace_phases.py $n > gpurun_out/r03i_trace_$n.txt 2>&1; done
REPS=8 tools/gpu_pmc.sh r03i_unet2048 unet_kernel -- python tools/unet_forward_loop.py 2048 > /dev/null
