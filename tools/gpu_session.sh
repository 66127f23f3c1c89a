export TMPDIR=/tmp
for L in build_tmp/libmmd_amd_a.so build_tmp/libmmd_amd_rd3.so; do MMD_AMD_LIB=$L python tools/unet_forward_loop.py 1024 2048 2>&1 | grep "n="; done | tee gpurun_out/r03a_rd_ab.txt
for n in 1024 2048; do MMD_AMD_LIB=$PWD/build_tmp/libmmd_amd_trace.so timeout 120 python tools/dbg/trace_phases.py $n > gpurun_out/r03a_trace_$n.txt 2>&1; done
REPS=8 tools/gpu_pmc.sh r03a_unet1024 unet_kernel -- python tools/unet_forward_loop.py 1024 > /dev/null
REPS=8 tools/gpu_pmc.sh r03a_unet2048 unet_kernel -- python tools/unet_forward_loop.py 2048 > /dev/null
