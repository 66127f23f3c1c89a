export TMPDIR=/tmp
for n in 512 2048; do MMD_AMD_LIB=$PWD/build_tmp/libmmd_amd_trace.so timeout 120 python tools/dbg/trace_phases.py $n > gpurun_out/r03_trace_${n}_all_direct.txt 2>&1; done
