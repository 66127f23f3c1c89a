export TMPDIR=/tmp
for n in 1024 2048; do MMD_AMD_LIB=$PWD/build_tmp/libmmd_amd_trace.so timeout 120 python tools/dbg/trace_phases.py $n > gpurun_out/r03f_trace_$n.txt 2>&1; done
