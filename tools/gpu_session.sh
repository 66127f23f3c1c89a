export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 300 python bench.py --steps 5 --warmup 1 2>$OUT/bench.err | tee $OUT/r03b_bench.json | cut -c1-600; tail -3 $OUT/bench.err
bash tools/gpu_rehearsal.sh r03b
tools/gpu_pmc.sh r03b_bench "unet_kernel|ddpm_guide" -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null
grep -A1 -E "VALU|WAVE_CYCLES|WAIT|ACTIVE|FETCH|WRITE|LDS" $OUT/r03b_bench_pmc.txt | grep -E "#|ddpm" | head -60
