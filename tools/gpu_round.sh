#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel stats of the SAME bench command, and PMC passes
# (FETCH_SIZE / WRITE_SIZE in separate runs, kernel-trace only) for the HBM traffic of the dominant kernel.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
python bench.py --steps 3 --warmup 1 2>$OUT/bench.err | tee $OUT/bench.json | cut -c1-400
rm -rf $OUT/prof; mkdir -p $OUT/prof
rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/prof.err
python tools/rocpd_summary.py $OUT/prof/bench_results.db > $OUT/kernel_stats.md && head -30 $OUT/kernel_stats.md
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_$c
  rocprofv3 --pmc $c --kernel-trace -f csv -d $OUT/pmc_$c -o pmc -- python tools/unet_forward_loop.py 2048 > /dev/null 2> $OUT/pmc_$c.err
  f=$(find $OUT/pmc_$c -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py "$f" $c | tee $OUT/pmc_$c.txt
done
rm -rf $OUT/prof/bench_results.db $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
