#!/bin/bash
# rocprofv3 kernel-trace stats of the bench command (summaries copied to profiles/ by hand afterwards).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.log
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prof.json 2> gpurun_out/prof.err
tail -3 gpurun_out/prof.err
find gpurun_out/prof -type f | head -20
f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -40 "$f"
# drop the big raw trace, keep the stats
find gpurun_out/prof -name '*kernel_trace.csv' -size +20M -delete
