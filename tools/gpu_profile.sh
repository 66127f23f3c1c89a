#!/bin/bash
# Full GPU-box session for a round's records: parity tests, smoke, bench, rocprofv3 kernel stats of the SAME bench command,
# PMC passes (separate runs, --kernel-trace only) for unet_kernel (a stream chunk's launch size) and for the kernels of a
# bench round (the guided step kernel), forward times by batch size, per-GPU shard costs, the layer-by-layer path, the N > 1 rehearsal.
# Usage: tools/gpu_profile.sh [tag]   (outputs under gpurun_out/<tag>_*)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
TAG=${1:-r06}
OUT=gpurun_out
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -rf 2>&1 | tail -25 | tee $OUT/${TAG}_pytest_gpu.log
cp $OUT/r06_parity.json $OUT/${TAG}_parity.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/${TAG}_smoke.log
# PMC first: bench.py quotes profiles/pmc_latest.json for `traffic` / `mfma_busy_pmc` (copied there after the session)
REPS=8 tools/gpu_pmc.sh ${TAG}_unet1024 unet_kernel -- python tools/unet_forward_loop.py 1024 > /dev/null
REPS=8 PMC_SETS="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" tools/gpu_pmc.sh ${TAG}_unet2048 unet_kernel -- python tools/unet_forward_loop.py 2048 > /dev/null
tools/gpu_pmc.sh ${TAG}_bench "unet_kernel|ddpm_guide" -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-power-probe --no-pmc > /dev/null
python tools/pmc_to_json.py $OUT/${TAG}_unet1024_pmc.txt 1024 $OUT/${TAG}_bench_pmc.txt $OUT/${TAG}_unet2048_pmc.txt > /dev/null && cp profiles/pmc_latest.json $OUT/${TAG}_pmc_latest.json
timeout 900 python bench.py --steps 10 --warmup 2 2>$OUT/bench.err | tee $OUT/${TAG}_bench.json | cut -c1-300
# BASELINE.json's other configs: the same line per workload
for w in config2 config3 config4 config5; do
  timeout 600 python bench.py --workload $w --steps 10 --warmup 2 2>>$OUT/bench.err | tee $OUT/${TAG}_bench_$w.json | cut -c1-260
done
cp $OUT/${TAG}_bench.json $OUT/${TAG}_bench_headline.json
rm -rf $OUT/prof; mkdir -p $OUT/prof
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-power-probe --no-pmc > $OUT/${TAG}_bench_prof.json 2> $OUT/prof.err
python tools/rocpd_summary.py $OUT/prof/bench_results.db > $OUT/${TAG}_rocprofv3_kernel_stats.md && head -16 $OUT/${TAG}_rocprofv3_kernel_stats.md
rm -rf $OUT/prof/bench_results.db
timeout 300 python tools/unet_forward_loop.py 256 512 1024 2048 4096 2>&1 | grep "n=" | tee $OUT/${TAG}_unet_sizes.txt
timeout 300 python tools/dbg/shard_cost.py strong 1 2 4 8 2>&1 | grep "W=" | tee $OUT/${TAG}_shard_cost.txt
timeout 300 python tools/dbg/shard_cost.py weak 1 2 4 8 2>&1 | grep "W=" | tee -a $OUT/${TAG}_shard_cost.txt
# the layer-by-layer TemporalUnet path (option 1 / option 0 forced): forward by batch size on the matrix pipe and on the vector ALUs, and
# a planning round of the headline workload with the option-1 network
for v in 0 1; do MMD_AMD_LAYERED_VALU=$v timeout 300 python tools/dbg/layered_time.py 64 256 1024 4096 2>&1 | grep "n=" | tee -a $OUT/${TAG}_layered_time.txt; done
for v in 0 1; do MMD_AMD_LAYERED_VALU=$v timeout 600 python tools/dbg/option1_round.py 3 2>&1 | grep "dim_mults" | tee -a $OUT/${TAG}_option1_round.txt; done
# config4 with the planner calls issued concurrently (one stream each) and one after the other (the default packs them into ONE launch sequence)
timeout 600 python bench.py --workload config4 --steps 10 --warmup 2 --concurrent-planners --no-cpu-baseline --no-pmc --no-power-probe 2>>$OUT/bench.err | tee $OUT/${TAG}_bench_config4_concurrent.json | cut -c1-260
timeout 600 python bench.py --workload config4 --steps 10 --warmup 2 --sequential-planners --no-cpu-baseline --no-pmc --no-power-probe 2>>$OUT/bench.err | tee $OUT/${TAG}_bench_config4_sequential.json | cut -c1-260
cp $OUT/bench_detail_*.json $OUT/ 2>/dev/null; for f in $OUT/bench_detail_*_n1.json; do cp $f $OUT/${TAG}_$(basename $f); done
timeout 300 python tools/dbg/planner_time.py 2>&1 | tail -2 | tee $OUT/${TAG}_planner_time.txt
timeout 300 python tools/dbg/planner_breakdown.py 2>&1 | tail -1 | tee -a $OUT/${TAG}_planner_time.txt
timeout 300 python tools/dbg/replan_time.py 2>&1 | grep "re-plan" | tee -a $OUT/${TAG}_planner_time.txt
bash tools/gpu_rehearsal.sh ${TAG}
