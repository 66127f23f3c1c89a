#!/bin/bash
# Full GPU-box session for a round's records: parity tests, smoke, bench, rocprofv3 kernel stats of the SAME bench command,
# PMC passes (separate runs, --kernel-trace only) for HBM traffic / MFMA utilisation / stalls of unet_kernel.
# Usage: tools/gpu_profile.sh [tag]   (outputs under gpurun_out/<tag>_*)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
TAG=${1:-r02}
OUT=gpurun_out
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | tail -25 | tee $OUT/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/${TAG}_smoke.log
timeout 900 python bench.py --steps 5 --warmup 1 2>$OUT/bench.err | tee $OUT/${TAG}_bench.json | cut -c1-300
rm -rf $OUT/prof; mkdir -p $OUT/prof
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_bench_prof.json 2> $OUT/prof.err
python tools/rocpd_summary.py $OUT/prof/bench_results.db > $OUT/${TAG}_rocprofv3_kernel_stats.md && head -16 $OUT/${TAG}_rocprofv3_kernel_stats.md
: > $OUT/${TAG}_pmc_unet_kernel.txt
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  rm -rf $OUT/pmcx
  REPS=8 timeout 300 rocprofv3 --pmc $set --kernel-trace -f csv -d $OUT/pmcx -o pmc -- python tools/unet_forward_loop.py ${PMC_N:-1024} > /dev/null 2> $OUT/pmcx.err
  f=$(find $OUT/pmcx -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then for c in $set; do python tools/pmc_summary.py "$f" $c | grep -E "unet_kernel|^#"; done >> $OUT/${TAG}_pmc_unet_kernel.txt
  else echo "FAILED: $set: $(tail -2 $OUT/pmcx.err | tr '\n' ' ')" >> $OUT/${TAG}_pmc_unet_kernel.txt; fi
done
rm -rf $OUT/pmcx $OUT/prof/bench_results.db
cat $OUT/${TAG}_pmc_unet_kernel.txt
timeout 300 python tools/unet_forward_loop.py 256 512 1024 2048 4096 2>&1 | grep "n=" | tee $OUT/${TAG}_unet_sizes.txt
