#!/bin/bash
# PMC passes (separate rocprofv3 runs, --kernel-trace only) over back-to-back launches of ONE kernel's loop.
# Usage: tools/gpu_pmc.sh <tag> <kernel-name regex> -- <command...>     -> gpurun_out/<tag>_pmc.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
TAG=$1; KRE=$2; shift 3
OUT=gpurun_out; mkdir -p $OUT
: > $OUT/${TAG}_pmc.txt
# PMC_SETS="A B;C" limits the passes to the given counter sets (';' between passes)
if [ -n "$PMC_SETS" ]; then IFS=';' read -ra SETS <<< "$PMC_SETS"; else SETS=(); fi
run_set() {
  set=$1
  rm -rf $OUT/pmcx
  timeout 300 rocprofv3 --pmc $set --kernel-trace -f csv -d $OUT/pmcx -o pmc -- "${CMD[@]}" > /dev/null 2> $OUT/pmcx.err
  f=$(find $OUT/pmcx -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then for c in $set; do python tools/pmc_summary.py "$f" $c | grep -E "$KRE|^#"; done >> $OUT/${TAG}_pmc.txt
  else echo "FAILED: $set: $(tail -2 $OUT/pmcx.err | tr '\n' ' ')" >> $OUT/${TAG}_pmc.txt; fi
}
CMD=("$@")
if [ ${#SETS[@]} -gt 0 ]; then
  for set in "${SETS[@]}"; do run_set "$set"; done
  rm -rf $OUT/pmcx; cat $OUT/${TAG}_pmc.txt; exit 0
fi
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  run_set "$set"
done
rm -rf $OUT/pmcx
cat $OUT/${TAG}_pmc.txt
