#!/bin/bash
# Where does unet_kernel wait?  Separate PMC passes (kernel-trace only) over tools/unet_forward_loop.py <n>.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
N=${1:-2048}
OUT=gpurun_out
mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ|TCC|TCP|TA|GRBM)_[A-Z0-9_a-z]+" | sort -u > $OUT/pmc_names.txt
: > $OUT/pmc_stall_$N.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SALU" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES"; do
  rm -rf $OUT/pmcx
  rocprofv3 --pmc $set --kernel-trace -f csv -d $OUT/pmcx -o pmc -- python tools/unet_forward_loop.py $N > /dev/null 2> $OUT/pmcx.err
  f=$(find $OUT/pmcx -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then for c in $set; do python tools/pmc_summary.py "$f" $c | grep -E "unet_kernel|^#" ; done >> $OUT/pmc_stall_$N.txt; else echo "FAILED: $set: $(tail -2 $OUT/pmcx.err | tr '\n' ' ')" >> $OUT/pmc_stall_$N.txt; fi
done
rm -rf $OUT/pmcx
cat $OUT/pmc_stall_$N.txt
