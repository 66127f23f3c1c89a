#!/bin/bash
# MFMA utilisation counters of the UNet kernels (separate PMC pass, kernel-trace only).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -i -E "MFMA|GRBM_GUI_ACTIVE|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|SQ_INSTS_VALU_MFMA" | head -30 > $OUT/pmc_list.txt
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_')
  rm -rf $OUT/pmcx
  rocprofv3 --pmc $set --kernel-trace -f csv -d $OUT/pmcx -o pmc -- python tools/unet_forward_loop.py 2048 > /dev/null 2> $OUT/pmcx.err
  f=$(find $OUT/pmcx -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then for c in $set; do python tools/pmc_summary.py "$f" $c | head -8; done | tee $OUT/pmc_$tag.txt; else tail -3 $OUT/pmcx.err; fi
done
rm -rf $OUT/pmcx
cat $OUT/pmc_list.txt | head -20
