cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; OUT=gpurun_out; mkdir -p $OUT
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  rm -rf $OUT/pmcx
  rocprofv3 --pmc $set --kernel-trace -f csv -d $OUT/pmcx -o pmc -- python tools/unet_forward_loop.py 2048 > /dev/null 2> $OUT/pmcx.err
  f=$(find $OUT/pmcx -name '*counter_collection.csv' | head -1)
  for c in $set; do python tools/pmc_summary.py "$f" $c | grep -E "unet_kernel"| sed "s/^/$c /"; done
done
rm -rf $OUT/pmcx
