export TMPDIR=/tmp
for f in 2 0 2 0; do echo "no_fused=$f"; MMD_AMD_NO_FUSED_STEP=$f timeout 300 python tools/dbg/shard_cost.py strong 1 2 4 8 2>&1 | grep "W=" | cut -c1-140; done > gpurun_out/r03_fused_guided_shard.txt; cat gpurun_out/r03_fused_guided_shard.txt
