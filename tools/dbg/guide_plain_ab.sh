#!/bin/bash
# A/B: the guide kernels with the planners' configuration folded in at compile time (side build -DMMD_GUIDE_PLAIN) against the product
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
F="--steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-power-probe"
for i in 1 2 3; do
for lib in mmd_amd/lib/libmmd_amd.so build_tmp/libmmd_amd_plain.so; do
  for w in headline config3; do
    MMD_AMD_LIB=$lib timeout 300 python tools/bench_with_lib.py --workload $w $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $w', round(d['value']), round(d['ms_per_step'],2))"
  done
done
done
