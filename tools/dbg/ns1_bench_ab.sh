#!/bin/bash
# config4 / config2 / one MPD call with unet_kernel<1> up to 256 (the product), up to 128, and without it (side builds)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
F="--steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-power-probe"
for i in 1 2; do
for lib in build_tmp/libmmd_amd_nons1.so build_tmp/libmmd_amd_ns1_128.so mmd_amd/lib/libmmd_amd.so; do
  for w in config4 config2; do
    MMD_AMD_LIB=$lib timeout 300 python tools/bench_with_lib.py --workload $w $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $w', round(d['value']), round(d['ms_per_step'],2))"
  done
done
done
