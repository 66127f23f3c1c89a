#!/bin/bash
# GPU-box session: parity tests (all, no -x), smoke, bench, small-batch UNet timings.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | tail -60 | tee $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
timeout 600 python bench.py --steps 3 --warmup 1 2>$OUT/bench.err | tee $OUT/bench.json | cut -c1-600
timeout 300 python tools/unet_forward_loop.py 256 512 1024 2048 2>&1 | tee $OUT/unet_sizes.txt
