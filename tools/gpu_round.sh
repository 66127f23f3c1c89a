#!/bin/bash
# One GPU-box session: parity tests, smoke, bench (+ rocprofv3 kernel stats of the same bench command).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.log
python bench.py --steps 3 --warmup 1 2>gpurun_out/bench.err | tee gpurun_out/bench.json
tail -5 gpurun_out/bench.err
