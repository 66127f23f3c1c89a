#!/bin/bash
# build ablation variants of the library (NOT shipped; for profiling only)
set -e
cd "$(dirname "$0")/.."
for a in 1; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -DMMD_ABL=$a \
    mmd_amd/csrc/unet.hip mmd_amd/csrc/guide.hip mmd_amd/csrc/api.hip mmd_amd/csrc/multi_agent.hip -o gpurun_in_abl$a.so &
done
wait
ls -la gpurun_in_abl*.so
