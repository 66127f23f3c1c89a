#!/bin/bash
# Build libmmd_amd.so for gfx950 (cross-compiles without a GPU).  Usage: ./build.sh [extra hipcc flags]
# -fno-slp-vectorize: the SLP vectorizer pairs fp32 ops into v_pk_* wherever it can and pays for it with v_mov shuffles; the
# kernels use packed fp32 arithmetic only where it is written out (GroupNorm + Mish epilogue, the guide's (x, y) pairs).
set -e
cd "$(dirname "$0")"
mkdir -p mmd_amd/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -fno-slp-vectorize \
  mmd_amd/csrc/unet.hip mmd_amd/csrc/unet_layers.hip mmd_amd/csrc/guide.hip mmd_amd/csrc/api.hip mmd_amd/csrc/multi_agent.hip mmd_amd/csrc/postprocess.hip -o mmd_amd/lib/libmmd_amd.so "$@"
echo "built mmd_amd/lib/libmmd_amd.so"
