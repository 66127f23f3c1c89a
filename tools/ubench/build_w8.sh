#!/bin/bash
# builds the W8 arms of tools/ubench/fatwave_conv.hip (and the kernel-form arms beside them) into build_tmp/ (cross-compiles)
cd "$(dirname "$0")/../.."
mkdir -p build_tmp
REST="mmd_amd/csrc/unet_layers.hip mmd_amd/csrc/guide.hip mmd_amd/csrc/api.hip mmd_amd/csrc/multi_agent.hip mmd_amd/csrc/postprocess.hip"
cc() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -w "$@" tools/ubench/fatwave_conv.hip $REST; }
cc -DPARTS=63 -o build_tmp/ub_base_p63 &
cc -DPARTS=7 -o build_tmp/ub_base_p7 &
cc -DPARTS=5 -o build_tmp/ub_base_p5 &
cc -DW8=63 -o build_tmp/ub_w8_p63 &
wait
cc -DW8=7 -o build_tmp/ub_w8_p7 &
cc -DW8=5 -o build_tmp/ub_w8_p5 &
cc -DW8=3 -o build_tmp/ub_w8_p3 &
cc -DW8=127 -DW8_SB=2 -o build_tmp/ub_w8_p127_sb2 &
wait
cc -DW8=127 -DW8_SB=1 -o build_tmp/ub_w8_p127_sb1 &
cc -DW8=71 -DW8_SB=2 -o build_tmp/ub_w8_p71_sb2 &
wait
ls -la build_tmp/ub_*
