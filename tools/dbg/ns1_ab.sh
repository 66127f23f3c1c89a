#!/bin/bash
# A/B of unet_kernel<1> (one trajectory per workgroup) against unet_kernel<2> by launch size: side builds -DMMD_NS1_MAX=0 / =512
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
for i in 1 2 3; do
MMD_AMD_LIB=build_tmp/libmmd_amd_nons1.so timeout 100 python tools/unet_forward_loop.py 16 64 128 160 192 224 256 320 2>&1 | grep "n=" | sed 's/^/two per workgroup: /' | cut -c1-60
MMD_AMD_LIB=build_tmp/libmmd_amd_ns1_512.so timeout 100 python tools/unet_forward_loop.py 16 64 128 160 192 224 256 320 2>&1 | grep "n=" | sed 's/^/one per workgroup: /' | cut -c1-60
done
