export TMPDIR=/tmp
for sk in 0 4 8 12 16 24 0; do echo "skew $sk"; MMD_AMD_UNET_SKEW=$sk REPS=40 timeout 300 python tools/unet_forward_loop.py 2048 4096 2>&1 | grep unet | cut -c1-40; done > gpurun_out/s14_skew.txt
