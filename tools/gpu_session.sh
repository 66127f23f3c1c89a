export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rf > gpurun_out/s19_pytest_all.log 2>&1; tail -4 gpurun_out/s19_pytest_all.log
REPS=40 timeout 300 python tools/unet_forward_loop.py 256 512 1024 2048 2>&1 | grep unet | cut -c1-45
