export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rf > gpurun_out/s18_pytest_all.log 2>&1; tail -6 gpurun_out/s18_pytest_all.log
