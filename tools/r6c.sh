set -x
mkdir -p gpurun_out/r6c
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r6c/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6c/pytest.log
tail -8 gpurun_out/r6c/pytest.log
