set -x
mkdir -p gpurun_out/r6d
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_tf_pin.py tests/test_isa_hygiene.py tests/test_gpu_planner.py tests/test_gpu_model.py -m gpu -q -x > gpurun_out/r6d/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6d/pytest.log
tail -30 gpurun_out/r6d/pytest.log
