set -x
mkdir -p gpurun_out/r6a
tools/micro/mfma_k16_rate > gpurun_out/r6a/mfma_k16_rate.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r6a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6a/pytest.log
tail -5 gpurun_out/r6a/pytest.log
timeout 900 python -m pytest tests/test_gpu_precision.py -q -s -k head_sd > gpurun_out/r6a/head_sd.log 2>&1
A=cadm_amd/libcadm_hip_var_r5.so; B=cadm_amd/libcadm_hip_dev.so
bash tools/ab.sh $A $B 3 > gpurun_out/r6a/ab_cfg2.txt 2>&1
bash tools/ab.sh $A $B 2 --config cfg3 > gpurun_out/r6a/ab_cfg3.txt 2>&1
bash tools/ab.sh $A $B 2 --config cfg4 > gpurun_out/r6a/ab_cfg4.txt 2>&1
bash tools/ab.sh $A $B 2 --config cfg5 > gpurun_out/r6a/ab_cfg5.txt 2>&1
cat gpurun_out/r6a/mfma_k16_rate.txt gpurun_out/r6a/ab_*.txt
grep "head sd" gpurun_out/r6a/head_sd.log
