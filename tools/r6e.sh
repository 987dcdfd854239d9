set -x
mkdir -p gpurun_out/r6e
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r6e/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6e/pytest.log
grep -E "passed|failed|FAILED|Error" gpurun_out/r6e/pytest.log | head -20
A=cadm_amd/libcadm_hip_var_b.so; B=cadm_amd/libcadm_hip_dev.so
bash tools/ab.sh $A $B 3 > gpurun_out/r6e/ab_cfg2.txt 2>&1
bash tools/ab.sh $A $B 2 --config cfg3 > gpurun_out/r6e/ab_cfg3.txt 2>&1
bash tools/ab.sh $A $B 2 --config cfg4 > gpurun_out/r6e/ab_cfg4.txt 2>&1
bash tools/ab.sh $A $B 2 --config cfg5 > gpurun_out/r6e/ab_cfg5.txt 2>&1
bash tools/ab.sh $A $B 2 --config m10 > gpurun_out/r6e/ab_m10.txt 2>&1
grep -h "M " gpurun_out/r6e/ab_*.txt
