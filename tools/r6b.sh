set -x
mkdir -p gpurun_out/r6b
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r6b/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6b/pytest.log
tail -5 gpurun_out/r6b/pytest.log
bash tools/abv.sh 3 libcadm_hip_var_a.so libcadm_hip_var_notouch.so libcadm_hip_var_b.so > gpurun_out/r6b/abv.txt 2>&1
cat gpurun_out/r6b/abv.txt
bash tools/kstats.sh --steps 100 > gpurun_out/r6b/kstats_cfg2.txt 2>&1
cat gpurun_out/r6b/kstats_cfg2.txt
