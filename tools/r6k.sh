bash tools/profile_r6.sh > gpurun_out/profile_r6.log 2>&1
tail -5 gpurun_out/profile_r6.log
python tools/horizon_time.py cfg2 2>&1 | grep -v amdgpu > gpurun_out/r6_horizon_cfg2.txt; cat gpurun_out/r6_horizon_cfg2.txt
ls gpurun_out/r6_cfg2 gpurun_out/r6_cfg3 gpurun_out/r6_train_256 gpurun_out/r6_context
