# round-6 evidence on the final build: robustness script, then the profile passes
bash tools/r6_robust.sh > gpurun_out/r6_robust.log 2>&1
bash tools/profile_r6.sh > gpurun_out/profile_r6.log 2>&1
python tools/horizon_time.py cfg2 > gpurun_out/r6_horizon_cfg2.txt 2>&1
tail -5 gpurun_out/r6_robust.log; tail -3 gpurun_out/profile_r6.log
