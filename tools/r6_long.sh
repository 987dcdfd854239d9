# long fuzz / soak runs on the shipped build (new seeds): tools/r6_long.sh  ->  gpurun_out/r6_long/
mkdir -p gpurun_out/r6_long
timeout 700 python tools/fuzz_rollout.py 600 6060 > gpurun_out/r6_long/fuzz_rollout.txt 2>&1
timeout 700 python tools/fuzz_train.py 600 6161 > gpurun_out/r6_long/fuzz_train.txt 2>&1
timeout 400 python tools/soak.py 300 > gpurun_out/r6_long/soak.txt 2>&1
for f in gpurun_out/r6_long/*.txt; do echo "== $f"; grep -v amdgpu.ids $f | tail -n 3; done
