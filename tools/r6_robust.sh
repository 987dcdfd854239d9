mkdir -p gpurun_out/r6_final
python -m pytest tests -q -m gpu > gpurun_out/r6_final/gpu_suite.txt 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_final/bench.json 2> gpurun_out/r6_final/bench.err
timeout 400 python tools/fuzz_train.py 150 51 > gpurun_out/r6_final/fuzz_train.txt 2>&1
timeout 400 python tools/fuzz_rollout.py 200 77 > gpurun_out/r6_final/fuzz_rollout.txt 2>&1
timeout 300 python tools/soak.py 60 > gpurun_out/r6_final/soak.txt 2>&1
timeout 300 python tools/soak.py 40 2000 > gpurun_out/r6_final/soak2000.txt 2>&1
python tools/train_scaling.py > gpurun_out/r6_final/train_scaling.txt 2>&1
for f in gpurun_out/r6_final/*.txt; do echo "== $f"; tail -n 3 $f; done; head -c 600 gpurun_out/r6_final/bench.json
