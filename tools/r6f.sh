ARGS="halfcheetah 1 200 5 20 1 7 12 0"
for INJ in 1 0; do
echo "=== current inject=$INJ"; timeout 120 python tools/repro_flavours.py - $ARGS $INJ 1 2 3 4 2>&1 | grep -v amdgpu.ids | grep "=="
done
timeout 300 python -m pytest tests/test_gpu_rowtiles.py -q -x 2>&1 | tail -3
timeout 300 python tools/fuzz_rollout.py 100 20260929 2>&1 | tail -2
