mkdir -p gpurun_out/r6g
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r6g/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6g/pytest.log
grep -E "passed|failed|FAILED|Error|rc" gpurun_out/r6g/pytest.log | head -20
timeout 300 python tools/fuzz_rollout.py 60 7 2>&1 | tail -1
bash tools/kstats.sh --steps 100 > gpurun_out/r6g/kstats_cfg2.txt 2>&1; cat gpurun_out/r6g/kstats_cfg2.txt | head -9
python bench.py --legs cfg3,cfg4,cfg5,m10 --no-cpu-baseline > gpurun_out/r6g/bench.json 2> gpurun_out/r6g/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6g/bench.json').read().strip().splitlines()[-1])
print('headline', d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])
for k,v in d.get('legs',{}).items(): print(k, {kk: (round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('value','roofline_frac','avg_launch_ms','ms_per_get_action','device_ms_per_get_action')})
print({k: v for k,v in d.get('train_step',{}).items()} if 'train_step' in d else '')
PY
