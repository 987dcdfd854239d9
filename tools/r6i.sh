D=cadm_amd/libcadm_hip_dev.so; A=cadm_amd/libcadm_hip_var_res15x.so; B=cadm_amd/libcadm_hip_var_res16x.so
for cfg in cfg2 cfg5 cfg4; do
for i in 1 2; do for L in $D $A $B; do
python bench.py --steps 100 --legs none --no-cpu-baseline --no-extras --lib $L --config $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg $L', round(d['device_resident']['value']/1e6,1), 'M', round(d['roofline']['avg_launch_ms']*1e3,1), 'us')"
done; done; done
