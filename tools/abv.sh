#!/bin/bash
# tools/abv.sh ROUNDS lib1 lib2 ...: same-box comparison of builds by bench.py's HEADLINE (get_action through the class, ms per call) and the device-resident protocol
N=$1; shift
for i in $(seq $N); do
  for L in "$@"; do
    python bench.py --steps 200 --warmup 20 --legs none --no-extras --no-cpu-baseline --lib cadm_amd/$L 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L  class %.4f ms  device-resident %.4f ms  rollout %.1f us' % (d['ms_per_step'], d['device_resident']['device_ms_per_get_action'], d['roofline']['avg_launch_ms']*1e3))"
  done
done
