#!/bin/bash
# one bench pass per library, in the order given, twice:  tools/abm.sh libA.so libB.so ...   (same box, interleaved)
for i in 1 2; do
  for L in "$@"; do
    python bench.py --steps 100 --legs none --no-cpu-baseline --no-extras --lib $L | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['device_resident']['value']/1e6,1), 'M', round(d['roofline']['avg_launch_ms']*1e3,1), 'us')"
  done
done
