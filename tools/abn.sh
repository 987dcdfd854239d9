#!/bin/bash
# tools/abn.sh ROUNDS "bench args" lib1 lib2 ... : interleaved same-box comparison of several builds of the developer library
N=$1; ARGS=$2; shift 2
for i in $(seq $N); do
  for L in "$@"; do
    python bench.py --steps 40 --legs none --no-extras --lib $L $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['device_resident']['value']/1e6,1), 'M', round(d['roofline']['avg_launch_ms']*1e3,1), 'us')"
  done
done
