#!/bin/bash
# A/B two builds of the library on the SAME box (box-to-box variance is ~2 %): tools/ab.sh libA.so libB.so [rounds] [bench args]
# prints the headline value and the rollout kernel's average launch time per run, interleaved A B A B ...
# (bench.py --lib PATH binds the run to that build through cadm_amd._lib.load_dev; the default run binds the product)
A=$1; B=$2; N=${3:-3}; shift 3
for i in $(seq $N); do
  for L in $A $B; do
    python bench.py --steps 100 --legs none --no-cpu-baseline --no-extras --lib $L "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['device_resident']['value']/1e6,1), 'M', round(d['roofline']['avg_launch_ms']*1e3,1), 'us')"
  done
done
