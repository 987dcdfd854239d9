#!/usr/bin/env python3
"""Wall time of the fused training step alone (bench.py's train_step leg): python tools/train_bench.py [steps] [lib ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from cadm_amd import _lib

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
libs = sys.argv[2:] or [None]
for rnd in range(3 if len(libs) > 1 else 1):
    for path in libs:
        lib = _lib.load_dev(os.path.join(ROOT, "cadm_amd", path)) if path else None
        r = bench.train_step_bench("cuda:0", lib, steps=steps, warmup=20)
        print(path or "product", json.dumps(r))
