#!/usr/bin/env python3
"""Wall time of the fused training step alone (bench.py's train_step leg): python tools/train_bench.py [steps]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
print(json.dumps(bench.train_step_bench("cuda:0", None, steps=steps, warmup=20)))
