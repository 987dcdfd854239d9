#!/usr/bin/env python3
"""Training step time by batch size and chain-kernel flavour (8 waves x 1 workgroup per CU / 4 waves x 3 per CU / 0 = the launcher's rule),
same box, developer library: python tools/train_ab.py [lib.so]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import time
import torch
from cadm_amd import _lib, synth
lib = _lib.load_dev(os.path.join(ROOT, "cadm_amd", sys.argv[1])) if len(sys.argv) > 1 else _lib.load_dev()
WD, CWD = (0.000025, 0.00005, 0.000075, 0.000075, 0.0001), (0.000025, 0.00005, 0.000075)
BS = [int(x) for x in os.environ.get("TRAIN_AB_B", "256,512,1024,4096").split(",")]
FLS = [int(x) for x in os.environ.get("TRAIN_AB_FL", "8,4,0").split(",")]       # + 16: work items spread over all XCDs, + 32: member-affine
for B in BS:
    prob = synth.make_problem(env="halfcheetah", context=True, E=5, with_back=True, seed=0)
    for fl in FLS:
        eng = synth.make_engine(prob, p=20, lib=lib)
        eng._check(lib.cadm_dev_set_train_flavour(eng._ctx, fl), "flavour")
        eng.train_configure(1e-3, WD, CWD, 1.0, 0.5, max_batch=B)
        batch = {k: eng._t(v) for k, v in synth.make_train_batch(prob, B=B, seed=1).items()}
        for _ in range(300):
            eng.train_step(batch, train=True)
        torch.cuda.synchronize()
        steps = 100 if B <= 1024 else 40
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.train_step(batch, train=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        tf = 3 * 2 * 5 * B * (2 * 134000 + 103040) / dt / 1e12
        print("B=%5d  flavour %d  %.4f ms/step  %.1f TFLOP/s (%.3f of the fp32 matrix peak)" % (B, fl, dt * 1e3, tf, tf / 157.3), flush=True)
        eng.close()
