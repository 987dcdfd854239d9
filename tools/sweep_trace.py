#!/usr/bin/env python3
"""Raw time stamps of ONE hidden-layer sweep (step 10, hidden layer 2, workgroup 0) of the rollout kernel, per wave, relative to
the earliest wave's sweep entry (CADM_PHASE_TIMING build: make -C cadm_amd/csrc timing).
   python tools/sweep_trace.py [cfg] [lib]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as ct

import numpy as np
import torch

from cadm_amd import _lib, synth
from cadm_amd.synth import make_engine

cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
libname = sys.argv[2] if len(sys.argv) > 2 else "libcadm_hip_timing.so"
cfg = synth.CONFIGS[cfgname]
prob = synth.make_problem(env=cfg["env"], context=cfg["context"], E=cfg["E"], m=1, H=cfg["H"], seed=0)
eng = make_engine(prob, p=cfg["p"], deterministic=cfg["deterministic"], lib=_lib.load_dev(os.path.join(ROOT, "cadm_amd", libname)))
tbuf = torch.zeros(8 * 24 + 8 * 16, dtype=torch.int64, device=eng.device)
eng._check(eng.lib.cadm_dev_set_timing_buffer(eng._ctx, ct.c_void_p(tbuf.data_ptr())))
mean, var = eng._t(prob["init_mean"]), eng._t(prob["init_var"])
ctx = eng.context_forward(prob["cp_obs"], prob["cp_act"]) if cfg["context"] else None
acts = eng.sample_actions(mean, var, cfg["n"], seed=1, call=1, it=0)
for _ in range(3):
    eng.rollout_returns(prob["obs"], ctx, acts, seed=1, call=1)
torch.cuda.synchronize()
t = tbuf.cpu().numpy()[8 * 24:].reshape(8, 16).astype(np.int64)
t0 = t[:, 0].min()
names = ["entry"] + ["chunk %d" % c for c in range(8)] + ["partials in", "sweep end", "barrier out"]
print("s_memtime ticks after the first wave entered the sweep (last group's chunk positions; 0 = not stamped)")
print("%-12s" % "event" + "".join(" %7s" % ("wave%d" % w) for w in range(8)))
for i, nm in enumerate(names):
    print("%-12s" % nm + "".join(" %7d" % (t[w, i] - t0 if t[w, i] else 0) for w in range(8)))
