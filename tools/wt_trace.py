#!/usr/bin/env python3
"""Timeline of one rollout step of the wave-tile kernel (workgroup 0, all 8 waves) from a -DCADM_WT_TRACE=<step> build of the developer library:
   tools/build_variant.sh tr "-DCADM_WT_TRACE=10 ..." && python tools/wt_trace.py libcadm_hip_var_tr.so [cfg3]"""
import ctypes as ct
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from cadm_amd import _lib, synth

lib = sys.argv[1]
cfgname = sys.argv[2] if len(sys.argv) > 2 else "cfg3"
cfg = synth.CONFIGS[cfgname]
prob = synth.make_problem(env=cfg["env"], context=cfg["context"], E=cfg["E"], m=1, H=cfg["H"], seed=0)
eng = synth.make_engine(prob, p=cfg["p"], deterministic=cfg["deterministic"], lib=_lib.load_dev(os.path.join(ROOT, "cadm_amd", lib)))
tbuf = torch.zeros(8 * 160, dtype=torch.int64, device=eng.device)
eng._check(eng.lib.cadm_dev_set_timing_buffer(eng._ctx, ct.c_void_p(tbuf.data_ptr())))
args = [eng._t(prob[k]) for k in ("obs", "cp_obs", "cp_act", "init_mean", "init_var")]
for c in range(5):
    eng.cem_plan(*args, cfg["n"], seed=0, call=c)
torch.cuda.synchronize()
t = tbuf.cpu().numpy().reshape(8, 160).astype(np.int64)
nb = 30
t0 = t[:, 156].min()
print("state phase (start, end) relative to the earliest wave; then per block: start | mfma | signal | epilogue  (cycles since step start; durations)")
print("wave  state_start state_end")
for w in range(8):
    print("%4d  %10d %9d" % (w, t[w, 156] - t0, t[w, 157] - t0))
print("block " + " ".join("   w%d:st  mf  sg  ep" % w for w in range(8)))
for q in range(nb):
    row = []
    for w in range(8):
        st, mf, sg, ep = t[w, 4 * q:4 * q + 4]
        row.append("%7d %4d %3d %3d" % (st - t0, mf - st, sg - mf, ep - sg))
    print("%5d " % q + " ".join(row))
ends = t[:, 4 * (nb - 1) + 3] - t0
print("step end per wave:", ends)
blk = np.diff(t[:, 0:4 * nb:4], axis=1)
print("mean block period per wave (7-chunk hidden blocks 8..26):", blk[:, 8:26].mean(1).round(0))
