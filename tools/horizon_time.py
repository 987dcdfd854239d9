#!/usr/bin/env python3
"""Rollout launch time against the horizon at cfg2's batch (hipEvents inside the library): t(H) = a + b H -- `a` is what a LAUNCH costs
besides its steps (dispatch, the workgroup prologue: LDS zero fill, bias tiles, resident / LDS-resident fragments, tables; the return
reduction), `b` one step.   python tools/horizon_time.py [cfg2|cfg3]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from cadm_amd import synth

cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
cfg = synth.CONFIGS[cfgname]
res = []
for H in (1, 2, 5, 10, 20, 30):
    prob = synth.make_problem(env=cfg["env"], context=cfg["context"], E=cfg["E"], m=1, H=H, seed=0)
    eng = synth.make_engine(prob, p=cfg["p"], deterministic=cfg["deterministic"])
    args = [eng._t(prob[k]) for k in ("obs", "cp_obs", "cp_act", "init_mean", "init_var")]
    for c in range(100):
        eng.cem_plan(*args, cfg["n"], seed=0, call=c)
    eng.profile_enable(True)
    for c in range(100):
        eng.cem_plan(*args, cfg["n"], seed=0, call=200 + c)
    torch.cuda.synchronize()
    ms, n = eng.profile_read()
    res.append((H, 1e3 * ms / n))
    print("H = %2d: %.2f us per rollout launch (%d launches)" % (H, 1e3 * ms / n, n), flush=True)
    eng.close()
Hs, ts = np.array([r[0] for r in res], float), np.array([r[1] for r in res])
b, a = np.polyfit(Hs[2:], ts[2:], 1)
print("fit over H >= 5: t = %.2f us + %.3f us x H   (per-launch part %.1f %% of the H = 30 launch)" % (a, b, 100 * a / ts[-1]))
