#!/usr/bin/env python3
"""Upper bound for folding the context features of layer 0 into a per-(member, context slot) bias: the same planner shapes with a vanilla
(no context) model -- K0 = 24 = ONE chunk in layer 0 instead of two.  python tools/ctx_fold_bound.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from cadm_amd import _lib, synth

for name, n, m in (("cfg2", 200, 1), ("cfg3", 2000, 1), ("cfg5/GPU", 1000, 1)):
    for rnd in range(2):
        for context in (True, False):
            prob = synth.make_problem(env="halfcheetah", context=context, E=5, m=m, H=30, seed=0)
            eng = synth.make_engine(prob, p=20, deterministic=False, lib=_lib.load_dev())
            args = [None if prob.get(k) is None else eng._t(prob[k]) for k in ("obs", "cp_obs", "cp_act", "init_mean", "init_var")]
            for c in range(20):
                eng.cem_plan(*args, n, seed=0, call=c)
            torch.cuda.synchronize()
            eng.profile_enable(True)
            for c in range(40):
                eng.cem_plan(*args, n, seed=0, call=100 + c)
            torch.cuda.synchronize()
            ms, nl = eng.profile_read()
            print("%-9s context=%-5s %8.1f us per rollout" % (name, context, 1e3 * ms / nl), flush=True)
            eng.close()
