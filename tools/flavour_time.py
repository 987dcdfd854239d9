#!/usr/bin/env python3
"""Rollout launch time by kernel flavour (developer library; hipEvents around the rollout launches inside the library):
  python tools/flavour_time.py [cfg3 cfg5 ...]      flavours: 0 = the launcher's plan, 1 / 2 = cooperative kernel with one / two row
  tiles per workgroup, 3 / 4 = wave-tile kernel with 8 / 4 tiles per workgroup.  Prints us per rollout (all launches of one rollout summed)
  and the fraction of the 833 TFLOP/s split-f16 roofline."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from cadm_amd import _lib, synth

names = sys.argv[1:] or ["cfg3"]
for cfgname in names:
    cfg = dict(synth.CONFIGS[{"cfg5g": "cfg5", "m10": "cfg2"}.get(cfgname, cfgname)])
    if cfgname == "cfg5g":
        cfg["n"] = 1000          # one GPU's shard of cfg5
    m = 10 if cfgname == "m10" else 1
    n = cfg["n"]
    prob = synth.make_problem(env=cfg["env"], context=cfg["context"], E=cfg["E"], m=m, H=cfg["H"], seed=0)
    for rnd in range(2):
        for flavour in (0, 1, 2, 3, 4):
            eng = synth.make_engine(prob, p=cfg["p"], deterministic=cfg["deterministic"], lib=_lib.load_dev())
            eng.dev_set_rollout("xdl", row_tiles=flavour)
            args = [eng._t(prob[k]) for k in ("obs", "cp_obs", "cp_act", "init_mean", "init_var")]
            try:
                for c in range(20):
                    eng.cem_plan(*args, n, seed=0, call=c)
                eng.profile_enable(True)
                reps = 20
                for c in range(reps):
                    eng.cem_plan(*args, n, seed=0, call=200 + c)
                torch.cuda.synchronize()
                ms, nl = eng.profile_read()
                us = 1e3 * ms / nl                  # (one bracket per rollout: all of its launches)
                rows = m * n * cfg["p"]
                K0, D = prob["K0"], prob["D"]
                flops = rows * cfg["H"] * 2.0 * (K0 * 200 + 3 * 200 * 200 + 200 * 2 * D)
                print("%-6s flavour %d: %8.1f us per rollout  %7.1f M row-steps/s  %.3f of 833 TFLOP/s" % (
                    cfgname, flavour, us, rows * cfg["H"] / us, flops / (us * 1e-6) / 833.3e12), flush=True)
            except Exception as exc:
                print("%-6s flavour %d: %s" % (cfgname, flavour, str(exc)[:120]), flush=True)
            eng.close()
