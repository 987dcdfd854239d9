#!/usr/bin/env python3
"""Rollout time of library VARIANTS (tools/build_variant.sh) with a forced kernel flavour (tools/flavour_time.py's numbering):
  python tools/variant_flavour_time.py cfg3 3,4 lib1.so lib2.so ...     (paths relative to cadm_amd/; timing-experiment builds are wrong by construction)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from cadm_amd import _lib, synth

cfgname, flavours = sys.argv[1], [int(x) for x in sys.argv[2].split(",")]
cfg = dict(synth.CONFIGS[{"cfg5g": "cfg5", "m10": "cfg2", "full8": "cfg3", "full4": "cfg3"}.get(cfgname, cfgname)])
if cfgname == "cfg5g":
    cfg["n"] = 1000
if cfgname == "full8":
    cfg["n"] = 1632          # 408 tiles per member (n / 4): exactly one 8-tile round of 51 workgroups per member
if cfgname == "full4":
    cfg["n"] = 816           # 204 tiles per member: one 4-tile round
m = 10 if cfgname == "m10" else 1
prob = synth.make_problem(env=cfg["env"], context=cfg["context"], E=cfg["E"], m=m, H=cfg["H"], seed=0)
for rnd in range(2):
    for path in sys.argv[3:]:
        for fl in flavours:
            eng = synth.make_engine(prob, p=cfg["p"], deterministic=cfg["deterministic"], lib=_lib.load_dev(os.path.join(ROOT, "cadm_amd", path)))
            eng.dev_set_rollout("xdl", row_tiles=fl)
            args = [eng._t(prob[k]) for k in ("obs", "cp_obs", "cp_act", "init_mean", "init_var")]
            for c in range(10):
                eng.cem_plan(*args, cfg["n"], seed=0, call=c)
            eng.profile_enable(True)
            for c in range(20):
                eng.cem_plan(*args, cfg["n"], seed=0, call=200 + c)
            torch.cuda.synchronize()
            ms, nl = eng.profile_read()
            print("%-34s %s flavour %d: %8.1f us per rollout" % (path, cfgname, fl, 1e3 * ms / nl), flush=True)
            eng.close()
