#!/usr/bin/env python3
"""Average rollout launch time (hipEvents inside the library) of library builds, interleaved on one box -- for the timing
EXPERIMENT variants of tools/build_variant.sh, whose results are wrong by construction (bench.py would refuse them):
  python tools/variant_time.py [cfg2|cfg3] lib1.so lib2.so ...       (paths relative to cadm_amd/)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from cadm_amd import _lib, synth

cfgname = sys.argv[1]
cfg = synth.CONFIGS[cfgname]
prob = synth.make_problem(env=cfg["env"], context=cfg["context"], E=cfg["E"], m=1, H=cfg["H"], seed=0)
for rnd in range(2):
    for path in sys.argv[2:]:
        eng = synth.make_engine(prob, p=cfg["p"], deterministic=cfg["deterministic"], lib=_lib.load_dev(os.path.join(ROOT, "cadm_amd", path)))
        args = [eng._t(prob[k]) for k in ("obs", "cp_obs", "cp_act", "init_mean", "init_var")]
        for c in range(150):
            eng.cem_plan(*args, cfg["n"], seed=0, call=c)
        eng.profile_enable(True)
        for c in range(60):
            eng.cem_plan(*args, cfg["n"], seed=0, call=200 + c)
        torch.cuda.synchronize()
        ms, n = eng.profile_read()
        print("%-36s %s rollout %.1f us per launch (%d launches)" % (path, cfgname, 1e3 * ms / n, n))
        eng.close()
