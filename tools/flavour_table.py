#!/usr/bin/env python3
"""Rollout time by kernel flavour and batch size, in units of one CU share (51 tiles per member at E = 5, 256 CUs): the table the
launcher's plan costs (CADM_COST_*, rollout_xdl.h: xdl_launch) were read from.   python tools/flavour_table.py [max_units]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from cadm_amd import _lib, synth

cfg = dict(synth.CONFIGS["cfg2"])
maxu = int(sys.argv[1]) if len(sys.argv) > 1 else 10
libp = os.path.join(ROOT, "cadm_amd", sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "-" else None      # (a variant build, tools/build_variant.sh)
ENV = sys.argv[3] if len(sys.argv) > 3 else cfg["env"]      # python tools/flavour_table.py 6 - slim_humanoid
HID = int(sys.argv[4]) if len(sys.argv) > 4 else 200
prob = synth.make_problem(env=ENV, context=cfg["context"], E=cfg["E"], m=1, H=cfg["H"], seed=0, hidden_sizes=(HID,) * 4)
print("units  tiles/member   plan     coop-1   coop-2   wave-8   wave-4   (us per rollout)")
for u in range(1, maxu + 1):
    n = 204 * u                      # n / 4 tiles per member
    row = []
    for fl in (0, 1, 2, 3, 4):
        eng = synth.make_engine(prob, p=cfg["p"], deterministic=False, lib=_lib.load_dev(libp) if libp else _lib.load_dev())
        eng.dev_set_rollout("xdl", row_tiles=fl)
        args = [eng._t(prob[k]) for k in ("obs", "cp_obs", "cp_act", "init_mean", "init_var")]
        for c in range(6):
            eng.cem_plan(*args, n, seed=0, call=c)
        eng.profile_enable(True)
        for c in range(10):
            eng.cem_plan(*args, n, seed=0, call=100 + c)
        torch.cuda.synchronize()
        ms, nl = eng.profile_read()
        row.append(1e3 * ms / nl)
        eng.close()
    print("%5d  %12d  %7.1f  %7.1f  %7.1f  %7.1f  %7.1f" % ((u, n // 4) + tuple(row)), flush=True)
