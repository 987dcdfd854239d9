#!/usr/bin/env python3
"""Is the launcher's plan (flavour 0) as fast as the best single rollout flavour on shapes the cost table was NOT measured on?
   python tools/plan_check.py      -> one line per shape: us per rollout for the plan and for forced flavours 1 / 2 / 3"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from cadm_amd import _lib, synth

SHAPES = [      # env, context, E, p, n, m, hidden, deterministic
    ("halfcheetah", False, 1, 1, 30000, 1, 200, True),       # vanilla: one member owns all 256 CUs; 1875 tiles
    ("halfcheetah", False, 1, 1, 40000, 1, 200, True),       # 2500 tiles
    ("halfcheetah", True, 5, 20, 3000, 1, 200, False),       # 750 tiles per member
    ("halfcheetah", True, 5, 20, 1400, 1, 200, False),       # 350
    ("halfcheetah", True, 5, 20, 2000, 1, 256, False),
    ("halfcheetah", True, 5, 20, 2000, 1, 128, False),
    ("ant", True, 5, 20, 2000, 1, 200, False),
    ("slim_humanoid", True, 5, 20, 2000, 1, 200, False),
    ("cartpole", True, 5, 5, 4000, 4, 200, False),
    ("halfcheetah", True, 2, 4, 10000, 1, 200, False),       # E = 2: 128 CUs per member
]
for env, context, E, p, n, m, hid, det in SHAPES:
    prob = synth.make_problem(env=env, context=context, E=E, m=m, H=10, hidden_sizes=(hid,) * 4, seed=0)
    row = []
    for fl in (0, 1, 2, 3):
        eng = synth.make_engine(prob, p=p, deterministic=det, lib=_lib.load_dev())
        eng.dev_set_rollout("xdl", row_tiles=fl)
        args = [eng._t(prob[k]) if prob.get(k) is not None else None for k in ("obs", "cp_obs", "cp_act", "init_mean", "init_var")]
        try:
            for c in range(3):
                eng.cem_plan(*args, n, seed=0, call=c)
            eng.profile_enable(True)
            for c in range(6):
                eng.cem_plan(*args, n, seed=0, call=100 + c)
            torch.cuda.synchronize()
            ms, nl = eng.profile_read()
            row.append(1e3 * ms / nl)
        except Exception as exc:
            print("   (flavour %d: %s)" % (fl, str(exc)[:160]))
            row.append(float("nan"))
        eng.close()
    ok = [x for x in row[1:] if x == x]
    best = min(ok) if ok else float("nan")
    print("%-14s ctx=%d E=%d p=%2d n=%5d m=%d hid=%3d: plan %8.1f   one-tile %8.1f  two-tile %8.1f  wave-tile %8.1f   plan / best single = %.3f" % (
        env, context, E, p, n, m, hid, row[0], row[1], row[2], row[3], row[0] / best), flush=True)
