#!/usr/bin/env python3
"""Bitwise comparison of two builds of the library on the same inputs (developer tool; needs a GPU):
   python tools/compare_libs.py libA.so libB.so      -- rollouts (returns + trajectories) and whole CEM plans over a set of
problem shapes.  A schedule-only change of the rollout kernel must print 'identical' everywhere."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from cadm_amd import _lib, synth

CASES = [  # env, context, hid, E, p, m, n, H, det
    ("halfcheetah", True, 200, 5, 20, 1, 200, 30, False),
    ("halfcheetah", True, 200, 5, 20, 1, 2000, 8, False),
    ("halfcheetah", True, 200, 5, 20, 3, 37, 7, False),
    ("halfcheetah", False, 200, 1, 1, 1, 200, 30, True),
    ("slim_humanoid", True, 200, 5, 20, 1, 1000, 6, False),
    ("slim_humanoid", True, 200, 5, 10, 2, 9, 5, False),
    ("ant", True, 128, 5, 5, 2, 60, 6, False),
    ("pendulum", True, 256, 5, 5, 2, 500, 6, False),
    ("cartpole", True, 256, 2, 4, 1, 33, 9, False),
    ("halfcheetah", True, 512, 5, 5, 1, 40, 4, False),
]


def main():
    libs = [_lib.load_dev(os.path.abspath(p)) for p in sys.argv[1:3]]
    bad = 0
    for env, context, hid, E, p, m, n, H, det in CASES:
        prob = synth.make_problem(env=env, context=context, E=E, m=m, H=H, hidden_sizes=(hid,) * 4, trained_like=True, seed=5)
        rng = np.random.default_rng(1)
        if prob["discrete"]:
            acts = np.eye(prob["A"], dtype=np.float32)[rng.integers(0, prob["A"], (m, n, H))]
        else:
            acts = rng.uniform(-1, 1, (m, n, H, prob["A"])).astype(np.float32)
        eps = rng.standard_normal((H, m, n, p, prob["D"])).astype(np.float32)
        outs = []
        for lib in libs:
            eng = synth.make_engine(prob, p=p, deterministic=det, lib=lib)
            ctx = eng.context_forward(prob["cp_obs"], prob["cp_act"]) if context else None
            res = []
            for kw in (dict(eps=None if det else eps), dict(seed=3, call=7, it=1)):
                rows, traj = eng.rollout_returns(prob["obs"], ctx, acts, want_traj=True, norm_actions=not prob["discrete"], **kw)
                res += [rows.cpu().numpy(), traj.cpu().numpy()]
            if not prob["discrete"] and n >= 50:
                res.append(eng.cem_plan(prob["obs"], prob["cp_obs"] if context else None, prob["cp_act"] if context else None,
                                        prob["init_mean"], prob["init_var"], n, seed=1, call=2).cpu().numpy())
            torch.cuda.synchronize()
            outs.append(res)
            eng.close()
        same = all(np.array_equal(x, y, equal_nan=True) for x, y in zip(*outs))
        finite = all(np.isfinite(x).all() for x in outs[1])
        bad += 0 if (same and finite) else 1
        print("%-14s ctx=%d hid=%d E=%d p=%d m=%d n=%d H=%d det=%d: %s%s" % (env, context, hid, E, p, m, n, H, det,
              "identical" if same else "DIFFERENT", "" if finite else " (non-finite!)"))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
