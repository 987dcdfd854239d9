#!/usr/bin/env python3
"""Differential check of the two rollout kernels on the same inputs: the split-f16 "xdl" kernel (production) against the
fp32-MFMA comparison kernel of the developer library (cadm_dev_set_rollout), per problem variant.  Developer tool; needs a GPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from cadm_amd import _lib, synth


def engines(prob, p, det=False):
    e32 = synth.make_engine(prob, p=p, deterministic=det, lib=_lib.load_dev())
    e32.dev_set_rollout("f32")
    ex = synth.make_engine(prob, p=p, deterministic=det)          # the product library
    return e32, ex


def main():
    rng = np.random.default_rng(0)
    for env, ctx, E, p in (("halfcheetah", True, 5, 20), ("halfcheetah", False, 1, 1), ("slim_humanoid", True, 5, 20), ("pendulum", True, 5, 5)):
        for trained in (False, True):
            for H in (1, 30):
                prob = synth.make_problem(env=env, context=ctx, E=E, m=1, H=H, trained_like=trained, seed=3)
                if os.environ.get("QW"):      # weights exactly representable in f16: the low weight part is all zeros
                    for k, v in prob["ff"].items():
                        if k.endswith("_weight"):
                            prob["ff"][k] = v.astype(np.float16).astype(np.float64)
                e32, ex = engines(prob, p, det=(E == 1))
                n = 48
                acts = rng.uniform(-1, 1, (1, n, H, prob["A"]))
                eps = rng.standard_normal((H, 1, n, p, prob["D"]))
                out = []
                for eng in (e32, ex):
                    c = eng.context_forward(prob["cp_obs"], prob["cp_act"]) if ctx else None
                    rows, traj = eng.rollout_returns(prob["obs"], c, acts, eps=None if E == 1 else eps, want_traj=True)
                    out.append((rows.cpu().numpy(), traj.cpu().numpy()))
                (r0, t0), (r1, t1) = out
                rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
                print("%-14s ctx=%d trained_like=%d H=%2d: returns rel %.2e  first-step obs rel %.2e  last-step obs rel %.2e"
                      % (env, ctx, trained, H, rel(r1, r0), rel(t1[0], t0[0]), rel(t1[-1], t0[-1])))
                e32.close(); ex.close()


if __name__ == "__main__":
    main()
