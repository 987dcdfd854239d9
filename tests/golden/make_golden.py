#!/usr/bin/env python3
"""Generates the committed golden fixtures (tests/golden/*.npz) from the CPU oracle.

The reference (TF 1.15) cannot run in this image and ships no fixtures, so the
vectors come from the oracle's fp64 ("truth") and fp32 ("TF-like") evaluations; they freeze the
oracle against regressions and let the GPU box check the HIP path without regenerating anything.
Inputs and the injected random draws are regenerated from seeds (cadm_amd.synth, numpy PCG64) --
only oracle OUTPUTS are stored.   usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from cadm_amd import synth  # noqa: E402
from helpers import oracle_problem, trunc_z  # noqa: E402
from oracle import nets as onets  # noqa: E402
from oracle import planner as oplanner  # noqa: E402

PLANNER_CASES = {   # name: (env, context, E, p, m, n, H, det, synth seed)
    "hc_cadm": ("halfcheetah", True, 5, 10, 2, 56, 6, False, 101),
    "hc_vanilla_det": ("halfcheetah", False, 1, 1, 1, 60, 8, True, 102),
    "humanoid_cadm": ("slim_humanoid", True, 5, 5, 1, 52, 4, False, 103),
}


def planner_case(name):
    env, context, E, p, m, n, H, det, seed = PLANNER_CASES[name]
    prob = synth.make_problem(env=env, context=context, E=E, m=m, H=H, seed=seed, trained_like=True)
    rng = np.random.default_rng(seed + 1000)
    z = trunc_z(rng, (5, m, n, H, prob["A"])).astype(np.float32)
    eps = rng.standard_normal((5, H, m, n, p, prob["D"])).astype(np.float32)
    return prob, dict(E=E, p=p, m=m, n=n, H=H, det=det, context=context), z, eps


def main():
    out = {}
    for name in PLANNER_CASES:
        prob, c, z, eps = planner_case(name)
        for tag, dt in (("f32", np.float32), ("f64", np.float64)):
            o = oracle_problem(prob, dt)
            cp = o["cp"] if c["context"] else None
            plan, info, ctx = oplanner.cem_plan(o["env"], o["ff"], cp, o["st"], o["obs"], o["cp_obs"], o["cp_act"],
                                                o["init_mean"], o["init_var"], z.astype(dt), eps.astype(dt), c["E"], c["p"],
                                                deterministic=c["det"], formulation="literal", return_info=True)
            out["%s/%s/plan" % (name, tag)] = oplanner.get_action_clip(plan)
            if ctx is not None:
                out["%s/%s/ctx" % (name, tag)] = ctx
            out["%s/%s/cand_returns" % (name, tag)] = np.stack([i["cand_returns"] for i in info])
            out["%s/%s/rows_it0" % (name, tag)] = info[0]["returns"]
            if tag == "f32":
                out["%s/elites" % name] = np.stack([i["elites"] for i in info]).astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "planner_golden.npz"), **out)
    print("wrote planner_golden.npz:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "planner_golden.npz")) // 1024, "KiB")

    # training step: losses + a handful of gradient tensors (fp64 autograd)
    import torch
    from oracle import train as otrain
    tout = {}
    WD = (0.000025, 0.00005, 0.000075, 0.000075, 0.0001)
    CWD = (0.000025, 0.00005, 0.000075)
    prob = synth.make_problem(env="halfcheetah", context=True, E=3, trained_like=True, with_back=True, seed=201)
    batch = synth.make_train_batch(prob, B=24, seed=202)
    cfg = dict(deterministic=False, back_coeff=0.5, weight_decay_coeff=1.0, weight_decays=WD, context_weight_decays=CWD,
               n_hidden=4, n_cp_hidden=3)
    dt = torch.float64
    ff, back, cp = (otrain.to_torch(prob[k], dt, True) for k in ("ff", "back", "cp"))
    st = otrain.to_torch(prob["stats"], dt)
    tb = {k: torch.tensor(v, dtype=dt) for k, v in batch.items()}
    res = otrain.train_losses("halfcheetah", ff, back, cp, st, tb, cfg)
    g = otrain.grads_of(res["loss"], {"ff_model": ff, "backward_model": back, "context_model": cp})
    tout["losses"] = np.array([float(res["mse"].detach()), float(res["back_mse"].detach()), float(res["recon"].detach())])
    for net, name in (("ff_model", "hidden_0_weight"), ("ff_model", "output_logvar_bias"), ("ff_model", "max_logvar"),
                      ("backward_model", "hidden_3_bias"), ("context_model", "cp_hidden_0_weight"), ("context_model", "cp_output_bias")):
        tout["grad/%s/%s" % (net, name)] = g[net][name].numpy()[..., :8, :16]   # corner slice keeps the fixture small
    np.savez_compressed(os.path.join(HERE, "train_golden.npz"), **tout)
    print("wrote train_golden.npz:", os.path.getsize(os.path.join(HERE, "train_golden.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
