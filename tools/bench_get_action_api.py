#!/usr/bin/env python3
"""User-facing latency of MLPEnsembleCEMDynamicsModel.get_action (numpy in, numpy out; cfg2 sizes) next to the
device-resident planner call that bench.py times.  python tools/bench_get_action_api.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from cadm_amd.dynamics.mlp_cadm_ensemble_cem_dynamics import MLPEnsembleCEMDynamicsModel
from cadm_amd.envs import EnvSpec


def main():
    env = EnvSpec("halfcheetah")
    model = MLPEnsembleCEMDynamicsModel("dyn", env, n_candidates=200, n_particles=20, ensemble_size=5, use_cem=True,
                                        n_forwards=30, back_coeff=0.5)
    r = np.random.default_rng(0)
    D, A, Hh = 18, 6, 10
    stats = [r.standard_normal(18), r.uniform(0.5, 2, 18), r.standard_normal(6), r.uniform(0.5, 2, 6), r.standard_normal(18),
             r.uniform(0.5, 2, 18), r.standard_normal(180), r.uniform(0.5, 2, 180), r.standard_normal(60), r.uniform(0.5, 2, 60),
             r.standard_normal(18), r.uniform(0.5, 2, 18)]
    keys = ("obs", "act", "delta", "cp_obs", "cp_act", "back_delta")
    model.set_normalization({k: (stats[2 * i], stats[2 * i + 1]) for i, k in enumerate(keys)})
    obs = r.standard_normal((1, D))
    cp_obs, cp_act = 0.1 * r.standard_normal((1, D * Hh)), r.uniform(-1, 1, (1, A * Hh))
    mean, var = np.zeros((1, 30, A)), np.full((1, 30, A), 0.25)
    for _ in range(5):
        model.get_action(obs, cp_obs, cp_act, mean, var)
    N = 200
    t0 = time.perf_counter()
    for _ in range(N):
        plan = model.get_action(obs, cp_obs, cp_act, mean, var)
        mean = np.concatenate([plan[:, 1:], np.zeros((1, 1, A))], axis=1)     # sampler.py:118-120 warm start
    dt_api = (time.perf_counter() - t0) / N
    eng = model.engine
    dobs, dcpo, dcpa, dmean, dvar = (eng._t(x) for x in (obs, cp_obs, cp_act, mean, var))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        eng.cem_plan(dobs, dcpo, dcpa, dmean, dvar, 200, seed=0, call=i)
    torch.cuda.synchronize()
    dt_dev = (time.perf_counter() - t0) / N
    print("get_action (numpy in/out): %.3f ms   device-resident cem_plan: %.3f ms   host overhead %.3f ms"
          % (dt_api * 1e3, dt_dev * 1e3, (dt_api - dt_dev) * 1e3))


if __name__ == "__main__":
    main()
