#!/usr/bin/env python3
"""cem_refit / sample_actions kernel time vs candidate count (what every rank runs on the GLOBAL n when sharded)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from cadm_amd import synth
from cadm_amd.synth import make_engine

prob = synth.make_problem(env="halfcheetah", context=True, E=5, seed=0)
eng = make_engine(prob, p=20)
for n in (200, 400, 800, 1600, 2000, 4000, 8000):
    mean, var = eng._t(prob["init_mean"]), eng._t(prob["init_var"])
    acts = eng.sample_actions(mean, var, n, seed=1, call=1, it=0)
    cand = torch.randn((1, n), device=eng.device)
    for _ in range(5):
        eng.cem_refit(cand, acts, mean, var)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        eng.cem_refit(cand, acts, mean, var)
    torch.cuda.synchronize(); tr = (time.perf_counter() - t0) / 200
    t0 = time.perf_counter()
    for _ in range(200):
        eng.sample_actions(mean, var, n, seed=1, call=1, it=0)
    torch.cuda.synchronize(); ts = (time.perf_counter() - t0) / 200
    G = 8
    cg = cand.view(G, 1, n // G).contiguous()
    t0 = time.perf_counter()
    for _ in range(200):
        eng.cem_refit(cg, acts, mean, var, G=G, regen=(1, 1, 0))
    torch.cuda.synchronize(); trg = (time.perf_counter() - t0) / 200
    t0 = time.perf_counter()
    for _ in range(200):
        eng.sample_actions(mean, var, n, seed=1, call=1, it=0, cand_offset=n // G, n_local=n // G)
    torch.cuda.synchronize(); tsl = (time.perf_counter() - t0) / 200
    print("n=%5d  refit %.1f us   sample all %.1f us  |  8-way shard: refit with regenerated elites %.1f us   sample own shard %.1f us"
          % (n, tr * 1e6, ts * 1e6, trg * 1e6, tsl * 1e6))
