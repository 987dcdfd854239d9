"""Determinism / stability soak: repeated plans must be bit-identical, twin engines trained on the same batches must stay
bit-identical (ragged batch sizes included).  python tools/soak.py [seconds] [candidates]
(candidates >= 1000 exercises the two-row-tile rollout flavour and its split launches)"""
import sys, os, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from cadm_amd import synth
from cadm_amd.synth import make_engine
t_end = time.time() + float(sys.argv[1]) if len(sys.argv) > 1 else time.time() + 30
prob = synth.make_problem(env="halfcheetah", context=True, E=5, m=1, H=30, trained_like=True, with_back=True, seed=3)
engA, engB = make_engine(prob, p=20), make_engine(prob, p=20)
WD = (0.000025, 0.00005, 0.000075, 0.000075, 0.0001); CWD = (0.000025, 0.00005, 0.000075)
for e in (engA, engB): e.train_configure(1e-3, WD, CWD, 1.0, 0.5, max_batch=256)
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 200
args = (prob["obs"], prob["cp_obs"], prob["cp_act"], prob["init_mean"], prob["init_var"], NC)
ref_plan = engA.cem_plan(*args, seed=7, call=1).clone()
it = 0; nplan = 0; ntrain = 0
rng = np.random.default_rng(0)
while time.time() < t_end:
    it += 1
    for _ in range(20):
        p1 = engA.cem_plan(*args, seed=7, call=1); nplan += 1
    assert torch.equal(p1, ref_plan), "plan changed at iteration %d" % it
    B = int(rng.choice([256, 100, 17, 256]))
    batch = synth.make_train_batch(prob, B=B, seed=it)
    for e in (engA, engB):
        bt = {k: e._t(v) for k, v in batch.items()}
        for _ in range(10):
            l = e.train_step(bt, train=True)
    ntrain += 10
    la = l
    for n in engA.net_names():
        for k in engA.nets[n]:
            assert torch.equal(engA.nets[n][k], engB.nets[n][k]), "training diverged between twin engines at %d: %s/%s" % (it, n, k)
    assert torch.isfinite(la).all()
    engA.repack(); ref_plan = engA.cem_plan(*args, seed=7, call=1).clone()      # weights changed: new reference plan
torch.cuda.synchronize()
print("soak OK (%d candidates): %d iterations, %d plans, %d train steps per engine, deterministic throughout" % (NC, it, nplan, ntrain))
