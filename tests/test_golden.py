"""Golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py from the oracle):
CPU: the oracle still reproduces them; GPU: the HIP path matches them."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden import PLANNER_CASES, planner_case  # noqa: E402

from helpers import assert_close, make_engine, oracle_problem  # noqa: E402
from oracle import planner as oplanner  # noqa: E402

GOLD = np.load(os.path.join(HERE, "golden", "planner_golden.npz"))
TGOLD = np.load(os.path.join(HERE, "golden", "train_golden.npz"))
WD, CWD = (0.000025, 0.00005, 0.000075, 0.000075, 0.0001), (0.000025, 0.00005, 0.000075)


@pytest.mark.parametrize("name", sorted(PLANNER_CASES))
def test_oracle_reproduces_planner_golden(name):
    prob, c, z, eps = planner_case(name)
    o = oracle_problem(prob, np.float64)
    plan, info, ctx = oplanner.cem_plan(o["env"], o["ff"], o["cp"] if c["context"] else None, o["st"], o["obs"], o["cp_obs"],
                                        o["cp_act"], o["init_mean"], o["init_var"], z.astype(np.float64), eps.astype(np.float64),
                                        c["E"], c["p"], deterministic=c["det"], formulation="indexed", return_info=True)
    np.testing.assert_allclose(oplanner.get_action_clip(plan), GOLD["%s/f64/plan" % name], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(np.stack([i["cand_returns"] for i in info]), GOLD["%s/f64/cand_returns" % name], rtol=1e-9)
    # the fp32 golden sits in the fp32 band around fp64 truth
    assert np.abs(GOLD["%s/f32/plan" % name] - GOLD["%s/f64/plan" % name]).max() < 5e-4


def test_oracle_reproduces_train_golden():
    import torch
    from cadm_amd import synth
    from oracle import train as otrain
    prob = synth.make_problem(env="halfcheetah", context=True, E=3, trained_like=True, with_back=True, seed=201)
    batch = synth.make_train_batch(prob, B=24, seed=202)
    cfg = dict(deterministic=False, back_coeff=0.5, weight_decay_coeff=1.0, weight_decays=WD, context_weight_decays=CWD,
               n_hidden=4, n_cp_hidden=3)
    dt = torch.float64
    ff, back, cp = (otrain.to_torch(prob[k], dt) for k in ("ff", "back", "cp"))
    res = otrain.train_losses("halfcheetah", ff, back, cp, otrain.to_torch(prob["stats"], dt),
                              {k: torch.tensor(v, dtype=dt) for k, v in batch.items()}, cfg)
    np.testing.assert_allclose([float(res["mse"]), float(res["back_mse"]), float(res["recon"])], TGOLD["losses"], rtol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(PLANNER_CASES))
def test_hip_matches_planner_golden(gpu, name):
    from cadm_amd import planner as hplanner
    prob, c, z, eps = planner_case(name)
    eng = make_engine(prob, p=c["p"], deterministic=c["det"])
    plan, info, ctx = hplanner.cem_plan(eng, prob["obs"], prob["cp_obs"] if c["context"] else None,
                                        prob["cp_act"] if c["context"] else None, prob["init_mean"], prob["init_var"], c["n"],
                                        z=eng._t(z), eps=eng._t(eps), return_info=True)
    if c["context"]:
        assert_close(ctx.cpu().numpy(), GOLD["%s/f32/ctx" % name], 1e-5, "context vs golden")
    assert_close(info[0]["rows"].cpu().numpy(), GOLD["%s/f32/rows_it0" % name], 1e-4, "row returns (CEM iteration 0) vs golden")
    for it in range(5):
        np.testing.assert_array_equal(np.sort(info[it]["elites"].cpu().numpy(), 1), np.sort(GOLD["%s/elites" % name][it], 1))
    assert_close(plan.cpu().numpy(), GOLD["%s/f32/plan" % name], 2e-4, "plan vs fp32 golden")
    assert_close(plan.cpu().numpy(), GOLD["%s/f64/plan" % name], 5e-4, "plan vs fp64 golden")


@pytest.mark.gpu
def test_hip_matches_train_golden(gpu):
    from cadm_amd import synth
    prob = synth.make_problem(env="halfcheetah", context=True, E=3, trained_like=True, with_back=True, seed=201)
    batch = synth.make_train_batch(prob, B=24, seed=202)
    eng = make_engine(prob, p=3)
    eng.train_configure(1e6, WD, CWD, 1.0, 0.5, max_batch=24, beta1=0.0, beta2=0.0, epsilon=1e6)   # w -= ~g (see test_gpu_train)
    dev = {k: eng._t(v) for k, v in batch.items()}
    before = {n: {k: v.clone() for k, v in eng.nets[n].items()} for n in eng.net_names()}
    losses = eng.train_step(dev, train=True).cpu().numpy()
    np.testing.assert_allclose(losses, TGOLD["losses"], rtol=5e-5)
    for key in TGOLD.files:
        if not key.startswith("grad/"):
            continue
        _, net, name = key.split("/")
        g = (before[net][name] - eng.nets[net][name]).cpu().numpy()[..., :8, :16]
        ref = TGOLD[key]
        assert np.abs(g - ref).max() <= 5e-3 * max(np.abs(ref).max(), 1e-6), "%s: gradient vs golden" % key
