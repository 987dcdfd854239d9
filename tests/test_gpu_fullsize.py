"""BASELINE.json's full sizes (cfg1..cfg4; cfg5's per-GPU share is cfg3's shape) through size-independent
properties: rows are independent, so (i) a random subset of candidates re-run through the CPU oracle must reproduce the
corresponding returns of the full launch, (ii) candidate shards compose bit for bit, (iii) the fused planner is
deterministic and equals the stepwise one, (iv) permuting candidates permutes the returns (noise-free model)."""
import numpy as np
import pytest
import torch

from cadm_amd import planner as hplanner
from cadm_amd import synth
from helpers import make_engine, oracle_problem, rel_err
from oracle import nets as onets
from oracle import planner as oplanner

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("cfgname", ["cfg1", "cfg2", "cfg3", "cfg4"])
def test_full_size_rollout_subset_vs_oracle_and_shards(gpu, cfgname):
    cfg = synth.CONFIGS[cfgname]
    E, p, n, H, det = cfg["E"], cfg["p"], cfg["n"], cfg["H"], cfg["deterministic"]
    prob = synth.make_problem(env=cfg["env"], context=cfg["context"], E=E, m=1, H=H, trained_like=True, seed=77)
    eng = make_engine(prob, p=p, deterministic=det)
    rng = np.random.default_rng(5)
    ctx = eng.context_forward(prob["cp_obs"], prob["cp_act"]) if cfg["context"] else None
    mean, var = eng._t(prob["init_mean"]), eng._t(prob["init_var"])
    actions = eng.sample_actions(mean, var, n, seed=3, call=9, it=0)                    # [1, n, H, A] on the device
    eps = torch.randn((H, 1, n, p, prob["D"]), device=eng.device)
    rows = eng.rollout_returns(prob["obs"], ctx, actions, eps=eps, it=0)                # [1, n, p] full launch
    assert rows.shape == (1, n, p) and torch.isfinite(rows).all()
    # (i) a subset of candidates through the fp32 / fp64 oracle
    sub = np.sort(rng.choice(n, size=min(12, n), replace=False))
    a_sub, e_sub = _np(actions)[:, sub], _np(eps)[:, :, sub]
    res = {}
    for dt in (np.float32, np.float64):
        o = oracle_problem(prob, dt)
        T = None
        if cfg["context"]:
            T = oplanner.context_table_indexed(onets.context_forward(o["cp"], o["cp_obs"], o["cp_act"], o["st"]), 0)
        res[dt] = oplanner.rollout_indexed(o["env"], o["ff"], o["st"], o["obs"], T, a_sub.astype(dt), e_sub.astype(dt), E, p, det)
    got = _np(rows)[:, sub]
    band = max(8 * rel_err(res[np.float32], res[np.float64]), 2e-5)
    assert rel_err(got, res[np.float32]) <= band, "returns of %d sampled candidates off: %.2e (band %.2e)" % (
        len(sub), rel_err(got, res[np.float32]), band)
    # (ii) two candidate shards (as two ranks would run them) compose bit for bit
    if n % 2 == 0:
        h = n // 2
        parts = [eng.rollout_returns(prob["obs"], ctx, actions, eps=eps[:, :, g * h:(g + 1) * h].contiguous(), it=0,
                                     cand_offset=g * h, n_local=h) for g in range(2)]
        np.testing.assert_array_equal(_np(torch.cat(parts, dim=1)), _np(rows))


@pytest.mark.parametrize("cfgname", ["cfg2", "cfg4"])
def test_full_size_plan_is_deterministic_and_equals_stepwise(gpu, cfgname):
    cfg = synth.CONFIGS[cfgname]
    prob = synth.make_problem(env=cfg["env"], context=cfg["context"], E=cfg["E"], m=1, H=cfg["H"], trained_like=True, seed=78)
    eng = make_engine(prob, p=cfg["p"], deterministic=cfg["deterministic"])
    args = (prob["obs"], prob["cp_obs"], prob["cp_act"], prob["init_mean"], prob["init_var"], cfg["n"])
    a = eng.cem_plan(*args, seed=11, call=4)
    b = eng.cem_plan(*args, seed=11, call=4)
    c = hplanner.cem_plan(eng, *args, seed=11, call=4)
    d = eng.cem_plan(*args, seed=11, call=5)
    np.testing.assert_array_equal(_np(a), _np(b))          # same (seed, call) -> same plan, run to run
    np.testing.assert_array_equal(_np(a), _np(c))          # one C call == the per-iteration composition
    assert not np.array_equal(_np(a), _np(d))              # a new call draws new candidates
    assert np.abs(_np(a)).max() <= 1.0


def test_full_size_candidate_permutation_equivariance(gpu):
    """cfg2 shape, noise-free: every row is independent, so permuting the candidates permutes the returns exactly."""
    cfg = synth.CONFIGS["cfg2"]
    prob = synth.make_problem(env=cfg["env"], context=True, E=cfg["E"], m=1, H=cfg["H"], trained_like=True, seed=79)
    eng = make_engine(prob, p=cfg["p"], deterministic=True)
    ctx = eng.context_forward(prob["cp_obs"], prob["cp_act"])
    actions = eng.sample_actions(eng._t(prob["init_mean"]), eng._t(prob["init_var"]), cfg["n"], seed=1, call=1, it=0)
    perm = torch.randperm(cfg["n"], device=eng.device)
    r0 = eng.rollout_returns(prob["obs"], ctx, actions, it=0)
    r1 = eng.rollout_returns(prob["obs"], ctx, actions[:, perm].contiguous(), it=0)
    np.testing.assert_array_equal(_np(r0[:, perm]), _np(r1))
