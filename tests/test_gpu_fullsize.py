"""BASELINE.json's full sizes (cfg1..cfg5 and the reference's m = 10 launch shape) through size-independent
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


def test_cfg5_eight_candidate_shards_on_one_gpu(gpu):
    """BASELINE configs[4]: cand = 8000 sharded 1000 per GPU over 8 GPUs.  One GPU plays the 8 ranks in turn
    (`cand_offset = r * 1000, n_local = 1000`): the shards must concatenate to the unsharded launch bit for bit (injected
    AND device-drawn noise -- the Philox counters are keyed by the GLOBAL row), 12 sampled candidates must match the
    oracle, and the refit over the gathered [8, m, 1000] layout (what ncclAllGather delivers,
    /root/reference/cadm/dynamics/core/utils.py:474-486 on the global return vector) must equal the unsharded refit."""
    cfg = synth.CONFIGS["cfg5"]
    E, p, n, H, G = cfg["E"], cfg["p"], cfg["n"], cfg["H"], 8
    nl = n // G
    prob = synth.make_problem(env=cfg["env"], context=True, E=E, m=1, H=H, trained_like=True, seed=81)
    eng = make_engine(prob, p=p)
    ctx = eng.context_forward(prob["cp_obs"], prob["cp_act"])
    mean, var = eng._t(prob["init_mean"]), eng._t(prob["init_var"])
    actions = eng.sample_actions(mean, var, n, seed=5, call=2, it=0)                      # every rank draws all 8000
    eps = torch.randn((H, 1, n, p, prob["D"]), device=eng.device)
    full = eng.rollout_returns(prob["obs"], ctx, actions, eps=eps, it=0)                  # [1, 8000, 20]
    full_rng = eng.rollout_returns(prob["obs"], ctx, actions, seed=5, call=2, it=0)
    parts, parts_rng = [], []
    for r in range(G):
        parts.append(eng.rollout_returns(prob["obs"], ctx, actions, eps=eps[:, :, r * nl:(r + 1) * nl].contiguous(), it=0,
                                         cand_offset=r * nl, n_local=nl))
        parts_rng.append(eng.rollout_returns(prob["obs"], ctx, actions, seed=5, call=2, it=0, cand_offset=r * nl, n_local=nl))
    np.testing.assert_array_equal(_np(torch.cat(parts, dim=1)), _np(full))
    np.testing.assert_array_equal(_np(torch.cat(parts_rng, dim=1)), _np(full_rng))
    # sampled candidates vs the oracle
    rng = np.random.default_rng(6)
    sub = np.sort(rng.choice(n, size=12, replace=False))
    a_sub, e_sub = _np(actions)[:, sub], _np(eps)[:, :, sub]
    res = {}
    for dt in (np.float32, np.float64):
        o = oracle_problem(prob, dt)
        T = oplanner.context_table_indexed(onets.context_forward(o["cp"], o["cp_obs"], o["cp_act"], o["st"]), 0)
        res[dt] = oplanner.rollout_indexed(o["env"], o["ff"], o["st"], o["obs"], T, a_sub.astype(dt), e_sub.astype(dt), E, p, False)
    band = max(8 * rel_err(res[np.float32], res[np.float64]), 2e-5)
    assert rel_err(_np(full)[:, sub], res[np.float32]) <= band
    # refit: gathered [G, m, n_local] layout == unsharded [m, n]
    cand_full = eng.particle_mean(full)                                                   # [1, 8000]
    gathered = torch.stack([eng.particle_mean(x) for x in parts], dim=0)                  # [8, 1, 1000]
    m1, v1, m2, v2 = mean.clone(), var.clone(), mean.clone(), var.clone()
    el1 = eng.cem_refit(cand_full, actions, m1, v1, G=1, want_elites=True)
    el2 = eng.cem_refit(gathered, actions, m2, v2, G=G, want_elites=True)
    np.testing.assert_array_equal(_np(el1), _np(el2))
    np.testing.assert_array_equal(_np(m1), _np(m2))
    np.testing.assert_array_equal(_np(v1), _np(v2))
    nm, nv, idx = oplanner.elite_refit(_np(mean), _np(var), _np(actions), _np(cand_full))
    np.testing.assert_array_equal(_np(el1), idx)
    # SURVEY 8e / VERDICT r4 #7: a rank samples ONLY its own shard (the draws are keyed by the global element index: what rank r writes
    # at its positions is what the full sampler writes there) and the refit draws the elites' sequences again by global candidate id
    # instead of reading them -- bit-identical statistics without any rank ever holding another rank's candidates
    for r in (0, 3, 7):
        shard = eng.sample_actions(mean, var, n, seed=5, call=2, it=0, cand_offset=r * nl, n_local=nl)
        np.testing.assert_array_equal(_np(shard)[:, r * nl:(r + 1) * nl], _np(actions)[:, r * nl:(r + 1) * nl])
    m3, v3 = mean.clone(), var.clone()
    hollow = torch.full_like(actions, float("nan"))                                       # (the refit must not read it)
    el3 = eng.cem_refit(gathered, hollow, m3, v3, G=G, want_elites=True, regen=(5, 2, 0))
    np.testing.assert_array_equal(_np(el3), _np(el1))
    np.testing.assert_array_equal(_np(m3), _np(m1))
    np.testing.assert_array_equal(_np(v3), _np(v1))
    # a later iteration (another stream id, a non-trivial distribution): still bit-identical
    act1 = eng.sample_actions(m1, v1, n, seed=5, call=2, it=1)
    cand1 = torch.randn((1, n), device=eng.device)
    m4, v4, m5, v5 = m1.clone(), v1.clone(), m1.clone(), v1.clone()
    eng.cem_refit(cand1, act1, m4, v4, G=1)
    eng.cem_refit(cand1.view(G, 1, nl).contiguous(), hollow, m5, v5, G=G, regen=(5, 2, 1))
    np.testing.assert_array_equal(_np(m5), _np(m4))
    np.testing.assert_array_equal(_np(v5), _np(v4))


def test_sharded_refit_checks_every_ranks_input_checksum(gpu):
    """a sharded cadm_cem_plan carries the checksum of the inputs each rank was fed as one extra word of its
    all-gather payload, and the refit of EVERY iteration compares all of them: ranks that drifted apart get a NaN plan on the very
    call (the host raises on it), not up to 255 calls later.  One GPU plays the ranks: the developer library's hooks run the sharded
    refit on a fabricated all-gather result (csrc/dev/dev_api.h)."""
    from cadm_amd import _lib
    from cadm_amd._lib import ptr
    prob = synth.make_problem(env="halfcheetah", context=True, E=5, m=2, H=6, trained_like=True, seed=91)
    eng = make_engine(prob, p=10, H=6, lib=_lib.load_dev())
    G, nl, m = 4, 16, 2
    n = G * nl
    mean, var = eng._t(prob["init_mean"]), eng._t(prob["init_var"])
    obs, cpo, cpa = eng._t(prob["obs"]), eng._t(prob["cp_obs"]), eng._t(prob["cp_act"])
    word = torch.zeros(1, device=eng.device)
    chk = lambda o: (eng._check(eng.lib.cadm_dev_input_checksum(eng._ctx, ptr(o), ptr(cpo), ptr(cpa), ptr(mean), ptr(var), m, ptr(word), eng.stream)),
                     word.clone())[1]
    w_same = chk(obs)
    assert torch.equal(chk(obs.clone()), w_same)                                      # a function of the values only
    o2 = obs.clone(); o2[1, 3] = torch.nextafter(o2[1, 3], o2[1, 3] + 1)               # one bit of one observation
    w_bit = chk(o2)
    o3 = obs.clone(); o3[0, 2], o3[0, 5] = obs[0, 5], obs[0, 2]                       # a permutation (a plain sum would not see it)
    w_perm = chk(o3)
    assert w_bit.view(torch.int32).item() != w_same.view(torch.int32).item() and w_perm.view(torch.int32).item() != w_same.view(torch.int32).item()
    cand = torch.randn((G, m * nl), device=eng.device)

    def refit(words, rank):
        payload = torch.cat([cand, torch.stack(words).view(G, 1)], dim=1).contiguous()     # [G, m * nl + 1]
        mm, vv = mean.clone(), var.clone()
        plan = torch.zeros_like(mean)
        eng._check(eng.lib.cadm_dev_refit_sharded(eng._ctx, ptr(payload), G, nl, m, rank, ptr(mm), ptr(vv), 7, 3, 0, ptr(plan), eng.stream))
        return _np(mm), _np(vv), _np(plan)

    good = refit([w_same] * G, 0)
    assert all(np.isfinite(x).all() for x in good)
    # the same refit through the product API on the contiguous [G, m, nl] layout (no checksum word): bit-identical
    m0, v0 = mean.clone(), var.clone()
    eng.cem_refit(cand.view(G, m, nl).contiguous(), torch.empty((m, n, 6, 6), device=eng.device), m0, v0, G=G, regen=(7, 3, 0))
    np.testing.assert_array_equal(good[0], _np(m0))
    np.testing.assert_array_equal(good[1], _np(v0))
    for rank in range(G):                     # every rank sees the same verdict, whichever rank was fed something else
        for bad_word in (w_bit, w_perm):
            words = [w_same] * G
            words[2] = bad_word
            mm, vv, plan = refit(words, rank)
            assert np.isnan(mm).all() and np.isnan(plan).all(), "rank %d did not notice rank 2's different inputs" % rank
    eng.close()


def test_production_shape_m10(gpu):
    """The reference's launch shape: m = 10 vectorised envs (/root/reference/run_scripts/run_cadm_pets.py:201,
    cadm/samplers/sampler.py:109-113) x cand 200 x part 20 x E 5, H 30.  Quirk Q2 (core/utils.py:433-439) only bites at
    m > 1 on ODD CEM iterations: sampled (env, candidate) rows against the oracle on it = 1 and it = 2, candidate shards
    compose bit for bit, and the fused planner is deterministic and equals the per-iteration composition."""
    E, p, n, H, m = 5, 20, 200, 30, 10
    prob = synth.make_problem(env="halfcheetah", context=True, E=E, m=m, H=H, trained_like=True, seed=83)
    eng = make_engine(prob, p=p)
    ctx = eng.context_forward(prob["cp_obs"], prob["cp_act"])
    mean, var = eng._t(prob["init_mean"]), eng._t(prob["init_var"])
    rng = np.random.default_rng(7)
    sub = np.sort(rng.choice(n, size=6, replace=False))
    for it in (1, 2):
        actions = eng.sample_actions(mean, var, n, seed=3, call=1, it=it)                 # [10, 200, 30, 6]
        eps = torch.randn((H, m, n, p, prob["D"]), device=eng.device)
        rows = eng.rollout_returns(prob["obs"], ctx, actions, eps=eps, it=it)             # [10, 200, 20]
        a_sub, e_sub = _np(actions)[:, sub], _np(eps)[:, :, sub]
        res = {}
        for dt in (np.float32, np.float64):
            o = oracle_problem(prob, dt)
            T = oplanner.context_table_indexed(onets.context_forward(o["cp"], o["cp_obs"], o["cp_act"], o["st"]), it)
            res[dt] = oplanner.rollout_indexed(o["env"], o["ff"], o["st"], o["obs"], T, a_sub.astype(dt), e_sub.astype(dt), E, p, False)
        band = max(8 * rel_err(res[np.float32], res[np.float64]), 2e-5)
        assert rel_err(_np(rows)[:, sub], res[np.float32]) <= band, "it=%d: %.2e (band %.2e)" % (it, rel_err(_np(rows)[:, sub], res[np.float32]), band)
        h = n // 2
        parts = [eng.rollout_returns(prob["obs"], ctx, actions, eps=eps[:, :, g * h:(g + 1) * h].contiguous(), it=it,
                                     cand_offset=g * h, n_local=h) for g in range(2)]
        np.testing.assert_array_equal(_np(torch.cat(parts, dim=1)), _np(rows))
    # Q2 must actually matter at this shape: the odd-iteration layout differs from the even one
    o = oracle_problem(prob, np.float32)
    octx = onets.context_forward(o["cp"], o["cp_obs"], o["cp_act"], o["st"])
    assert not np.array_equal(oplanner.context_table_indexed(octx, 0), oplanner.context_table_indexed(octx, 1))
    args = (prob["obs"], prob["cp_obs"], prob["cp_act"], prob["init_mean"], prob["init_var"], n)
    a = eng.cem_plan(*args, seed=11, call=4)
    b = eng.cem_plan(*args, seed=11, call=4)
    c = hplanner.cem_plan(eng, *args, seed=11, call=4)
    np.testing.assert_array_equal(_np(a), _np(b))
    np.testing.assert_array_equal(_np(a), _np(c))
    assert _np(a).shape == (m, H, 6) and np.abs(_np(a)).max() <= 1.0
