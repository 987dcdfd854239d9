"""CPU: pins the oracle's planner restatement (no reference tests exist -- SURVEY.md 8c):
literal tile/transpose/reshape transcription vs the index-mapped form, fp32 vs fp64,
the verified index facts of SURVEY.md Appendix A.4, and analytic known-answer cases."""
import numpy as np
import pytest

from cadm_amd import synth
from helpers import oracle_problem, trunc_z
from oracle import envs as oenvs
from oracle import nets as onets
from oracle import planner as op


@pytest.mark.parametrize("env,m", [("halfcheetah", 1), ("halfcheetah", 3), ("slim_humanoid", 2), ("ant", 2),
                                   ("pendulum", 2), ("cartpole", 2)])
def test_literal_equals_indexed_cem(env, m):
    E, p, n, H = 5, 10, 50, 4
    prob = synth.make_problem(env=env, E=E, m=m, H=H, trained_like=True, seed=1)
    o = oracle_problem(prob, np.float64)
    rng = np.random.default_rng(5)
    z = trunc_z(rng, (5, m, n, H, prob["A"]))
    eps = rng.standard_normal((5, H, m, n, p, prob["D"]))
    args = (o["env"], o["ff"], o["cp"], o["st"], o["obs"], o["cp_obs"], o["cp_act"], o["init_mean"], o["init_var"], z, eps, E, p)
    a = op.cem_plan(*args, formulation="literal")
    b = op.cem_plan(*args, formulation="indexed")
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12)


def test_index_facts_appendix_a4():
    """member -> particles, particle -> context encoder (Q1), and the per-iteration context scramble (Q2)."""
    E, p = 5, 20
    assert [op.member_of_particle(j, p, E) for j in range(p)] == [j // 4 for j in range(p)]
    m, C = 3, 2
    ctx = np.arange(E * m * C, dtype=np.float64).reshape(E, m, C)
    lit = op.context_table_literal(ctx, 5)
    for it in range(5):
        np.testing.assert_array_equal(lit[it], op.context_table_indexed(ctx, it))
    np.testing.assert_array_equal(lit[0], lit[2])
    np.testing.assert_array_equal(lit[0], lit[4])
    assert not np.array_equal(lit[0], lit[1])            # m > 1: odd iterations scramble contexts across envs
    np.testing.assert_array_equal(lit[1], lit[3])
    ctx1 = ctx[:, :1]                                      # m = 1: Q2 is a no-op
    l1 = op.context_table_literal(ctx1, 2)
    np.testing.assert_array_equal(l1[0], l1[1])
    # Q1: particle j of member j // 4 reads encoder j % 5 -> member 1 consumes encoders {4,0,1,2}
    assert [j % E for j in range(4, 8)] == [4, 0, 1, 2]


def test_rs_literal_equals_indexed_and_first_max():
    E, p, m, n, H = 5, 5, 2, 20, 3
    prob = synth.make_problem(env="halfcheetah", E=E, m=m, H=H, trained_like=True, seed=2)
    o = oracle_problem(prob, np.float64)
    rng = np.random.default_rng(3)
    acts = rng.uniform(-1, 1, (m, n, H, 6))
    eps = rng.standard_normal((H, m, n, p, 18))
    a, ca = op.rs_plan(o["env"], o["ff"], o["cp"], o["st"], o["obs"], o["cp_obs"], o["cp_act"], acts, eps, E, p, formulation="literal")
    b, cb = op.rs_plan(o["env"], o["ff"], o["cp"], o["st"], o["obs"], o["cp_obs"], o["cp_act"], acts, eps, E, p, formulation="indexed")
    np.testing.assert_allclose(ca, cb, rtol=1e-12)
    np.testing.assert_array_equal(a, b)


def test_zero_weights_known_answer():
    """All weights zero: mu = b = 0 -> delta = delta_mean (+ eps * exp((clamp(0) + 2 log sd) / 2))."""
    prob = synth.make_problem(env="slim_humanoid", E=5, m=1, H=1, seed=0)
    o = oracle_problem(prob, np.float64)
    ff = {k: (np.zeros_like(v) if ("weight" in k or "bias" in k) else v) for k, v in o["ff"].items()}
    D = 45
    x = np.zeros((5, 3, prob["K0"]))
    eps = np.ones((5, 3, D))
    delta, mu, lv = onets.dynamics_forward(ff, x, o["st"]["delta_mean"], o["st"]["delta_std"], eps, False)
    assert np.all(mu == 0)
    lv_expected = -10.0 + np.log1p(np.exp((0.5 - np.log1p(np.exp(0.5))) + 10.0))
    np.testing.assert_allclose(lv, lv_expected, rtol=1e-12)
    np.testing.assert_allclose(delta, np.broadcast_to(o["st"]["delta_mean"] + np.exp((lv_expected + 2 * np.log(o["st"]["delta_std"])) / 2), delta.shape), rtol=1e-12)
    d_det, _, _ = onets.dynamics_forward(ff, x, o["st"]["delta_mean"], o["st"]["delta_std"], eps, True)
    np.testing.assert_allclose(d_det, np.broadcast_to(o["st"]["delta_mean"], d_det.shape), rtol=1e-12)


def test_deterministic_single_member_ignores_eps():
    prob = synth.make_problem(env="halfcheetah", context=False, E=1, m=1, H=5, seed=0)
    o = oracle_problem(prob, np.float64)
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, (1, 7, 5, 6))
    r1 = op.rollout_indexed(o["env"], o["ff"], o["st"], o["obs"], None, acts, rng.standard_normal((5, 1, 7, 1, 18)), 1, 1, True)
    r2 = op.rollout_indexed(o["env"], o["ff"], o["st"], o["obs"], None, acts, rng.standard_normal((5, 1, 7, 1, 18)), 1, 1, True)
    np.testing.assert_array_equal(r1, r2)


def test_top_k_ties_lower_index_first_and_refit():
    ret = np.array([[1.0, 3.0, 3.0, 2.0, 3.0, 0.0]])
    np.testing.assert_array_equal(op.top_k_indices(ret, 4), [[1, 2, 4, 3]])
    mean = np.zeros((1, 2, 1)); var = np.ones((1, 2, 1))
    actions = np.arange(12, dtype=np.float64).reshape(1, 6, 2, 1)
    nm, nv, idx = op.elite_refit(mean, var, actions, ret, num_elites=2, alpha=0.1)
    el = actions[0, [1, 2]]
    np.testing.assert_allclose(nm[0], 0.9 * el.mean(0))
    np.testing.assert_allclose(nv[0], 0.1 + 0.9 * el.var(0))       # biased variance


def test_softplus_matches_tf_thresholds():
    x = np.array([-30.0, -13.95, -13.9, 0.0, 13.9, 13.95, 30.0], np.float32)
    y = onets.tf_softplus(x)
    assert y[0] == np.exp(np.float32(-30.0)) and y[-1] == np.float32(30.0)
    np.testing.assert_allclose(y[3], np.log(2.0), rtol=1e-6)
    np.testing.assert_allclose(y, np.log1p(np.exp(x.astype(np.float64))), rtol=2e-6)


def test_constrained_var_and_warm_start():
    mean = np.array([[[0.9], [-0.5]]]); var = np.array([[[0.25], [0.25]]])
    cv = op.constrained_var(mean, var)
    np.testing.assert_allclose(cv[0, :, 0], [min(0.05 ** 2, 0.25), min(0.25 ** 2, 0.25)])
    prev = np.ones((2, 3, 1)); sol = np.arange(6, dtype=float).reshape(2, 3, 1)
    nxt, act = op.warm_start_shift(prev, sol)
    np.testing.assert_array_equal(nxt[:, :, 0], [[1, 2, 0], [4, 5, 0]])
    np.testing.assert_array_equal(act[:, 0], [0, 3])


@pytest.mark.parametrize("name", sorted(oenvs.ENVS))
def test_env_closures_roundtrip(name):
    env = oenvs.make_env(name)
    rng = np.random.default_rng(0)
    o, o2 = rng.standard_normal((4, env.obs_dim)), rng.standard_normal((4, env.obs_dim))
    np.testing.assert_allclose(env.obs_postproc(o, env.targ_proc(o, o2)), o2, rtol=1e-12)   # postproc inverts targ_proc
    assert env.obs_preproc(o).shape[-1] == env.proc_obs_dim
    a = rng.uniform(-1, 1, (4, env.act_dim))
    assert env.reward(o, a, o2).shape == (4,)
    # product-side closures agree with the oracle's
    from cadm_amd.envs import make_env_spec
    spec = make_env_spec(name)
    np.testing.assert_array_equal(spec.obs_preproc(o), env.obs_preproc(o))
    np.testing.assert_array_equal(spec.targ_proc(o, o2), env.targ_proc(o, o2))
    np.testing.assert_array_equal(spec.obs_postproc(o, o2), env.obs_postproc(o, o2))
    assert spec.observation_space.shape[0] == env.obs_dim and spec.proc_observation_space_dims == env.proc_obs_dim


def test_fp32_oracle_tracks_fp64():
    E, p, n, H, m = 5, 5, 20, 30, 1
    prob = synth.make_problem(env="halfcheetah", E=E, m=m, H=H, seed=3)
    rng = np.random.default_rng(1)
    acts = rng.uniform(-1, 1, (m, n, H, 6)); eps = rng.standard_normal((H, m, n, p, 18))
    res = {}
    for dt in (np.float32, np.float64):
        o = oracle_problem(prob, dt)
        T = op.context_table_indexed(onets.context_forward(o["cp"], o["cp_obs"], o["cp_act"], o["st"]), 0)
        res[dt] = op.rollout_indexed(o["env"], o["ff"], o["st"], o["obs"], T, acts.astype(dt), eps.astype(dt), E, p, False)
        assert res[dt].dtype == dt
    assert np.abs(res[np.float32] - res[np.float64]).max() / np.abs(res[np.float64]).max() < 1e-4
