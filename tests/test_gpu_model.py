"""GPU: the drop-in classes end to end -- the reference's constructor kwargs (run_cadm_pets.py:30-56 /
run_pets.py), get_action / get_context_pred / fit / save / load, MPCController dispatch and the CEM
warm start of the samplers."""
import os

import numpy as np
import pytest

from cadm_amd import synth
from cadm_amd.dynamics.mlp_cadm_ensemble_cem_dynamics import MLPEnsembleCEMDynamicsModel as CaDMModel
from cadm_amd.dynamics.mlp_ensemble_cem_dynamics import MLPEnsembleCEMDynamicsModel as VanillaModel
from cadm_amd.envs import make_env_spec
from cadm_amd.policies.mpc_controller import CEMWarmStart, MPCController

pytestmark = pytest.mark.gpu


def _cadm_kwargs(**over):
    kw = dict(name="dyn_model", env=make_env_spec("halfcheetah"), learning_rate=0.001, hidden_sizes=(200,) * 4,
              valid_split_ratio=0.1, rolling_average_persitency=0.99, hidden_nonlinearity="swish", batch_size=64,
              normalize_input=True, n_forwards=8, n_candidates=64, ensemble_size=5, n_particles=10, use_cem=True,
              deterministic=False, weight_decays=(0.000025, 0.00005, 0.000075, 0.000075, 0.0001), weight_decay_coeff=1.0,
              cp_hidden_sizes=(256, 128, 64), context_weight_decays=(0.000025, 0.00005, 0.000075), context_out_dim=10,
              context_hidden_nonlinearity="relu", history_length=10, future_length=10, state_diff=1, back_coeff=0.5)
    kw.update(over)
    return kw


def _synthetic_transitions(rng, N, D=18, A=6, Hh=10, F=10):
    """Linear-ish synthetic dynamics so that fit() has something to learn."""
    obs = rng.standard_normal((N, F * D))
    act = rng.uniform(-1, 1, (N, F * A))
    Wd = 0.1 * rng.standard_normal((D + A, D))
    o3, a3 = obs.reshape(N, F, D), act.reshape(N, F, A)
    nxt = o3 + np.concatenate([o3, a3], -1) @ Wd + 0.01 * rng.standard_normal((N, F, D))
    cp_obs = 0.1 * rng.standard_normal((N, D * Hh))
    cp_act = rng.uniform(-1, 1, (N, A * Hh))
    fb = np.ones((N, F))
    fb[:, 7:] = rng.integers(0, 2, (N, 3))
    return obs, act, nxt.reshape(N, F * D), cp_obs, cp_act, fb


def test_cadm_fit_plan_save_load(gpu, tmp_path):
    rng = np.random.default_rng(0)
    model = CaDMModel(**_cadm_kwargs())
    obs, act, nxt, cpo, cpa, fb = _synthetic_transitions(rng, 60)
    with pytest.raises(RuntimeError):
        model.get_action(obs[:2, :18], cpo[:2], cpa[:2], np.zeros((2, 8, 6)), np.full((2, 8, 6), 0.25))   # no stats yet
    l0 = None
    model.fit(obs, act, nxt, cpo, cpa, fb, epochs=3, rng=np.random.default_rng(1))
    assert model.normalization is not None and model._dataset["obs"].shape[0] == 60
    # losses go down over further epochs on the same data
    eng = model.engine
    ds = model._dataset
    tr = model._preprocess_inputs(ds["obs"], ds["act"], ds["delta"], ds["cp_obs"], ds["cp_act"], ds["future_bool"],
                                  ds["obs_next"], ds["back_delta"])
    names = ("obs", "act", "delta", "obs_next", "back_delta", "cp_obs", "cp_act")
    batch = {k: eng._t(v[:128])[None].expand(5, -1, -1).contiguous() for k, v in zip(names, tr)}
    l0 = eng.train_step(batch, train=False).cpu().numpy()
    model.fit(obs, act, nxt, cpo, cpa, fb, epochs=8, rng=np.random.default_rng(2))     # dataset doubles (reference :421-433)
    assert model._dataset["obs"].shape[0] == 120
    l1 = eng.train_step(batch, train=False).cpu().numpy()
    assert l1[0] < l0[0], "mse did not improve: %r -> %r" % (l0, l1)
    # planning
    m = 3
    pol = MPCController("policy", model.env, model, use_cem=True, n_candidates=64, horizon=8, context=True)
    ws = CEMWarmStart(m, 8, 6)
    o = rng.standard_normal((m, 18))
    sol, _ = pol.get_actions(o, cp_obs=cpo[:m], cp_act=cpa[:m], init_mean=ws.prev_sol, init_var=ws.init_var)
    assert sol.shape == (m, 8, 6) and np.abs(sol).max() <= 1.0 and sol.dtype == np.float32
    a0 = ws.step(sol)
    np.testing.assert_array_equal(a0, sol[:, 0])
    np.testing.assert_array_equal(ws.prev_sol[:, :-1], sol[:, 1:])
    assert np.all(ws.prev_sol[:, -1] == 0)
    cpred = model.get_context_pred(cpo[:m], cpa[:m])
    assert cpred.shape == (5, m, 10)
    # checkpoint: reference layout = flat list in tf.trainable_variables() order + _norm_stats
    path = str(tmp_path / "params_epoch_0")
    model.save(path)
    import joblib
    lst = joblib.load(path)
    assert len(lst) == 8 + 14 + 14                       # context 4x(W,b); ff & backward 6x(W,b)+2 bounds
    assert lst[0].shape == (5, 240, 256) and lst[8].shape == (5, 34, 200) and lst[-1].shape == (1, 18)
    assert os.path.exists(path + "_norm_stats")
    model2 = CaDMModel(**_cadm_kwargs(seed=123))
    model2.load(path)
    model._call = model2._call = 0
    model2.seed = model.seed
    a = model.get_action(o, cpo[:m], cpa[:m], np.zeros((m, 8, 6)), np.full((m, 8, 6), 0.25))
    b = model2.get_action(o, cpo[:m], cpa[:m], np.zeros((m, 8, 6)), np.full((m, 8, 6), 0.25))
    np.testing.assert_array_equal(a, b)


def test_cadm_random_shooting_and_errors(gpu):
    model = CaDMModel(**_cadm_kwargs(use_cem=False, normalize_input=False))
    rng = np.random.default_rng(3)
    o = rng.standard_normal((2, 18))
    a = model.get_action(o, 0.1 * rng.standard_normal((2, 180)), rng.uniform(-1, 1, (2, 60)))
    assert a.shape == (2, 6) and np.abs(a).max() <= 1.0
    # empty batch of environments (m = 0): empty result, like the reference's dynamic-m graph
    assert model.get_action(o[:0], np.zeros((0, 180)), np.zeros((0, 60))).shape == (0, 6)
    assert model.get_action(o[:0], np.zeros((0, 180)), np.zeros((0, 60)), np.zeros((0, 8, 6)), np.zeros((0, 8, 6))).shape == (0, 8, 6)
    with pytest.raises(NotImplementedError):       # a cross-feature nonlinearity / an output nonlinearity: not in the kernels
        CaDMModel(**_cadm_kwargs(hidden_nonlinearity="softmax"))
    with pytest.raises(NotImplementedError):
        CaDMModel(**_cadm_kwargs(output_nonlinearity="tanh"))
    with pytest.raises(ValueError):
        CaDMModel(**_cadm_kwargs(n_particles=7))
    class UnknownEnv:                      # right duck type, but no compiled-in closures for this class
        observation_space = make_env_spec("halfcheetah").observation_space
        action_space = make_env_spec("halfcheetah").action_space
        proc_observation_space_dims = 18
    with pytest.raises(ValueError, match="compiled-in env kind"):
        CaDMModel(**_cadm_kwargs(env=UnknownEnv()))


def test_vanilla_model_cfg1_shape(gpu):
    """configs[0]: Vanilla-DM CEM (ens=1, part=1, cand=200, H=30), deterministic."""
    model = VanillaModel("dyn_model", make_env_spec("halfcheetah"), hidden_nonlinearity="swish", batch_size=32,
                         n_forwards=30, n_candidates=200, ensemble_size=1, n_particles=1, use_cem=True, deterministic=True,
                         weight_decays=(0.000025, 0.00005, 0.000075, 0.000075, 0.0001), weight_decay_coeff=1.0,
                         normalize_input=True, valid_split_ratio=0.1)
    rng = np.random.default_rng(4)
    N = 300
    obs = rng.standard_normal((N, 18)); act = rng.uniform(-1, 1, (N, 6))
    nxt = obs + 0.1 * rng.standard_normal((N, 18))
    model.fit(obs, act, nxt, epochs=2, rng=np.random.default_rng(5))
    plan = model.get_action(obs[:1], np.zeros((1, 30, 6)), np.full((1, 30, 6), 0.25))
    assert plan.shape == (1, 30, 6) and np.isfinite(plan).all()
    pol = MPCController("policy", model.env, model, use_cem=True, context=False)
    sol, _ = pol.get_actions(obs[:2], init_mean=np.zeros((2, 30, 6)), init_var=np.full((2, 30, 6), 0.25))
    assert sol.shape == (2, 30, 6)


def test_discrete_cartpole_model(gpu):
    model = CaDMModel(**_cadm_kwargs(env=make_env_spec("cartpole"), use_cem=False, normalize_input=False, n_candidates=50))
    rng = np.random.default_rng(6)
    a = model.get_action(rng.standard_normal((3, 4)), np.zeros((3, 40)), np.zeros((3, 20)))
    assert a.shape == (3,) and set(np.unique(a)) <= {0, 1}


def test_predict_matches_oracle_one_step(gpu):
    """predict(): one-step mean next state of every member == oracle forward with eps = 0."""
    from helpers import assert_close, oracle_problem
    from oracle import nets as onets
    prob = synth.make_problem(env="halfcheetah", E=5, m=7, trained_like=True, seed=40)
    model = CaDMModel(**_cadm_kwargs(normalize_input=True))
    model.engine.set_net("context_model", prob["cp"])
    model.engine.set_net("ff_model", prob["ff"])
    st = prob["stats"]
    model.set_normalization({"obs": (st["obs_mean"], st["obs_std"]), "delta": (st["delta_mean"], st["delta_std"]),
                             "act": (st["act_mean"], st["act_std"]), "cp_obs": (st["cp_obs_mean"], st["cp_obs_std"]),
                             "cp_act": (st["cp_act_mean"], st["cp_act_std"]),
                             "back_delta": (st["back_delta_mean"], st["back_delta_std"])})
    rng = np.random.default_rng(1)
    act = rng.uniform(-1, 1, (7, 6))
    nxt, std = model.predict(prob["obs"], act, prob["cp_obs"], prob["cp_act"], return_std=True)
    assert nxt.shape == (5, 7, 18) and std.shape == (5, 7, 18) and (std > 0).all()
    o = oracle_problem(prob, np.float32)
    ctx = onets.context_forward(o["cp"], o["cp_obs"], o["cp_act"], o["st"])          # [E,m,C]
    nobs = onets.normalize(o["env"].obs_preproc(o["obs"]), o["st"]["obs_mean"], o["st"]["obs_std"])
    nact = onets.normalize(act.astype(np.float32), o["st"]["act_mean"], o["st"]["act_std"])
    x = np.concatenate([np.tile(nobs[None], (5, 1, 1)), np.tile(nact[None], (5, 1, 1)), ctx], -1)
    delta, mu, lv = onets.dynamics_forward(o["ff"], x, o["st"]["delta_mean"], o["st"]["delta_std"], np.zeros((5, 7, 18), np.float32), False)
    ref = o["env"].obs_postproc(np.broadcast_to(o["obs"][None], delta.shape), delta)
    assert_close(nxt, ref, 2e-5, "predict() mean next state")
    assert_close(std, np.exp((lv + 2 * np.log(o["st"]["delta_std"])) / 2), 2e-5, "predict() std")


def test_device_planner_state_matches_sampler_bookkeeping(gpu):
    """Device-resident warm start + history ring buffer == the reference samplers' numpy bookkeeping,
    over 14 steps (> history_length, so the shift path runs) with episode ends in between."""
    from cadm_amd.caller import DevicePlannerState
    from oracle.caller import SamplerState
    m, H = 3, 8
    model = CaDMModel(**_cadm_kwargs(normalize_input=False, n_candidates=64))
    dev = DevicePlannerState(model, m)
    ref = SamplerState(m, H, 18, 6, 10, state_diff=1)
    rng = np.random.default_rng(0)
    obs = rng.standard_normal((m, 18)).astype(np.float32)
    for step in range(14):
        np.testing.assert_array_equal(dev.hist_obs.cpu().numpy(), ref.history_state.astype(np.float32))
        np.testing.assert_array_equal(dev.hist_act.cpu().numpy(), ref.history_act.astype(np.float32))
        np.testing.assert_array_equal(dev.prev_sol.cpu().numpy(), ref.prev_sol.astype(np.float32))
        # same planner call through the class API (host state) and through the device state: same seed/call -> same plan
        call_before = model._call
        plan_ref = model.get_action(obs, ref.history_state, ref.history_act, ref.prev_sol, ref.init_var)
        model._call = call_before
        act_dev = dev.act(obs).cpu().numpy()
        act_ref = ref.after_plan(plan_ref)
        np.testing.assert_array_equal(act_dev, act_ref.astype(np.float32))
        nxt = (obs + 0.1 * rng.standard_normal((m, 18))).astype(np.float32)
        done = np.array([step == 5, False, step in (3, 11)])
        dev.observe(obs, act_dev, nxt, done)
        ref.after_step(obs, act_ref.astype(np.float32), nxt, done)
        obs = nxt
    np.testing.assert_array_equal(dev.counts.cpu().numpy(), np.array(ref.state_counts, dtype=np.int32))
