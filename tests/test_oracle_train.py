"""CPU: pins the oracle's training restatement: TF1 Adam closed form, which variables get a
gradient, and fit()'s host-side data path."""
import numpy as np
import pytest
import torch

from cadm_amd import synth
from oracle import envs as oenvs
from oracle import train as ot

WD = (0.000025, 0.00005, 0.000075, 0.000075, 0.0001)
CWD = (0.000025, 0.00005, 0.000075)


def test_tf1_adam_closed_form():
    """One parameter, constant gradient g: after t steps TF1 Adam has moved by
    sum_k lr*sqrt(1-b2^k)/(1-b1^k) * m_k / (sqrt(v_k) + eps) with m_k = g(1-b1^k), v_k = g^2(1-b2^k)."""
    lr, b1, b2, eps, g = 1e-3, 0.9, 0.999, 1e-8, 0.37
    w = {"n": {"w": torch.zeros(1, dtype=torch.float64)}}
    opt = ot.TF1Adam(lr, b1, b2, eps)
    expect = 0.0
    for k in range(1, 6):
        opt.step(w, {"n": {"w": torch.full((1,), g, dtype=torch.float64)}})
        lr_t = lr * np.sqrt(1 - b2 ** k) / (1 - b1 ** k)
        expect -= lr_t * (g * (1 - b1 ** k)) / (np.sqrt(g * g * (1 - b2 ** k)) + eps)
        np.testing.assert_allclose(float(w["n"]["w"]), expect, rtol=1e-12)
    # eps sits OUTSIDE the bias-corrected sqrt: differs from torch.optim.Adam for tiny gradients
    w2 = torch.zeros(1, dtype=torch.float64, requires_grad=True)
    topt = torch.optim.Adam([w2], lr=lr, betas=(b1, b2), eps=eps)
    w2.grad = torch.full((1,), 1e-9, dtype=torch.float64)
    topt.step()
    w3 = {"n": {"w": torch.zeros(1, dtype=torch.float64)}}
    ot.TF1Adam(lr, b1, b2, eps).step(w3, {"n": {"w": torch.full((1,), 1e-9, dtype=torch.float64)}})
    assert abs(float(w3["n"]["w"]) - float(w2.detach())) > 1e-5


def _setup(det, with_back, context=True):
    prob = synth.make_problem(env="halfcheetah", context=context, E=2, trained_like=True, with_back=with_back, seed=1)
    dt = torch.float64
    ff = ot.to_torch(prob["ff"], dt, True)
    back = ot.to_torch(prob["back"], dt, True) if with_back else None
    cp = ot.to_torch(prob["cp"], dt, True) if context else None
    st = ot.to_torch(prob["stats"], dt)
    batch = {k: torch.tensor(v, dtype=dt) for k, v in synth.make_train_batch(prob, B=8).items()}
    cfg = dict(deterministic=det, back_coeff=0.5 if with_back else 0.0, weight_decay_coeff=1.0, weight_decays=WD,
               context_weight_decays=CWD, n_hidden=4, n_cp_hidden=3)
    out = ot.train_losses("halfcheetah", ff, back, cp, st, batch, cfg)
    return out, ot.grads_of(out["loss"], {"ff_model": ff, "backward_model": back, "context_model": cp}), (ff, back, cp)


def test_which_variables_have_gradients():
    """the backward net's logvar bias / bounds get no gradient; its logvar weight only L2."""
    out, g, (ff, back, cp) = _setup(det=False, with_back=True)
    assert all(v is not None for v in g["ff_model"].values())
    assert all(v is not None for v in g["context_model"].values())
    gb = g["backward_model"]
    assert gb["output_logvar_bias"] is None and gb["max_logvar"] is None and gb["min_logvar"] is None
    np.testing.assert_allclose(gb["output_logvar_weight"].numpy(), 1e-4 * back["output_logvar_weight"].detach().numpy(), rtol=1e-12)
    out, g, _ = _setup(det=True, with_back=False, context=False)
    gf = g["ff_model"]
    assert gf["output_logvar_bias"] is None and gf["max_logvar"] is None
    assert float(out["recon"]) == float(out["mse"])


def test_recon_excludes_reg_and_l2():
    out, _, (ff, back, cp) = _setup(det=False, with_back=True)
    reg = 0.01 * ff["max_logvar"].sum() - 0.01 * ff["min_logvar"].sum()
    assert float(out["loss"] - out["recon"] - reg) > 0          # the remainder is the positive L2 term


def test_preprocess_inputs_and_stats():
    rng = np.random.default_rng(0)
    N, F, D, A, Hh = 7, 3, 18, 6, 2
    obs = rng.standard_normal((N, F * D)); act = rng.standard_normal((N, F * A)); nxt = rng.standard_normal((N, F * D))
    cpo = rng.standard_normal((N, D * Hh)); cpa = rng.standard_normal((N, A * Hh))
    fb = (rng.uniform(size=(N, F)) > 0.3).astype(float)
    out = ot.preprocess_inputs(obs, act, obs, cpo, cpa, fb, nxt, nxt, D, A, Hh, F)
    n = int(fb.sum())
    assert all(o.shape[0] == n for o in out)
    # row (i, f) keeps sample i's history and its f-th future step
    i, f = np.argwhere(fb > 0)[3]
    np.testing.assert_array_equal(out[0][3], obs[i, f * D:(f + 1) * D])
    np.testing.assert_array_equal(out[5][3], cpo[i])
    env = oenvs.make_env("halfcheetah")
    norm = ot.compute_normalization(env, obs[:, :D], act[:, :A], obs[:, :D], cpo, cpa, nxt[:, :D])
    s = ot.normalization_stats(norm, D, A, Hh, discrete=False, state_diff=True)
    assert np.all(s["cp_obs_mean"] == 0) and np.all(s["cp_obs_std"] == 1)      # state_diff forces (0,1)
    np.testing.assert_allclose(s["obs_std"], env.obs_preproc(obs[:, :D]).std(0))   # population std (ddof = 0)


@pytest.mark.parametrize("env_name,state_diff,normalize_input", [("halfcheetah", True, True), ("halfcheetah", False, True),
                                                                 ("cartpole", True, True), ("halfcheetah", True, False)])
def test_model_normalization_stats_match_the_oracle(env_name, state_diff, normalize_input):
    """`compute_normalization` / `get_normalization_stats` of the drop-in class against the oracle restatement of
    /root/reference/cadm/dynamics/mlp_cadm_ensemble_cem_dynamics.py:590-645 -- population statistics in float64, cp_obs
    forced to (0, 1) under state_diff (:616-618), action statistics forced to (0, 1) for discrete envs (:610-612,622-624),
    everything (0, 1) without normalize_input (:627-644).  Host-only: the methods never touch the engine."""
    from cadm_amd.dynamics.mlp_cadm_ensemble_cem_dynamics import MLPEnsembleCEMDynamicsModel
    from cadm_amd.envs import make_env_spec
    from oracle import envs as oenvs
    spec = make_env_spec(env_name)
    env = oenvs.make_env(env_name)
    D, A, Hh = env.obs_dim, env.act_dim, 4
    rng = np.random.default_rng(12)
    N = 200
    obs, delta, back = rng.standard_normal((N, D)) * 3 + 1, rng.standard_normal((N, D)), rng.standard_normal((N, D))
    act = rng.uniform(-1, 1, (N, A))
    cp_obs, cp_act = rng.standard_normal((N, D * Hh)) * 0.3, rng.uniform(-1, 1, (N, A * Hh))
    m = object.__new__(MLPEnsembleCEMDynamicsModel)              # host-side methods only: no engine, no GPU
    m.env, m.normalize_input, m.state_diff = spec, normalize_input, state_diff
    m.discrete = env.discrete
    m.obs_space_dims, m.action_space_dims, m.proc_obs_space_dims, m.history_length = D, A, env.proc_obs_dim, Hh
    m.normalization = None
    m.compute_normalization(obs, act, delta, cp_obs, cp_act, back)
    got = m.get_normalization_stats()
    keys = ("obs_mean", "obs_std", "act_mean", "act_std", "delta_mean", "delta_std", "cp_obs_mean", "cp_obs_std",
            "cp_act_mean", "cp_act_std", "back_delta_mean", "back_delta_std")
    if normalize_input:
        want = ot.normalization_stats(ot.compute_normalization(env, obs, act, delta, cp_obs, cp_act, back), D, A, Hh,
                                          env.discrete, state_diff)
    else:
        P = env.proc_obs_dim
        want = dict(obs_mean=np.zeros(P), obs_std=np.ones(P), act_mean=np.zeros(A), act_std=np.ones(A), delta_mean=np.zeros(D),
                    delta_std=np.ones(D), cp_obs_mean=np.zeros(D * Hh), cp_obs_std=np.ones(D * Hh), cp_act_mean=np.zeros(A * Hh),
                    cp_act_std=np.ones(A * Hh), back_delta_mean=np.zeros(D), back_delta_std=np.ones(D))
    for k, v in zip(keys, got):
        np.testing.assert_array_equal(np.asarray(v, np.float64), np.asarray(want[k], np.float64), err_msg=k)


def test_vanilla_model_reads_a_reference_layout_three_key_norm_stats_file(tmp_path):
    """the reference's vanilla model saves only obs / delta / act statistics
    (/root/reference/cadm/dynamics/mlp_ensemble_cem_dynamics.py:343-351); its stats accessor returns the 6-tuple (:353-373)."""
    import joblib
    from collections import OrderedDict
    from cadm_amd.dynamics.mlp_ensemble_cem_dynamics import MLPEnsembleCEMDynamicsModel as Vanilla
    from cadm_amd.envs import make_env_spec
    rng = np.random.default_rng(1)
    norm = OrderedDict(obs=(rng.standard_normal(18), rng.uniform(0.5, 2, 18)), delta=(rng.standard_normal(18), rng.uniform(0.5, 2, 18)),
                       act=(rng.standard_normal(6), rng.uniform(0.5, 2, 6)))
    joblib.dump(norm, str(tmp_path / "params_norm_stats"))
    m = object.__new__(Vanilla)
    m.env, m.normalize_input, m.state_diff, m.discrete = make_env_spec("halfcheetah"), True, False, False
    m.obs_space_dims, m.action_space_dims, m.proc_obs_space_dims, m.history_length = 18, 6, 18, 0
    m.normalization = joblib.load(str(tmp_path / "params_norm_stats"))
    six = m.get_normalization_stats()
    assert len(six) == 6
    np.testing.assert_array_equal(six[0], norm["obs"][0]); np.testing.assert_array_equal(six[3], norm["act"][1])
    twelve = m._stats12()                     # what the engine consumes: the missing three default to (0, 1) of width 0 / D
    assert len(twelve) == 12 and twelve[6].shape == (0,) and twelve[10].shape == (18,) and float(twelve[11].min()) == 1.0
    m.compute_normalization(rng.standard_normal((50, 18)), rng.standard_normal((50, 6)), rng.standard_normal((50, 18)))
    assert list(m.normalization) == ["obs", "delta", "act"]
