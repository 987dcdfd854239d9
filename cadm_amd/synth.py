"""Seeded synthetic workloads for the planner / trainer (no simulator, no dataset).

Shapes and distributions follow SURVEY.md §8d / BASELINE.md §3: weights use the
reference initialiser trunc_normal(sigma = 1/(2 sqrt(in))), bias 0
(/root/reference/cadm/dynamics/core/utils.py:636-641), max/min_logvar 0.5 / -10
(:338-339); norm stats mu ~ N(0,1), sigma ~ U(0.5,2); obs ~ N(0,1);
cp_obs ~ 0.1 N(0,1); cp_act ~ U(-1,1); CEM init mean 0, var 0.25
(/root/reference/cadm/samplers/sampler.py:52-53).
numpy only (make_engine imports the HIP engine lazily) -- used by bench.py, __graft_entry__.smoke(), tools/ and tests.
"""
from collections import OrderedDict

import numpy as np

ENV_SHAPES = {  # name: (obs_dim D, act_dim A, proc_obs_dim P, discrete)
    "halfcheetah": (18, 6, 18, False),
    "cripple_halfcheetah": (18, 6, 18, False),
    "ant": (28, 8, 27, False),
    "slim_humanoid": (45, 17, 45, False),
    "cartpole": (4, 2, 4, True),
    "pendulum": (3, 1, 3, False),
}

# BASELINE.json configs (m = 1, H = 30)
CONFIGS = {
    "cfg1": dict(env="halfcheetah", context=False, E=1, p=1, n=200, H=30, deterministic=True),
    "cfg2": dict(env="halfcheetah", context=True, E=5, p=20, n=200, H=30, deterministic=False),
    "cfg3": dict(env="halfcheetah", context=True, E=5, p=20, n=2000, H=30, deterministic=False),
    "cfg4": dict(env="slim_humanoid", context=True, E=5, p=20, n=1000, H=30, deterministic=False),
    "cfg5": dict(env="halfcheetah", context=True, E=5, p=20, n=8000, H=30, deterministic=False),
}


def _trunc_normal(rng, shape, std):
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return out * std


def _dense(rng, E, din, dout, gain=1.0):
    W = _trunc_normal(rng, (E, din, dout), gain / (2.0 * np.sqrt(din)))
    return W, np.zeros((E, 1, dout))


def dynamics_params(rng, E, K0, hidden_sizes, D, gain=1.0, bias_scale=0.0):
    p = OrderedDict()
    sizes = [K0] + list(hidden_sizes)
    for i in range(len(sizes) - 1):
        W, b = _dense(rng, E, sizes[i], sizes[i + 1], gain)
        p["hidden_%d_weight" % i] = W
        p["hidden_%d_bias" % i] = b + bias_scale * rng.standard_normal(b.shape)
    for head in ("output_mu", "output_logvar"):
        W, b = _dense(rng, E, sizes[-1], D, gain)
        p[head + "_weight"] = W
        p[head + "_bias"] = b + bias_scale * rng.standard_normal(b.shape)
    p["max_logvar"] = np.ones((1, D)) / 2.0
    p["min_logvar"] = -np.ones((1, D)) * 10.0
    return p


def context_params(rng, E, cp_in, cp_hidden_sizes, C, gain=1.0, bias_scale=0.0):
    p = OrderedDict()
    sizes = [cp_in] + list(cp_hidden_sizes)
    for i in range(len(sizes) - 1):
        W, b = _dense(rng, E, sizes[i], sizes[i + 1], gain)
        p["cp_hidden_%d_weight" % i] = W
        p["cp_hidden_%d_bias" % i] = b + bias_scale * rng.standard_normal(b.shape)
    W, b = _dense(rng, E, sizes[-1], C, gain)
    p["cp_output_weight"] = W
    p["cp_output_bias"] = b + bias_scale * rng.standard_normal(b.shape)
    return p


def norm_stats(rng, D, A, P, Hh, state_diff=True, discrete=False):
    s = OrderedDict()

    def ms(k):
        return rng.standard_normal(k), rng.uniform(0.5, 2.0, k)

    s["obs_mean"], s["obs_std"] = ms(P)
    s["act_mean"], s["act_std"] = (np.zeros(A), np.ones(A)) if discrete else ms(A)
    s["delta_mean"], s["delta_std"] = ms(D)
    s["cp_obs_mean"], s["cp_obs_std"] = (np.zeros(D * Hh), np.ones(D * Hh)) if state_diff else ms(D * Hh)
    s["cp_act_mean"], s["cp_act_std"] = (np.zeros(A * Hh), np.ones(A * Hh)) if discrete else ms(A * Hh)
    s["back_delta_mean"], s["back_delta_std"] = ms(D)
    return s


def make_problem(env="halfcheetah", context=True, E=5, m=1, hidden_sizes=(200,) * 4,
                 cp_hidden_sizes=(256, 128, 64), C=10, Hh=10, H=30, seed=0, trained_like=False,
                 with_back=False, **_unused):
    """Weights + stats + planner inputs for one synthetic planning problem.
    ``trained_like`` rescales the heads so the per-step delta is of unit order and
    gives the biases small random values (the reference init leaves them at 0)."""
    D, A, P, discrete = ENV_SHAPES[env]
    rng = np.random.default_rng(seed)
    Cc = C if context else 0
    K0 = P + A + Cc
    gain, bs = (2.0, 0.1) if trained_like else (1.0, 0.0)
    prob = dict(env=env, D=D, A=A, P=P, C=Cc, E=E, m=m, H=H, Hh=Hh, discrete=discrete, K0=K0,
                hidden_sizes=tuple(hidden_sizes), cp_hidden_sizes=tuple(cp_hidden_sizes))
    prob["ff"] = dynamics_params(rng, E, K0, hidden_sizes, D, gain, bs)
    prob["cp"] = context_params(rng, E, (D + A) * Hh, cp_hidden_sizes, C, gain, bs) if context else None
    prob["back"] = dynamics_params(rng, E, K0, hidden_sizes, D, gain, bs) if with_back else None
    prob["stats"] = norm_stats(rng, D, A, P, Hh, state_diff=True, discrete=discrete)
    prob["obs"] = rng.standard_normal((m, D))
    prob["cp_obs"] = 0.1 * rng.standard_normal((m, D * Hh))
    prob["cp_act"] = rng.uniform(-1.0, 1.0, (m, A * Hh))
    prob["init_mean"] = np.zeros((m, H, A))
    prob["init_var"] = np.full((m, H, A), 0.25)
    return prob


def make_train_batch(prob, B=256, seed=1):
    """Synthetic [E,B,.] bootstrap batch for the training step."""
    rng = np.random.default_rng(seed)
    E, D, A, Hh = prob["E"], prob["D"], prob["A"], prob["Hh"]
    b = OrderedDict()
    b["obs"] = rng.standard_normal((E, B, D))
    b["act"] = rng.uniform(-1, 1, (E, B, A))
    b["obs_next"] = b["obs"] + 0.1 * rng.standard_normal((E, B, D))
    b["delta"] = rng.standard_normal((E, B, D))
    b["back_delta"] = rng.standard_normal((E, B, D))
    b["cp_obs"] = 0.1 * rng.standard_normal((E, B, D * Hh))
    b["cp_act"] = rng.uniform(-1, 1, (E, B, A * Hh))
    return b


def make_engine(prob, p, H=None, deterministic=False, quirks=True, device=None, **kw):
    """HipEngine loaded with a synthetic problem's weights and statistics (product code: no oracle involved)."""
    from .engine import HipEngine
    eng = HipEngine(prob["env"], prob["E"], p, prob["D"], prob["A"], prob["P"], prob["C"], prob["hidden_sizes"],
                    prob["H"] if H is None else H, deterministic=deterministic, discrete=prob["discrete"],
                    reference_quirks=quirks, history_length=prob["Hh"], cp_hidden_sizes=prob["cp_hidden_sizes"],
                    back_model=prob.get("back") is not None, device=device, **kw)
    if prob["cp"] is not None:
        eng.set_net("context_model", prob["cp"])
    eng.set_net("ff_model", prob["ff"])
    if prob.get("back") is not None:
        eng.set_net("backward_model", prob["back"])
    eng.set_stats(prob["stats"])
    return eng
