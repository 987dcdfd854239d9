"""Oracle restatement of the reference networks (numpy, dtype-generic).

Test infrastructure only (see oracle/__init__.py).

Follows /root/reference/cadm/dynamics/core/utils.py:
  create_dense_layer      :635-647   y = act(x @ W[e] + b[e]),  W [E,in,out], b [E,1,out]
  dynamics MLP + heads    :309-339   (vanilla twin :39-71)
  forward()               :341-370   (vanilla :73-92)
  context encoder         :569-624
  normalize/denormalize   :627-632
Third-party semantics restated: tf.nn.softplus (TF 1.15 Eigen functor
``softplus_op.h``: threshold = log(eps)+2; x > -threshold -> x;
x < threshold -> exp(x); else log1p(exp(x))).
"""
from collections import OrderedDict

import numpy as np

EPS_STD = 1e-10  # utils.py:628,632


def normalize(x, mean, std):  # utils.py:627-628
    dt = x.dtype.type
    return (x - mean) / (std + dt(EPS_STD))


def denormalize(x, mean, std):  # utils.py:631-632
    dt = x.dtype.type
    return x * (std + dt(EPS_STD)) + mean


def tf_softplus(x):
    """tf.nn.softplus as implemented by TF 1.15's Eigen functor."""
    dt = x.dtype.type
    threshold = dt(np.log(np.finfo(x.dtype).eps)) + dt(2.0)
    too_large = x > -threshold
    too_small = x < threshold
    with np.errstate(over="ignore"):
        x_exp = np.exp(x)
    out = np.log1p(x_exp)
    out = np.where(too_small, x_exp, out)
    out = np.where(too_large, x, out)
    return out.astype(x.dtype)


def swish(x):  # mlp_cadm_ensemble_cem_dynamics.py:23  x * sigmoid(x)
    dt = x.dtype.type
    with np.errstate(over="ignore"):
        return x * (dt(1) / (dt(1) + np.exp(-x)))


def relu(x):
    return np.maximum(x, x.dtype.type(0))


ACTIVATIONS = {  # mlp_cadm_ensemble_cem_dynamics.py:17-24
    None: lambda x: x,
    "relu": relu,
    "tanh": np.tanh,
    "sigmoid": lambda x: x.dtype.type(1) / (x.dtype.type(1) + np.exp(-x)),
    "swish": swish,
}


def trunc_normal(rng, shape, std):
    """tf.truncated_normal_initializer: N(0,std) re-drawn outside 2 std."""
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return out * std


def init_dense(rng, E, din, dout):  # utils.py:636-641
    W = trunc_normal(rng, (E, din, dout), 1.0 / (2.0 * np.sqrt(din)))
    b = np.zeros((E, 1, dout))
    return W, b


def init_dynamics_params(rng, E, K0, hidden_sizes, D):
    """Parameter list in tf.trainable_variables() creation order
    (utils.py:313-339): hidden_i_weight, hidden_i_bias, ..., output_mu_*,
    output_logvar_*, max_logvar, min_logvar."""
    p = OrderedDict()
    sizes = [K0] + list(hidden_sizes)
    for i in range(len(sizes) - 1):
        p["hidden_%d_weight" % i], p["hidden_%d_bias" % i] = init_dense(rng, E, sizes[i], sizes[i + 1])
    p["output_mu_weight"], p["output_mu_bias"] = init_dense(rng, E, sizes[-1], D)
    p["output_logvar_weight"], p["output_logvar_bias"] = init_dense(rng, E, sizes[-1], D)
    p["max_logvar"] = np.ones((1, D)) / 2.0       # utils.py:338
    p["min_logvar"] = -np.ones((1, D)) * 10.0     # utils.py:339
    return p


def init_context_params(rng, E, cp_in, cp_hidden_sizes, C):
    """utils.py:591-612: cp_hidden_i_*, cp_output_*."""
    p = OrderedDict()
    sizes = [cp_in] + list(cp_hidden_sizes)
    for i in range(len(sizes) - 1):
        p["cp_hidden_%d_weight" % i], p["cp_hidden_%d_bias" % i] = init_dense(rng, E, sizes[i], sizes[i + 1])
    p["cp_output_weight"], p["cp_output_bias"] = init_dense(rng, E, sizes[-1], C)
    return p


def cast_params(p, dtype):
    return OrderedDict((k, np.asarray(v, dtype=dtype)) for k, v in p.items())


def n_hidden(p):
    return sum(1 for k in p if k.startswith("hidden_") and k.endswith("_weight"))


def dense(x, W, b, act):  # utils.py:643-646
    return act(np.matmul(x, W) + b)


def mlp_hidden(p, x, hidden_act):
    for i in range(n_hidden(p)):
        x = dense(x, p["hidden_%d_weight" % i], p["hidden_%d_bias" % i], hidden_act)
    return x


def dynamics_forward(p, x, delta_mean, delta_std, eps, deterministic,
                     hidden_act=swish, out_act=ACTIVATIONS[None]):
    """utils.py:341-370.  x [E,R,K0] -> (delta [E,R,D], mu, logvar_clamped).
    ``eps`` replaces tf.random.normal (same shape as mu); ignored if deterministic.
    delta_mean/std are the (back-)delta stats used to denormalise."""
    dt = x.dtype.type
    h = mlp_hidden(p, x, hidden_act)
    mu = dense(h, p["output_mu_weight"], p["output_mu_bias"], out_act)
    logvar = dense(h, p["output_logvar_weight"], p["output_logvar_bias"], out_act)
    dmu = denormalize(mu, delta_mean, delta_std)
    if deterministic:
        return dmu, mu, logvar
    max_lv, min_lv = p["max_logvar"], p["min_logvar"]
    logvar = max_lv - tf_softplus(max_lv - logvar)          # :356
    logvar = min_lv + tf_softplus(logvar - min_lv)          # :357
    dlogvar = logvar + dt(2) * np.log(delta_std)            # :360
    dstd = np.exp(dlogvar / dt(2.0))                        # :363
    return dmu + eps * dstd, mu, logvar                     # :365


def context_forward(cp, cp_obs, cp_act, stats, hidden_act=relu, out_act=ACTIVATIONS[None]):
    """utils.py:401-406 + :614-622.  cp_obs [m,D*Hh], cp_act [m,A*Hh] -> ctx [E,m,C].
    The reference always uses ReLU here (layers.py:34; the ctor kwarg is ignored,
    SURVEY.md Appendix C)."""
    E = cp["cp_output_weight"].shape[0]
    bo = np.tile(cp_obs[None], (E, 1, 1))
    ba = np.tile(cp_act[None], (E, 1, 1))
    return context_forward_bs(cp, bo, ba, stats, hidden_act, out_act)


def context_forward_bs(cp, bs_cp_obs, bs_cp_act, stats, hidden_act=relu, out_act=ACTIVATIONS[None]):
    """Training-graph form (utils.py:619-622): inputs already [E,B,.]."""
    x = np.concatenate([normalize(bs_cp_obs, stats["cp_obs_mean"], stats["cp_obs_std"]),
                        normalize(bs_cp_act, stats["cp_act_mean"], stats["cp_act_std"])], axis=-1)
    n = sum(1 for k in cp if k.startswith("cp_hidden_") and k.endswith("_weight"))
    for i in range(n):
        x = dense(x, cp["cp_hidden_%d_weight" % i], cp["cp_hidden_%d_bias" % i], hidden_act)
    return dense(x, cp["cp_output_weight"], cp["cp_output_bias"], out_act)


STAT_KEYS = ("obs_mean", "obs_std", "act_mean", "act_std", "delta_mean", "delta_std",
             "cp_obs_mean", "cp_obs_std", "cp_act_mean", "cp_act_std",
             "back_delta_mean", "back_delta_std")


def cast_stats(stats, dtype):
    return {k: np.asarray(v, dtype=dtype) for k, v in stats.items()}
