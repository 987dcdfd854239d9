#!/usr/bin/env python3
"""Golden vectors for the PLANNER GRAPH WIRING, executed by the reference's own graph-building code.

What this is: /root/reference/cadm/dynamics/core/utils.py (`create_ensemble_pure_context_predictor`,
`create_plus_cadm_ensemble_cem_mlp`, `create_plus_ensemble_cem_mlp`, `create_dense_layer`, `normalize`) and the env closures of
/root/reference/cadm/envs/*.py (every env kind) are imported UNCHANGED and run.  The reference builds a TensorFlow 1.15
graph; TensorFlow is not installed here, so `tensorflow` is replaced by the numpy-EAGER stand-in below: every `tf.*` call
the graph builder makes is mapped to the numpy function with the same published semantics (float32), "placeholders" are
concrete arrays, and the unrolled CEM graph therefore computes its result while it is being "built".

What this is NOT: TensorFlow.  The tile / transpose / reshape / concat chain (quirks Q1 / Q2), the member assignment, the
CEM update, the variable creation order and the order in which random draws are consumed are the reference's own lines; the
arithmetic of each op is numpy's.  DESIGN.md therefore still labels the planner oracle "parity unpinned" in the strict sense;
this golden removes the risk hand-restatement cannot: a wrong reading of the graph's data movement.

Run in the build container only:   python tests/golden/make_graph_golden.py        -> tests/golden/graph_golden.npz
(the outputs; the inputs are regenerated from tests/golden/graph_inputs.py on both sides).
"""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import graph_inputs as gi  # noqa: E402

REF = "/root/reference"
F32 = np.float32


def _a(x):
    return x if isinstance(x, np.ndarray) else np.asarray(x)


def build_tf(weights, draws):
    """numpy-eager stand-in for the `tensorflow` names core/utils.py and the env closures use."""
    tf = types.ModuleType("tensorflow")
    tf.float32, tf.int32 = np.float32, np.int32
    tf.reshape = lambda x, shape, name=None: np.reshape(_a(x), [int(s) for s in shape])
    tf.tile = lambda x, multiples, name=None: np.tile(_a(x), [int(s) for s in multiples])
    tf.transpose = lambda x, perm=None: np.transpose(_a(x), perm)
    tf.concat = lambda values, axis, name=None: np.concatenate([_a(v) for v in values], axis=axis)
    tf.reduce_mean = lambda x, axis=None: np.mean(_a(x), axis=axis, dtype=_a(x).dtype)
    tf.reduce_sum = lambda x, axis=None: np.sum(_a(x), axis=axis, dtype=_a(x).dtype)
    tf.square, tf.sqrt, tf.exp, tf.sin, tf.cos = np.square, np.sqrt, np.exp, np.sin, np.cos
    tf.minimum, tf.multiply = np.minimum, lambda a, b, name=None: a * b
    tf.identity = lambda x: x
    tf.sigmoid = lambda x: F32(1) / (F32(1) + np.exp(-x))
    tf.tanh = np.tanh
    tf.shape = lambda x: tuple(int(s) for s in _a(x).shape)
    tf.range = lambda start, limit=None, delta=1: np.arange(start, limit, delta, dtype=np.int32)
    tf.gather = lambda params, indices: _a(params)[_a(indices)]
    tf.argmax = lambda x, axis, output_type=np.int64: np.argmax(x, axis=axis).astype(output_type)
    tf.one_hot = lambda idx, depth: np.eye(depth, dtype=F32)[idx]
    tf.matmul = lambda a, b: np.matmul(a, b)
    tf.cast = lambda x, dtype=None, **kw: _a(x).astype(dtype)
    tf.logical_and, tf.less, tf.greater = np.logical_and, np.less, np.greater
    tf.clip_by_value = lambda x, lo, hi: np.clip(x, F32(lo), F32(hi))

    def softplus(x):       # tf.nn.softplus, TF 1.15 Eigen functor: threshold = log(eps) + 2
        x = _a(x)
        thr = F32(np.log(np.finfo(np.float32).eps)) + F32(2.0)
        with np.errstate(over="ignore"):
            ex = np.exp(x)
        return np.where(x > -thr, x, np.where(x < thr, ex, np.log1p(ex))).astype(x.dtype)

    topk_log = []          # (input, indices) of every tf.nn.top_k the graph builder calls: the candidate returns and elite sets

    def top_k(x, k, sorted=True):      # descending, ties -> lower index first
        idx = np.argsort(-_a(x), axis=-1, kind="stable")[..., :k]
        topk_log.append((np.array(x, np.float32), idx.astype(np.int32)))
        return np.take_along_axis(_a(x), idx, axis=-1), idx.astype(np.int32)

    tf._topk_log = topk_log

    tf.nn = types.SimpleNamespace(softplus=softplus, top_k=top_k, relu=lambda x: np.maximum(x, F32(0)),
                                  l2_loss=lambda w: np.sum(np.square(w), dtype=np.float32) / F32(2), softmax=None)
    tf.math = types.SimpleNamespace(log=np.log, atan2=np.arctan2)

    def truncated_normal(shape, mean=0.0, stddev=1.0, **kw):      # mean + stddev * Z, Z ~ N(0,1) re-drawn beyond 2 sigma
        return (mean + stddev * draws.truncated(shape)).astype(np.float32)

    def normal(shape, **kw):
        return draws.normal(shape)

    def uniform(shape, minval=0, maxval=None, dtype=np.float32, **kw):
        if dtype == np.int32:                            # discrete random shooting: action indices in [0, maxval)
            return draws.randint(shape, maxval)
        return draws.uniform(shape, minval, maxval)

    tf.random = types.SimpleNamespace(truncated_normal=truncated_normal, normal=normal, uniform=uniform)
    # variables: the creation order IS tf.trainable_variables() order (what save / load rely on)
    scope = []

    created = {}           # full name -> array: a second request for the same name returns the SAME variable (AUTO_REUSE)

    class _Scope:
        def __init__(self, name, reuse=None, **kw): self.name = name
        def __enter__(self): scope.append(self.name)
        def __exit__(self, *a): scope.pop()

    def get_variable(name, shape=None, initializer=None, **kw):
        full = "/".join(scope + [name])
        if full not in created:
            created[full] = weights.dense(full, tuple(int(s) for s in shape))
        return created[full]

    def Variable(value, dtype=None, name=None, **kw):      # tf.Variable never reuses: a second call gets a uniquified name
        full, k = "/".join(scope + [name]), 0
        while full in created:
            k += 1
            full = "/".join(scope + ["%s_%d" % (name, k)])
        created[full] = weights.plain(full, value)
        return created[full]

    tf._scope, tf._created = scope, created

    tf.Variable = Variable
    tf.compat = types.SimpleNamespace(v1=types.SimpleNamespace(get_variable=get_variable, variable_scope=_Scope))
    tf.variable_scope = _Scope
    tf.truncated_normal_initializer = lambda stddev=1.0, **k: ("truncated_normal", stddev)
    tf.constant_initializer = lambda v=0.0: ("constant", v)
    tf.zeros_initializer = lambda: ("zeros",)
    tf.contrib = types.SimpleNamespace(layers=types.SimpleNamespace(xavier_initializer=lambda: ("xavier",)))
    return tf


class _Placeholder(types.ModuleType):
    """Import-only placeholder for packages the env / utils modules import but this path never uses: attribute access
    yields an empty class (CamelCase names, so `class X(gym.Env)` parses) or a nested placeholder."""
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        child = type(name, (), {}) if name[:1].isupper() else _Placeholder(self.__name__ + "." + name)
        setattr(self, name, child)
        return child


class _PlaceholderFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = ("baselines", "gym", "mujoco_py", "tensorboardX", "pyprind", "mpi4py")

    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Placeholder(spec.name)

    def exec_module(self, module):
        pass


ENVS = {"halfcheetah": ("cadm.envs.half_cheetah_env", "HalfCheetahEnv"),
        "cripple_halfcheetah": ("cadm.envs.half_cheetah_cripple_env", "CrippleHalfCheetahEnv"),
        "ant": ("cadm.envs.ant_env", "AntEnv"), "slim_humanoid": ("cadm.envs.slim_humanoid_env", "SlimHumanoidEnv"),
        "cartpole": ("cadm.envs.classic_control", "RandomCartPole_Force_Length"),
        "pendulum": ("cadm.envs.classic_control", "RandomPendulumAll")}


def run_case(case):
    c = gi.CASES[case]
    inp = gi.make_inputs(case)
    weights, draws = gi.Weights(c["seed"]), gi.Draws(c["seed"])
    for k in [k for k in sys.modules if k == "tensorflow" or k.startswith("cadm")]:
        del sys.modules[k]
    sys.modules["tensorflow"] = tf = build_tf(weights, draws)
    if not any(isinstance(f, _PlaceholderFinder) for f in sys.meta_path):
        sys.meta_path.append(_PlaceholderFinder())          # consulted only after the real finders fail
    if REF not in sys.path:
        sys.path.insert(0, REF)
    U = importlib.import_module("cadm.dynamics.core.utils")                 # the reference's graph builder, unchanged
    mod, cls = ENVS[c["env"]]
    Env = getattr(importlib.import_module(mod), cls)                         # the env class's closures, unchanged
    me = types.SimpleNamespace(max_torque=2.0)                               # `self` of the closures (pendulum: classic_control.py:185)
    pre = lambda o: Env.obs_preproc(me, o)
    post = lambda o, d: Env.obs_postproc(me, o, d)
    reward = Env.tf_reward_fn(me)
    swish = lambda x: x * tf.sigmoid(x)                                     # mlp_cadm_ensemble_cem_dynamics.py:23
    st = inp["stats"]
    D, A, E = c["D"], c["A"], c["E"]
    wd = (0.000025, 0.00005, 0.000075, 0.000075, 0.0001)
    if c.get("vanilla"):                                                    # mlp_ensemble_cem_dynamics.py:92-130
        with tf.compat.v1.variable_scope("ff_model"):
            out = U.create_plus_ensemble_cem_mlp(
                output_dim=D, hidden_sizes=c["hidden"], hidden_nonlinearity=swish, output_nonlinearity=tf.identity, input_obs_dim=D,
                input_act_dim=A, input_obs_var=inp["obs"], input_act_var=None, n_forwards=c["H"], ensemble_size=E, weight_decays=wd,
                reward_fn=reward, n_candidates=c["n"], norm_obs_mean_var=st["obs_mean"], norm_obs_std_var=st["obs_std"],
                norm_act_mean_var=st["act_mean"], norm_act_std_var=st["act_std"], norm_delta_mean_var=st["delta_mean"],
                norm_delta_std_var=st["delta_std"], n_particles=c["p"], bs_input_obs_var=inp["bs_obs"], bs_input_act_var=inp["bs_act"],
                cem_init_mean_var=inp["init_mean"], cem_init_var_var=inp["init_var"], obs_preproc_fn=pre, obs_postproc_fn=post,
                deterministic=False)
        (_, _, output_var, optimal_action, mu, logvar, max_lv, min_lv, l2_regs) = out
        return {case + "/cand_returns": np.stack([x for x, _ in tf._topk_log]), case + "/elites": np.stack([i for _, i in tf._topk_log]),
                case + "/plan": np.asarray(optimal_action, np.float32), case + "/train_mu": np.asarray(mu, np.float32),
                case + "/train_logvar": np.asarray(logvar, np.float32), case + "/train_output": np.asarray(output_var, np.float32),
                case + "/var_names": np.array([n for n, _ in weights.vars]),
                case + "/var_shapes": np.array([",".join(map(str, v.shape)) for _, v in weights.vars]),
                case + "/draw_kinds": np.array([k for k, _ in draws.log]),
                case + "/draw_shapes": np.array([",".join(map(str, z.shape)) for _, z in draws.log])}
    with tf.compat.v1.variable_scope("context_model"):                      # dynamics.py:140-156
        bs_cp, _, cp_forward = U.create_ensemble_pure_context_predictor(
            context_hidden_sizes=c["cp_hidden"], context_hidden_nonlinearity=tf.nn.relu, output_nonlinearity=tf.identity,
            ensemble_size=E, cp_input_dim=(D + A) * c["Hh"], context_weight_decays=(0.000025, 0.00005, 0.000075, 0.000075),
            bs_input_cp_obs_var=inp["bs_cp_obs"], bs_input_cp_act_var=inp["bs_cp_act"], norm_cp_obs_mean_var=st["cp_obs_mean"],
            norm_cp_obs_std_var=st["cp_obs_std"], norm_cp_act_mean_var=st["cp_act_mean"], norm_cp_act_std_var=st["cp_act_std"],
            cp_output_dim=c["C"])
    with tf.compat.v1.variable_scope("ff_model"):                           # dynamics.py:159-211
        out = U.create_plus_cadm_ensemble_cem_mlp(
            output_dim=D, hidden_sizes=c["hidden"], hidden_nonlinearity=swish, output_nonlinearity=tf.identity, input_obs_dim=D,
            input_act_dim=A, input_obs_var=inp["obs"], input_act_var=None, input_cp_obs_var=inp["cp_obs"],
            input_cp_act_var=inp["cp_act"], n_forwards=c["H"], ensemble_size=E, weight_decays=wd, reward_fn=reward,
            n_candidates=c["n"], norm_obs_mean_var=st["obs_mean"], norm_obs_std_var=st["obs_std"], norm_act_mean_var=st["act_mean"],
            norm_act_std_var=st["act_std"], norm_delta_mean_var=st["delta_mean"], norm_delta_std_var=st["delta_std"],
            norm_cp_obs_mean_var=st["cp_obs_mean"], norm_cp_obs_std_var=st["cp_obs_std"], norm_cp_act_mean_var=st["cp_act_mean"],
            norm_cp_act_std_var=st["cp_act_std"], n_particles=c["p"], bs_input_obs_var=inp["bs_obs"], bs_input_act_var=inp["bs_act"],
            bs_input_cp_var=bs_cp, cp_output_dim=c["C"], history_length=c["Hh"],
            cem_init_mean_var=None if c.get("rs") else inp["init_mean"], cem_init_var_var=None if c.get("rs") else inp["init_var"],
            obs_preproc_fn=pre, obs_postproc_fn=post, deterministic=False, discrete=bool(c.get("discrete")),
            build_policy_graph=True, cp_forward=cp_forward)
    (_, _, output_var, optimal_action, mu, logvar, max_lv, min_lv, l2_regs, inference_cp, _) = out
    res = {case + "/plan": np.asarray(optimal_action, np.float32), case + "/context": np.asarray(inference_cp, np.float32),
           case + "/train_mu": np.asarray(mu, np.float32), case + "/train_logvar": np.asarray(logvar, np.float32),
           case + "/train_output": np.asarray(output_var, np.float32), case + "/bs_cp": np.asarray(bs_cp, np.float32),
           case + "/var_names": np.array([n for n, _ in weights.vars]),
           case + "/var_shapes": np.array([",".join(map(str, v.shape)) for _, v in weights.vars]),
           case + "/draw_kinds": np.array([k for k, _ in draws.log]),
           case + "/draw_shapes": np.array([",".join(map(str, z.shape)) for _, z in draws.log])}
    if tf._topk_log:         # CEM: per iteration the [m, n] particle-mean returns the graph ranked and the 50 indices it kept
        res[case + "/cand_returns"] = np.stack([x for x, _ in tf._topk_log])
        res[case + "/elites"] = np.stack([i for _, i in tf._topk_log])
    return res


def main():
    out = {}
    for case in gi.CASES:
        out.update(run_case(case))
    np.savez_compressed(os.path.join(HERE, "graph_golden.npz"), **out)
    for k in out:
        if k.endswith("/plan") or k.endswith("/draw_kinds"):
            print(k, out[k].shape)
    print("variables:", list(out["hc_cadm_m2/var_names"]))


if __name__ == "__main__":
    main()
