#!/usr/bin/env python3
"""Golden values for the TRAINING LOSS COMPOSITION and the model's VARIABLE LIST, produced by the reference's own
constructor.

`MLPEnsembleCEMDynamicsModel.__init__` (/root/reference/cadm/dynamics/mlp_cadm_ensemble_cem_dynamics.py:26-342) is
imported unchanged and run on the numpy-eager `tensorflow` stand-in of make_graph_golden.py (extended here with
placeholders fed in creation order, variable reuse, a do-nothing optimizer).  Because the stand-in is eager, the
constructor COMPUTES what it would normally only wire up: the context / forward / backward networks on a bootstrap batch
and, with the reference's own lines :269-313, mse_loss, back_mse_loss, the three l2 terms, mu_loss, var_loss, reg_loss,
recon_loss and loss.  It also shows which variables the constructor creates, in tf.trainable_variables() order, and the
constructed object's own `_preprocess_inputs`, `compute_normalization` and `get_normalization_stats` (pure numpy) are run
on windowed samples as `fit` would.

What this is not: TensorFlow (see make_graph_golden.py).  Run in the build container only:
    python tests/golden/make_loss_golden.py        -> tests/golden/loss_golden.npz
"""
import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import graph_inputs as gi  # noqa: E402
import make_graph_golden as mg  # noqa: E402

REF = "/root/reference"


class _PH(np.ndarray):
    """A fed placeholder: hashable by identity (`fit` uses placeholders as feed_dict keys); anything computed FROM it is a plain
    array / numpy scalar again (a 0-d subclass instance would make the constructor's `recon_loss += ...` an in-place update of
    `mse_loss`, which tensors are not)."""
    __hash__ = object.__hash__

    def __eq__(self, other):
        return self is other if isinstance(other, _PH) else np.ndarray.__eq__(self, other)

    def __array_wrap__(self, out_arr, context=None, return_scalar=False):
        out = np.asarray(out_arr)
        return out[()] if out.ndim == 0 else out

    def __getitem__(self, idx):
        return np.asarray(self)[idx]


def run_fit(model, tf, c, case):
    """The reference's own `fit` (dynamics.py:382-569) on the constructed model, with a session that records what it is fed
    and answers with SCRIPTED losses (no TensorFlow, no training): pins the host loop -- targets, dataset columns,
    normalisation, window split, `_preprocess_inputs`, bootstrap indices, per-epoch `shuffle_rows`, batch slicing, the
    validation batch and the rolling-average early stop -- and the order in which it consumes `np.random`."""
    d = gi.make_fit_inputs(case, N=23)
    N, F, D, A = d["obs"].shape[0], c["F"], c["D"], c["A"]
    draws = []
    real = {k: getattr(np.random, k) for k in ("permutation", "randint", "uniform")}

    def logged(kind):
        def f(*a, **k):
            out = real[kind](*a, **k)
            draws.append((kind, np.array(out)))
            return out
        return f

    v_script = [10.0, 8.0, 7.0, 6.5, 6.4, 6.45, 6.8, 7.5, 9.0, 12.0, 15.0, 20.0]
    fed = {"train": [], "valid": []}
    keys = ("bs_obs_ph", "bs_act_ph", "bs_delta_ph", "bs_obs_next_ph", "bs_back_delta_ph", "bs_cp_obs_ph", "bs_cp_act_ph")
    stats_keys = ("norm_obs_mean_ph", "norm_obs_std_ph", "norm_act_mean_ph", "norm_act_std_ph", "norm_delta_mean_ph", "norm_delta_std_ph",
                  "norm_cp_obs_mean_ph", "norm_cp_obs_std_ph", "norm_cp_act_mean_ph", "norm_cp_act_std_ph", "norm_back_delta_mean_ph",
                  "norm_back_delta_std_ph")

    class Session:
        def run(self, fetches, feed_dict=None):
            batch = [np.array(feed_dict[getattr(model, k)], np.float64) for k in keys]
            stats = [np.array(feed_dict[getattr(model, k)], np.float64) for k in stats_keys]
            if len(fetches) == 4:                    # a training step: [mse, back_mse, recon, train_op]
                fed["train"].append((batch, stats))
                return 1.0, 0.5, 1.25, None
            fed["valid"].append((batch, stats))
            return 1.0, 0.5, v_script[len(fed["valid"]) - 1]

    tf.compat.v1.get_default_session = lambda: Session()
    np.random.seed(c["seed"])
    for k in real:
        setattr(np.random, k, logged(k))
    try:
        model._dataset = None
        model.batch_size = 16
        model.fit(d["obs"].reshape(N, -1), d["act"].reshape(N, -1), d["obs_next"].reshape(N, -1), d["cp_obs"], d["cp_act"], d["future_bool"],
                  epochs=len(v_script), valid_split_ratio=0.25, rolling_average_persitency=0.9)
    finally:
        for k, f in real.items():
            setattr(np.random, k, f)
    res = {case + "/fit_draw_kinds": np.array([k for k, _ in draws]), case + "/fit_v_script": np.array(v_script),
           case + "/fit_epochs_run": np.int64(len(fed["valid"])), case + "/fit_steps_per_epoch": np.int64(len(fed["train"]) // len(fed["valid"]))}
    for i, (k, v) in enumerate(draws):
        res[case + "/fit_draw_%02d" % i] = v
    names = ("obs", "act", "delta", "obs_next", "back_delta", "cp_obs", "cp_act")
    for which in ("train", "valid"):
        for j, nm in enumerate(names):          # batches of one kind have equal shapes except the last of an epoch: store flattened rows
            res[case + "/fit_%s_%s" % (which, nm)] = np.concatenate([b[j].reshape(b[j].shape[0], -1, b[j].shape[-1]) for b, _ in fed[which]], axis=1)
        res[case + "/fit_%s_sizes" % which] = np.array([b[0].shape[1] for b, _ in fed[which]], np.int64)
    for k, v in zip(stats_keys, fed["train"][0][1]):
        res[case + "/fit_stat_" + k[5:-3]] = v
    return res


def run_case(case):
    c = gi.LOSS_CASES[case]
    inp = gi.make_loss_inputs(case)
    weights, draws = gi.Weights(c["seed"]), gi.Draws(c["seed"])
    for k in [k for k in sys.modules if k == "tensorflow" or k.startswith("cadm")]:
        del sys.modules[k]
    sys.modules["tensorflow"] = tf = mg.build_tf(weights, draws)
    tf.nn.tanh, tf.nn.softmax = np.tanh, (lambda x: x)            # default arguments of the layer classes, unused on this path
    if not any(isinstance(f, mg._PlaceholderFinder) for f in sys.meta_path):
        sys.meta_path.append(mg._PlaceholderFinder())
    if REF not in sys.path:
        sys.path.insert(0, REF)
    # ---- what the constructor needs beyond the graph builder ----
    feed = list(gi.placeholder_feed(inp, vanilla=bool(c.get("vanilla"))))     # arrays in the order the constructor creates its placeholders

    def placeholder(dtype, shape=None, name=None):
        arr = feed.pop(0)
        assert len(shape) == arr.ndim and all(s is None or int(s) == d for s, d in zip(shape, arr.shape)), (shape, arr.shape)
        return arr.view(_PH)                         # hashable by identity: `fit` uses placeholders as feed_dict keys

    v1 = tf.compat.v1
    v1.placeholder = placeholder
    v1.AUTO_REUSE = object()
    v1.GraphKeys = types.SimpleNamespace(TRAINABLE_VARIABLES="trainable_variables")
    v1.get_default_graph = lambda: types.SimpleNamespace(get_name_scope=lambda: "/".join(tf._scope))
    v1.get_collection = lambda key, scope=None: []
    v1.trainable_variables = lambda: [v for _, v in weights.vars]
    v1.get_default_session = lambda: None

    class _Opt:
        def __init__(self, lr): self.lr = lr
        def minimize(self, loss): return None

    v1.train = types.SimpleNamespace(AdamOptimizer=_Opt)
    tf.constant = lambda v, **k: np.float32(v)
    M = importlib.import_module("cadm.dynamics.mlp_cadm_ensemble_cem_dynamics")        # the reference model, unchanged
    mod, cls = mg.ENVS[c["env"]]
    Env = getattr(importlib.import_module(mod), cls)
    env = types.SimpleNamespace(observation_space=types.SimpleNamespace(shape=(c["D"],)), proc_observation_space_dims=c["P"],
                                action_space=types.SimpleNamespace(shape=(c["A"],)),
                                obs_preproc=lambda o: Env.obs_preproc(None, o), obs_postproc=lambda o, d: Env.obs_postproc(None, o, d),
                                tf_reward_fn=lambda: Env.tf_reward_fn(None))
    if c.get("vanilla"):
        MV = importlib.import_module("cadm.dynamics.mlp_ensemble_cem_dynamics")               # the vanilla reference model, unchanged
        model = MV.MLPEnsembleCEMDynamicsModel(
            "dyn_model", env, hidden_sizes=c["hidden"], hidden_nonlinearity="swish", n_forwards=c["H"], n_candidates=c["n"],
            ensemble_size=c["E"], n_particles=c["p"], use_cem=True, deterministic=c["deterministic"], weight_decays=c["weight_decays"],
            weight_decay_coeff=c["weight_decay_coeff"])
        assert not feed, "placeholders left unfed: %d" % len(feed)
        res = {case + "/var_names": np.array([n for n, _ in weights.vars]),
               case + "/var_shapes": np.array([",".join(map(str, v.shape)) for _, v in weights.vars])}
        for k in ("mse_loss", "l2_reg_loss", "mu_loss", "var_loss", "reg_loss", "recon_loss", "loss"):
            res[case + "/" + k] = np.asarray(getattr(model, k), np.float32)
        d = gi.make_fit_inputs(case)
        rows = [d[k].reshape(-1, d[k].shape[-1]) for k in ("obs", "act", "delta")]         # vanilla fit takes flat [N, .] samples
        model.compute_normalization(*rows)
        for k, v in zip(("obs_mean", "obs_std", "act_mean", "act_std", "delta_mean", "delta_std"), model.get_normalization_stats()):
            res[case + "/norm_" + k] = np.asarray(v, np.float64)
        return res
    model = M.MLPEnsembleCEMDynamicsModel(
        "dyn_model", env, hidden_sizes=c["hidden"], hidden_nonlinearity="swish", n_forwards=c["H"], n_candidates=c["n"],
        ensemble_size=c["E"], n_particles=c["p"], use_cem=True, deterministic=c["deterministic"], weight_decays=c["weight_decays"],
        weight_decay_coeff=c["weight_decay_coeff"], cp_hidden_sizes=c["cp_hidden"], context_weight_decays=c["context_weight_decays"],
        context_out_dim=c["C"], context_hidden_nonlinearity="relu", history_length=c["Hh"], future_length=c["F"], state_diff=True,
        back_coeff=c["back_coeff"])
    assert not feed, "placeholders left unfed: %d" % len(feed)
    res = {case + "/var_names": np.array([n for n, _ in weights.vars]),
           case + "/var_shapes": np.array([",".join(map(str, v.shape)) for _, v in weights.vars])}
    for k in ("mse_loss", "back_mse_loss", "l2_reg_loss", "context_l2_reg_loss", "back_l2_reg_loss", "l2_loss", "mu_loss", "var_loss",
              "reg_loss", "recon_loss", "loss"):
        if hasattr(model, k):
            res[case + "/" + k] = np.asarray(getattr(model, k), np.float32)
    res[case + "/delta_pred"] = np.asarray(model.delta_pred, np.float32)
    # fit()'s host-side data path, the reference's own methods on that model object (pure numpy, dynamics.py:590-696)
    d = gi.make_fit_inputs(case)
    rows = model._preprocess_inputs(d["obs"], d["act"], d["delta"], d["cp_obs"], d["cp_act"], d["future_bool"], d["obs_next"],
                                    d["back_delta"])
    for k, v in zip(("obs", "act", "delta", "obs_next", "back_delta", "cp_obs", "cp_act"), rows):
        res[case + "/rows_" + k] = np.asarray(v, np.float64)
    _obs, _act, _delta, _obs_next, _back_delta, _cp_obs, _cp_act = rows
    model.compute_normalization(_obs, _act, _delta, _cp_obs, _cp_act, _back_delta)
    for k, v in zip(("obs_mean", "obs_std", "act_mean", "act_std", "delta_mean", "delta_std", "cp_obs_mean", "cp_obs_std",
                     "cp_act_mean", "cp_act_std", "back_delta_mean", "back_delta_std"), model.get_normalization_stats()):
        res[case + "/norm_" + k] = np.asarray(v, np.float64)
    if case == "hc_cadm_prob":
        env.targ_proc = lambda o, n: Env.targ_proc(None, o, n)
        res.update(run_fit(model, tf, c, case))
    if c.get("save"):        # the reference's own save() (dynamics.py:571-577) writes the checkpoint fixture: joblib list + _norm_stats
        tf.compat.v1.get_default_session = lambda: types.SimpleNamespace(run=lambda fetches, feed_dict=None: [np.array(p) for p in fetches])
        os.makedirs(os.path.join(HERE, "ref_ckpt"), exist_ok=True)
        model.save(os.path.join(HERE, "ref_ckpt", "params_epoch_7"))
    return res


def main():
    out = {}
    for case in gi.LOSS_CASES:
        out.update(run_case(case))
    np.savez_compressed(os.path.join(HERE, "loss_golden.npz"), **out)
    for k, v in out.items():
        if v.ndim == 0:
            print(k, float(v))
    first = next(iter(gi.LOSS_CASES))
    print("variables:", list(out[first + "/var_names"]))


if __name__ == "__main__":
    main()
