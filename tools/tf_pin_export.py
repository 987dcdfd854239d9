#!/usr/bin/env python3
"""Export a TRUE TensorFlow-1.15 pin of the reference's hot path -- run by someone who HAS the reference's stack.

Why this exists.  cadm_amd's oracle restates the reference's arithmetic in numpy / torch, and the golden fixtures under tests/golden/
execute the reference's own Python on a numpy stand-in for `tensorflow`: wiring, quirks, variable order and loss composition are the
reference's, but every op's arithmetic is numpy's ("parity strictly unpinned").  TensorFlow 1.15, gym and mujoco-py are not
installable where cadm_amd is built.  This script closes the gap from the other side: on a machine with the reference checkout and its
dependencies it builds the reference's OWN `MLPEnsembleCEMDynamicsModel` in a real `tf.Session`, feeds it fixed seeded inputs and writes
every DETERMINISTIC quantity of the path into one `.npz`:

  * the model's parameters as `save()` would write them (tf.trainable_variables() order) and the normalisation statistics fed;
  * `get_context_pred(cp_obs, cp_act)`                                   (dynamics.py:369-380, core/utils.py:569-624)
  * the training graph's scalars on a fixed bootstrap batch: mse_loss, back_mse_loss, mu_loss, var_loss, reg_loss, l2_reg_loss,
    context_l2_reg_loss, back_l2_reg_loss, recon_loss, loss             (dynamics.py:269-314)
  * the parameters after 1 and after 3 runs of `train_op` on that batch  (TF's autodiff + tf.train.AdamOptimizer: gradients and Adam)

(The planner's outputs are not deterministic in the reference -- TF's RNG is never seeded, `set_seed` is dead code -- so `get_action` is
pinned at these deterministic boundaries, like everywhere else in the test-suite.)

The file is DATA (arrays and a JSON string).  Drop it at tests/golden/tf_pin.npz (or point $CADM_TF_PIN at it) and run
`pytest tests/test_tf_pin.py`: the oracle (CPU) and the HIP kernels (-m gpu) are then held to real TensorFlow at 1e-5.

  python tools/tf_pin_export.py --reference /path/to/CaDM --dataset halfcheetah --out tf_pin.npz [--batch 64] [--seed 0]
                                [--hidden_size 200] [--deterministic_flag 0] [--normalize_flag]
Requires (the reference's own requirements): tensorflow==1.15, gym==0.16, mujoco-py 2.0.2.9 (for the MuJoCo datasets; cartpole /
pendulum need only gym), joblib.  Nothing from cadm_amd is imported: the script is self-contained on purpose.
"""
import argparse
import json
import os
import sys

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True, help="path of the reference checkout (the directory that holds cadm/ and run_scripts/)")
    ap.add_argument("--dataset", default="halfcheetah", help="halfcheetah | cripple_halfcheetah | ant | slim_humanoid | cartpole | pendulum")
    ap.add_argument("--out", default="tf_pin.npz")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--batch", type=int, default=64, help="rows per member of the fixed bootstrap batch")
    ap.add_argument("--m", type=int, default=3, help="histories for get_context_pred")
    ap.add_argument("--hidden_size", type=int, default=200)
    ap.add_argument("--context_out_dim", type=int, default=10)
    ap.add_argument("--ensemble_size", type=int, default=5)
    ap.add_argument("--history_length", type=int, default=10)
    ap.add_argument("--back_coeff", type=float, default=0.5)
    ap.add_argument("--deterministic_flag", type=int, default=0)
    ap.add_argument("--normalize_flag", action="store_true")
    ap.add_argument("--lr", type=float, default=0.001)
    args = ap.parse_args()

    sys.path.insert(0, os.path.abspath(args.reference))
    import tensorflow as tf
    assert tf.__version__.startswith("1.15"), "the reference runs on TensorFlow 1.15 (found %s)" % tf.__version__
    np.random.seed(args.seed)
    tf.compat.v1.set_random_seed(args.seed)

    # the environment object exactly as the reference's scripts build it (cadm/envs/config.py: get_environment_config)
    from cadm.envs.config import get_environment_config
    config = {
        "dataset": args.dataset, "normalize_flag": args.normalize_flag, "seed": args.seed, "n_candidates": 200, "horizon": 30,
        "use_cem": True, "max_path_length": 200, "num_rollouts": 10, "n_parallel": 5,
        "hidden_sizes": (args.hidden_size,) * 4, "hidden_nonlinearity": "swish", "deterministic": args.deterministic_flag > 0,
        "weight_decays": (0.000025, 0.00005, 0.000075, 0.000075, 0.0001), "weight_decay_coeff": 1.0,
        "ensemble_size": args.ensemble_size, "n_particles": 20,
        "context_hidden_sizes": (256, 128, 64), "context_weight_decays": (0.000025, 0.00005, 0.000075),
        "context_out_dim": args.context_out_dim, "context_hidden_nonlinearity": "relu", "history_length": args.history_length,
        "future_length": 10, "state_diff": 1, "back_coeff": args.back_coeff, "learning_rate": args.lr, "batch_size": 256,
        "valid_split_ratio": 0.1, "rolling_average_persitency": 0.99, "save_name": "tf_pin/", "total_test": 0, "test_range": None,
        "num_test": 0, "no_test_flag": True, "only_test_flag": False,
    }
    env, config = get_environment_config(config)
    from cadm.dynamics.mlp_cadm_ensemble_cem_dynamics import MLPEnsembleCEMDynamicsModel

    sess = tf.compat.v1.Session()
    with sess.as_default():
        model = MLPEnsembleCEMDynamicsModel(
            name="dyn_model", env=env, learning_rate=config["learning_rate"], hidden_sizes=config["hidden_sizes"],
            valid_split_ratio=config["valid_split_ratio"], rolling_average_persitency=config["rolling_average_persitency"],
            hidden_nonlinearity=config["hidden_nonlinearity"], batch_size=config["batch_size"], normalize_input=True,
            n_forwards=config["horizon"], n_candidates=config["n_candidates"], ensemble_size=config["ensemble_size"],
            n_particles=config["n_particles"], use_cem=config["use_cem"], deterministic=config["deterministic"],
            weight_decays=config["weight_decays"], weight_decay_coeff=config["weight_decay_coeff"],
            cp_hidden_sizes=config["context_hidden_sizes"], context_weight_decays=config["context_weight_decays"],
            context_out_dim=config["context_out_dim"], context_hidden_nonlinearity=config["context_hidden_nonlinearity"],
            history_length=config["history_length"], future_length=config["future_length"], state_diff=config["state_diff"],
            back_coeff=config["back_coeff"])
        sess.run(tf.compat.v1.global_variables_initializer())

        E, B, Hh = args.ensemble_size, args.batch, args.history_length
        D = int(np.prod(env.observation_space.shape))
        A = int(np.prod(env.action_space.shape)) if hasattr(env.action_space, "shape") and env.action_space.shape else int(env.action_space.n)
        P = int(env.proc_observation_space_dims)
        rng = np.random.RandomState(args.seed + 1)

        # trained-like parameters: the initialiser leaves every bias at 0 and the heads tiny; give them structure so that every term
        # of the graph is exercised (assigned through the variables themselves: what `load` does, dynamics.py:579-588)
        params0 = sess.run(model.params)
        new = []
        for v, a in zip(model.params, params0):
            if "bias" in v.name:
                a = a + 0.1 * rng.standard_normal(a.shape).astype(np.float32)
            elif "output_" in v.name and a.ndim == 3:
                a = a * 2.0
            new.append(a.astype(np.float32))
        for v, a in zip(model.params, new):
            v.load(a, sess)

        def ms(k):
            return rng.standard_normal(k).astype(np.float32), rng.uniform(0.5, 2.0, k).astype(np.float32)
        # statistics: stored on the model as `compute_normalization` would (dynamics.py:590-602), then read back through the model's own
        # `get_normalization_stats` (:604-650: with state_diff the history's obs statistics are 0 / 1, discrete actions likewise) --
        # what the graph is fed below is what `fit` and `get_action` would feed
        model.normalization = {}
        for key, k in (("obs", P), ("act", A), ("delta", D), ("cp_obs", D * Hh), ("cp_act", A * Hh), ("back_delta", D)):
            model.normalization[key] = ms(k)
        tup = model.get_normalization_stats()
        order = ("obs", "act", "delta", "cp_obs", "cp_act", "back_delta")
        stats = {}
        for i, key in enumerate(order):
            stats[key + "_mean"] = np.asarray(tup[2 * i], np.float32)
            stats[key + "_std"] = np.asarray(tup[2 * i + 1], np.float32)
        batch = {
            "obs": rng.standard_normal((E, B, D)).astype(np.float32),
            "act": rng.uniform(-1, 1, (E, B, A)).astype(np.float32),
            "delta": rng.standard_normal((E, B, D)).astype(np.float32),
            "back_delta": rng.standard_normal((E, B, D)).astype(np.float32),
            "cp_obs": (0.1 * rng.standard_normal((E, B, D * Hh))).astype(np.float32),
            "cp_act": rng.uniform(-1, 1, (E, B, A * Hh)).astype(np.float32),
        }
        batch["obs_next"] = (batch["obs"] + 0.1 * rng.standard_normal((E, B, D))).astype(np.float32)
        feed = {getattr(model, "bs_%s_ph" % k): v for k, v in batch.items()}
        feed.update({getattr(model, "norm_%s_ph" % k): v for k, v in stats.items()})

        out = {"config_json": np.array(json.dumps({k: (list(v) if isinstance(v, tuple) else v) for k, v in config.items()
                                                   if isinstance(v, (int, float, str, bool, tuple, list, type(None)))})),
               "tf_version": np.array(tf.__version__), "dataset": np.array(args.dataset), "D": D, "A": A, "P": P, "E": E, "B": B, "Hh": Hh,
               "n_params": len(model.params), "param_names": np.array([v.name for v in model.params])}
        for i, a in enumerate(sess.run(model.params)):
            out["param_%03d" % i] = a
        for k, v in stats.items():
            out["stat_" + k] = v
        for k, v in batch.items():
            out["batch_" + k] = v

        # (1) get_context_pred through the class method (it reads the statistics from the model object, as get_action does)
        cp_obs = (0.1 * rng.standard_normal((args.m, D * Hh))).astype(np.float32)
        cp_act = rng.uniform(-1, 1, (args.m, A * Hh)).astype(np.float32)
        out["ctx_cp_obs"], out["ctx_cp_act"] = cp_obs, cp_act
        out["ctx_pred"] = np.asarray(model.get_context_pred(cp_obs, cp_act))

        # (2) the training graph's scalars on the fixed batch
        names = ["mse_loss", "back_mse_loss", "l2_reg_loss", "context_l2_reg_loss", "recon_loss", "loss"]
        names += [n for n in ("mu_loss", "var_loss", "reg_loss", "back_l2_reg_loss") if hasattr(model, n)]
        vals = sess.run([getattr(model, n) for n in names], feed_dict=feed)
        for n, v in zip(names, vals):
            out["loss_" + n] = np.float64(v)

        # (3) Adam: parameters after 1 and after 3 training steps on the same batch
        for step in (1, 2, 3):
            sess.run(model.train_op, feed_dict=feed)
            if step in (1, 3):
                for i, a in enumerate(sess.run(model.params)):
                    out["after%d_param_%03d" % (step, i)] = a
        out["adam"] = np.array([args.lr, 0.9, 0.999, 1e-8])      # tf.train.AdamOptimizer defaults (dynamics.py:316)

    np.savez_compressed(args.out, **out)
    print("wrote %s: %d parameter tensors, %d loss scalars, context pred %s, parameters after 1 and 3 Adam steps"
          % (args.out, len(model.params), len(names), out["ctx_pred"].shape))


if __name__ == "__main__":
    main()
