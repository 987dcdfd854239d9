"""The TRAINING LOSS COMPOSITION and the model's VARIABLE LIST pinned to the reference's own constructor:
tests/golden/loss_golden.npz was written by /root/reference/cadm/dynamics/mlp_cadm_ensemble_cem_dynamics.py's
`MLPEnsembleCEMDynamicsModel.__init__`, imported unchanged and executed on the numpy-eager `tensorflow` stand-in
(tests/golden/make_loss_golden.py says what that does and does not prove).  Inputs are regenerated from
tests/golden/graph_inputs.py; the oracle's `train_losses` and the HIP training step (forward only) must reproduce the
reference's mse_loss / back_mse_loss / recon_loss / loss."""
import os
import sys
from collections import OrderedDict

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import graph_inputs as gi  # noqa: E402

GOLD = np.load(os.path.join(HERE, "golden", "loss_golden.npz"))
NETS = ("context_model", "ff_model", "backward_model")


def rebuild(case):
    """(config, nets, inputs) with the weights exactly as the generator handed them to the reference constructor."""
    c = gi.LOSS_CASES[case]
    W = gi.Weights(c["seed"])
    nets = {k: OrderedDict() for k in NETS}
    for name, shp in zip(GOLD[case + "/var_names"], GOLD[case + "/var_shapes"]):
        parts = str(name).split("/")                             # <model>/<net>/<layer scope>/<variable>; vanilla: no <net> level
        net, pname = (parts[1], parts[3]) if len(parts) == 4 else ("ff_model", parts[-1])
        shape = tuple(int(s) for s in str(shp).split(","))
        if pname.endswith(("_weight", "_bias")):
            nets[net][pname] = W.dense(str(name), shape)
        else:
            key = {"max_log_var": "max_logvar", "min_log_var": "min_logvar"}.get(pname, pname)      # the vanilla builder's spelling
            nets[net][key] = W.plain(str(name), np.ones(shape) / 2.0 if key == "max_logvar" else -np.ones(shape) * 10)
    return c, nets, gi.make_loss_inputs(case)


def test_constructor_variable_order_is_the_checkpoint_order():
    """`save` writes tf.trainable_variables() (dynamics.py:266,571-577): context_model, ff_model, backward_model, each dynamics
    net as hidden_i_{weight,bias}, output_mu_*, output_logvar_*, max_logvar, min_logvar -- ONE logvar pair per net, also for
    the deterministic backward model."""
    from cadm_amd.engine import ctx_param_names, dyn_param_names
    names = ["/".join((str(x).split("/")[1], str(x).split("/")[3])) for x in GOLD["hc_cadm_prob/var_names"]]
    want = (["context_model/" + x for x in ctx_param_names(3)] + ["ff_model/" + x for x in dyn_param_names(4)]
            + ["backward_model/" + x for x in dyn_param_names(4)])
    assert names == want


@pytest.mark.parametrize("case", sorted(gi.LOSS_CASES))
def test_oracle_losses_match_the_reference_constructor(case):
    import torch
    from oracle import train as otrain
    c, nets, inp = rebuild(case)
    t = lambda d: otrain.to_torch(d, torch.float32) if d else None
    batch = {k: torch.tensor(inp["bs_" + k]) for k in ("obs", "act", "delta", "obs_next", "back_delta", "cp_obs", "cp_act")}
    cfg = dict(deterministic=c["deterministic"], back_coeff=c["back_coeff"], weight_decay_coeff=c["weight_decay_coeff"],
               weight_decays=c["weight_decays"], context_weight_decays=c["context_weight_decays"], n_hidden=len(c["hidden"]),
               n_cp_hidden=len(c["cp_hidden"]))
    out = otrain.train_losses(c["env"], t(nets["ff_model"]), t(nets["backward_model"]), t(nets["context_model"]), t(inp["stats"]),
                              batch, cfg)
    pairs = [("mse", "mse_loss"), ("recon", "recon_loss"), ("loss", "loss")] + ([] if c.get("vanilla") else [("back_mse", "back_mse_loss")])
    for ok, gk in pairs:
        np.testing.assert_allclose(float(out[ok]), float(GOLD[case + "/" + gk]), rtol=2e-5, err_msg=gk)


@pytest.mark.parametrize("case", sorted(gi.LOSS_CASES))
def test_fit_rows_and_normalisation_match_the_reference_methods(case):
    """`_preprocess_inputs`, `compute_normalization`, `get_normalization_stats` (dynamics.py:590-696) as executed by the reference
    model object itself, against the oracle's restatement (which tests/test_oracle_train.py ties the product class to)."""
    from oracle import envs as oenvs
    from oracle import train as otrain
    c = gi.LOSS_CASES[case]
    d = gi.make_fit_inputs(case)
    if c.get("vanilla"):      # mlp_ensemble_cem_dynamics.py:344-372: three statistics over flat samples
        obs, act, delta = (d[k].reshape(-1, d[k].shape[-1]) for k in ("obs", "act", "delta"))
        env = oenvs.make_env(c["env"])
        for k, v in (("obs", env.obs_preproc(obs)), ("act", act), ("delta", delta)):
            np.testing.assert_allclose(v.mean(0), GOLD[case + "/norm_%s_mean" % k], rtol=0, atol=1e-12)
            np.testing.assert_allclose(v.std(0), GOLD[case + "/norm_%s_std" % k], rtol=0, atol=1e-12)
        return
    rows = otrain.preprocess_inputs(d["obs"], d["act"], d["delta"], d["cp_obs"], d["cp_act"], d["future_bool"], d["obs_next"],
                                    d["back_delta"], c["D"], c["A"], c["Hh"], c["F"])
    names = ("obs", "act", "delta", "obs_next", "back_delta", "cp_obs", "cp_act")
    for k, v in zip(names, rows):
        np.testing.assert_array_equal(v, GOLD[case + "/rows_" + k], err_msg=k)
    r = dict(zip(names, rows))
    norm = otrain.compute_normalization(oenvs.make_env(c["env"]), r["obs"], r["act"], r["delta"], r["cp_obs"], r["cp_act"], r["back_delta"])
    st = otrain.normalization_stats(norm, c["D"], c["A"], c["Hh"], discrete=False, state_diff=True)
    for k, v in st.items():
        np.testing.assert_allclose(v, GOLD[case + "/norm_" + k], rtol=0, atol=1e-12, err_msg=k)


def test_fit_host_loop_matches_the_reference_fit():
    """The reference's own `fit` (dynamics.py:382-569), run on the constructed model with a recording session that answers with
    scripted losses: every batch it fed (training and validation, all epochs), the statistics it fed, the order in which it
    consumed np.random, and the epoch at which its rolling-average early stop fired -- against the oracle's restatement
    replaying the SAME draws (tests/test_gpu_fit.py ties the product's fit to that restatement on the device)."""
    from cadm_amd.dynamics.mlp_cadm_ensemble_cem_dynamics import ReplayIndexStream
    from oracle import envs as oenvs
    from oracle import train as otrain
    case = "hc_cadm_prob"
    c = gi.LOSS_CASES[case]
    g = lambda k: GOLD[case + "/fit_" + k]
    kinds = [str(k) for k in g("draw_kinds")]
    epochs_run, per_epoch = int(g("epochs_run")), int(g("steps_per_epoch"))
    assert kinds == ["permutation", "randint"] + ["uniform"] * epochs_run            # :444, :465, then one shuffle_rows per epoch
    draws = [GOLD[case + "/fit_draw_%02d" % i] for i in range(len(kinds))]
    log = [("permutation", draws[0]), ("bootstrap", draws[1])] + [("epoch_order", np.argsort(u, axis=-1)) for u in draws[2:]]
    d = gi.make_fit_inputs(case, N=23)
    N = d["obs"].shape[0]
    data = {k: (d[k].reshape(N, -1) if k in ("obs", "act", "obs_next") else d[k]) for k in ("obs", "act", "obs_next", "cp_obs", "cp_act", "future_bool")}
    stats, seq = otrain.fit_feed_sequence(oenvs.make_env(c["env"]), data, ReplayIndexStream(log), c["E"], epochs_run, 16,
                                          valid_split_ratio=0.25)
    for k, v in stats.items():
        np.testing.assert_allclose(v, g("stat_" + k), rtol=0, atol=1e-12, err_msg=k)
    names = ("obs", "act", "delta", "obs_next", "back_delta", "cp_obs", "cp_act")
    assert sum(len(b) for b, _ in seq) == epochs_run * per_epoch
    for nm in names:
        got = np.concatenate([b[nm] for batches, _ in seq for b in batches], axis=1)
        np.testing.assert_array_equal(got, g("train_" + nm), err_msg="train " + nm)
        got = np.concatenate([vb[nm] for _, vb in seq], axis=1)
        np.testing.assert_array_equal(got, g("valid_" + nm), err_msg="valid " + nm)
    assert [b["obs"].shape[1] for batches, _ in seq for b in batches] == list(g("train_sizes"))
    # the early stop: the reference stopped after `epochs_run` epochs of the scripted validation losses
    last, _ = otrain.early_stop_trace(list(g("v_script")), 0.9)
    assert last + 1 == epochs_run < len(g("v_script"))


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(gi.LOSS_CASES))
def test_hip_training_forward_matches_the_reference_constructor(gpu, case):
    from cadm_amd import synth
    c, nets, inp = rebuild(case)
    vanilla = bool(c.get("vanilla"))
    has_back = len(nets["backward_model"]) > 0
    prob = synth.make_problem(env=c["env"], context=not vanilla, E=c["E"], m=1, hidden_sizes=c["hidden"],
                              cp_hidden_sizes=c["cp_hidden"] or (8,), C=c["C"], Hh=c["Hh"], H=c["H"], with_back=has_back, seed=0)
    prob["ff"] = nets["ff_model"]
    if not vanilla:
        prob["cp"] = nets["context_model"]
    if has_back:
        prob["back"] = nets["backward_model"]
    prob["stats"] = dict(prob["stats"], **{k: np.asarray(v, np.float64) for k, v in inp["stats"].items()
                                           if not vanilla or k.split("_")[0] in ("obs", "act", "delta")})
    eng = synth.make_engine(prob, p=c["p"], deterministic=c["deterministic"])
    eng.train_configure(1e-3, c["weight_decays"], c["context_weight_decays"], c["weight_decay_coeff"], c["back_coeff"],
                        max_batch=c["B"])
    keys = ("obs", "act", "delta") + (() if vanilla else ("cp_obs", "cp_act")) + (("obs_next", "back_delta") if has_back else ())
    batch = {k: eng._t(inp["bs_" + k]) for k in keys}
    got = eng.train_step(batch, train=False).cpu().numpy()                     # [mse, back_mse, recon]
    want = [float(GOLD[case + "/mse_loss"]), float(GOLD[case + "/back_mse_loss"]) if has_back else 0.0, float(GOLD[case + "/recon_loss"])]
    np.testing.assert_allclose(got, want, rtol=5e-5)
    eng.close()
