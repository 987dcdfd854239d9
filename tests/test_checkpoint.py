"""checkpoint interop.  The fixture is built from the ORACLE's parameter dicts in the reference's
`tf.trainable_variables()` order (/root/reference/cadm/dynamics/mlp_cadm_ensemble_cem_dynamics.py:140-266,571-577) --
independently of the engine's own save path."""
import os
import subprocess
import sys
from collections import OrderedDict

import joblib
import numpy as np
import pytest

from cadm_amd import checkpoint as ck
from cadm_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_layout_list(prob):
    """What `sess.run(self.params)` returns: context_model, ff_model, backward_model, creation order inside each."""
    out = []
    for net in (prob["cp"], prob["ff"], prob.get("back")):
        if net is not None:
            out += [np.asarray(v, np.float32) for v in net.values()]
    return out


@pytest.mark.parametrize("context,back", [(True, True), (True, False), (False, False)])
def test_describe_names_every_entry(context, back):
    prob = synth.make_problem(env="halfcheetah", context=context, E=5, with_back=back, seed=2)
    arrays = reference_layout_list(prob)
    info = ck.describe(arrays)
    assert info["E"] == 5 and info["hidden_sizes"] == (200,) * 4 and info["obs_dim"] == 18 and info["back_model"] == back
    assert info["context_dim"] == (10 if context else 0) and info["input_dim"] == 18 + 6 + (10 if context else 0)
    named = ck.to_named(arrays)
    want = []
    for net, d in (("context_model", prob["cp"]), ("ff_model", prob["ff"]), ("backward_model", prob.get("back"))):
        if d is not None:
            want += ["%s/%s" % (net, k) for k in d]
    assert list(named) == want
    back_again = ck.from_named(named)
    assert len(back_again) == len(arrays) and all(np.array_equal(a, b) for a, b in zip(arrays, back_again))
    with pytest.raises(ValueError):
        ck.describe(arrays[:-1])
    with pytest.raises(ValueError):
        ck.describe(arrays[1:])


def test_cli_round_trip(tmp_path):
    prob = synth.make_problem(env="slim_humanoid", context=True, E=3, with_back=True, seed=5)
    arrays = reference_layout_list(prob)
    src = str(tmp_path / "params_epoch_3")
    joblib.dump(arrays, src)
    stats = OrderedDict((k, (np.arange(3.0) + i, np.ones(3) + i)) for i, k in enumerate(("obs", "delta", "act", "cp_obs", "cp_act", "back_delta")))
    joblib.dump(stats, src + "_norm_stats")
    tool = [sys.executable, os.path.join(ROOT, "tools", "ckpt_convert.py")]
    assert "context_model/cp_hidden_0_weight" in subprocess.run(tool + ["inspect", src], capture_output=True, text=True, check=True).stdout
    subprocess.run(tool + ["to-npz", src, str(tmp_path / "named.npz")], check=True, capture_output=True)
    z = np.load(str(tmp_path / "named.npz"))
    assert "ff_model/output_logvar_weight" in z.files and "norm_stats/back_delta_std" in z.files
    subprocess.run(tool + ["from-npz", str(tmp_path / "named.npz"), str(tmp_path / "again")], check=True, capture_output=True)
    again = joblib.load(str(tmp_path / "again"))
    assert all(np.array_equal(a, b) for a, b in zip(arrays, again)) and len(again) == len(arrays)
    s2 = joblib.load(str(tmp_path / "again_norm_stats"))
    assert list(s2) == list(stats) and all(np.array_equal(s2[k][1], stats[k][1]) for k in stats)


@pytest.mark.gpu
def test_reference_layout_checkpoint_loads_and_plans_like_the_oracle(gpu, tmp_path):
    """A checkpoint file in the reference's layout -> model.load -> injected-noise CEM plan equals the oracle's plan
    computed from the SAME parameter dicts."""
    from cadm_amd import planner as hplanner
    from cadm_amd.dynamics.mlp_cadm_ensemble_cem_dynamics import MLPEnsembleCEMDynamicsModel
    from cadm_amd.envs import make_env_spec
    from helpers import oracle_problem, trunc_z
    from oracle import planner as oplanner
    E, p, n, H, m = 5, 10, 64, 8, 2
    prob = synth.make_problem(env="halfcheetah", context=True, E=E, m=m, H=H, with_back=True, trained_like=True, seed=21)
    path = str(tmp_path / "params_epoch_7")
    joblib.dump(reference_layout_list(prob), path)
    st = prob["stats"]
    joblib.dump(OrderedDict(obs=(st["obs_mean"], st["obs_std"]), delta=(st["delta_mean"], st["delta_std"]),
                            act=(st["act_mean"], st["act_std"]), cp_obs=(st["cp_obs_mean"], st["cp_obs_std"]),
                            cp_act=(st["cp_act_mean"], st["cp_act_std"]), back_delta=(st["back_delta_mean"], st["back_delta_std"])),
                path + "_norm_stats")
    model = MLPEnsembleCEMDynamicsModel("dyn", make_env_spec("halfcheetah"), hidden_nonlinearity="swish", n_forwards=H,
                                        n_candidates=n, ensemble_size=E, n_particles=p, use_cem=True, state_diff=1,
                                        back_coeff=0.5, normalize_input=True)
    model.load(path)
    model._push_stats()
    rng = np.random.default_rng(3)
    z = trunc_z(rng, (5, m, n, H, 6)).astype(np.float32)
    eps = rng.standard_normal((5, H, m, n, p, 18)).astype(np.float32)
    eng = model.engine
    plan = hplanner.cem_plan(eng, prob["obs"], prob["cp_obs"], prob["cp_act"], prob["init_mean"], prob["init_var"], n,
                             z=eng._t(z), eps=eng._t(eps)).cpu().numpy()
    o = oracle_problem(prob, np.float32)
    ref = oplanner.cem_plan(o["env"], o["ff"], o["cp"], o["st"], o["obs"], o["cp_obs"], o["cp_act"], o["init_mean"], o["init_var"],
                            z, eps, E, p)
    ref = oplanner.get_action_clip(ref)
    assert np.abs(plan - ref).max() <= 2e-4, np.abs(plan - ref).max()


# ---- a checkpoint written by the REFERENCE's own save() (dynamics.py:571-577), see tests/golden/make_loss_golden.py ----
REF_CKPT = os.path.join(ROOT, "tests", "golden", "ref_ckpt", "params_epoch_7")


def _ref_ckpt_truth():
    """The variables the reference constructor created for that checkpoint, regenerated from the shared input module."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import graph_inputs as gi
    gold = np.load(os.path.join(ROOT, "tests", "golden", "loss_golden.npz"))
    c = gi.LOSS_CASES["ckpt_e1"]
    W = gi.Weights(c["seed"])
    out = []
    for name, shp in zip(gold["ckpt_e1/var_names"], gold["ckpt_e1/var_shapes"]):
        parts, shape = str(name).split("/"), tuple(int(s) for s in str(shp).split(","))
        pname = parts[3]
        arr = W.dense(str(name), shape) if pname.endswith(("_weight", "_bias")) else \
            W.plain(str(name), np.ones(shape) / 2.0 if pname == "max_logvar" else -np.ones(shape) * 10)
        out.append(("%s/%s" % (parts[1], pname), arr))
    return c, gold, out


def test_reference_written_checkpoint_is_named_correctly():
    c, gold, truth = _ref_ckpt_truth()
    arrays = joblib.load(REF_CKPT)
    info = ck.describe(arrays)
    assert info["E"] == 1 and info["hidden_sizes"] == c["hidden"] and info["context_dim"] == c["C"] and info["back_model"]
    named = ck.to_named(arrays)
    assert list(named) == [k for k, _ in truth]
    for k, v in truth:
        np.testing.assert_array_equal(named[k], v, err_msg=k)
    stats = joblib.load(REF_CKPT + "_norm_stats")
    assert list(stats) == ["obs", "delta", "act", "cp_obs", "cp_act", "back_delta"]
    np.testing.assert_allclose(stats["obs"][0], gold["ckpt_e1/norm_obs_mean"], rtol=0, atol=1e-12)


@pytest.mark.gpu
def test_reference_written_checkpoint_loads_into_the_model(gpu):
    from cadm_amd.dynamics.mlp_cadm_ensemble_cem_dynamics import MLPEnsembleCEMDynamicsModel
    from cadm_amd.envs import make_env_spec
    c, gold, truth = _ref_ckpt_truth()
    model = MLPEnsembleCEMDynamicsModel("dyn", make_env_spec("halfcheetah"), hidden_sizes=c["hidden"], hidden_nonlinearity="swish",
                                        n_forwards=c["H"], n_candidates=64, ensemble_size=c["E"], n_particles=c["p"], use_cem=True,
                                        state_diff=1, back_coeff=c["back_coeff"], normalize_input=True, cp_hidden_sizes=c["cp_hidden"],
                                        context_out_dim=c["C"], history_length=c["Hh"], future_length=c["F"])
    model.load(REF_CKPT)
    eng = model.engine
    for key, want in truth:
        net, pname = key.split("/")
        np.testing.assert_array_equal(eng.nets[net][pname].cpu().numpy(), want, err_msg=key)
    got = model.get_normalization_stats()
    for g, k in zip(got, ("obs_mean", "obs_std", "act_mean", "act_std", "delta_mean", "delta_std", "cp_obs_mean", "cp_obs_std",
                          "cp_act_mean", "cp_act_std", "back_delta_mean", "back_delta_std")):
        np.testing.assert_allclose(np.asarray(g, np.float64), gold["ckpt_e1/norm_" + k], rtol=0, atol=1e-7, err_msg=k)
    plan = model.get_action(np.zeros((1, 18)), np.zeros((1, 18 * c["Hh"])), np.zeros((1, 6 * c["Hh"])), np.zeros((1, c["H"], 6)),
                            np.full((1, c["H"], 6), 0.25))
    assert plan.shape == (1, c["H"], 6) and np.isfinite(plan).all()
