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
