"""Consumer of a REAL TensorFlow-1.15 pin (tools/tf_pin_export.py): when a file written by that script on a machine with the reference's
stack is present at tests/golden/tf_pin.npz (or $CADM_TF_PIN), the oracle (CPU) and the HIP kernels (-m gpu) are held to TensorFlow's own
numbers -- `get_context_pred`, the training graph's loss scalars on a fixed batch, and the parameters after 1 and 3 `train_op` steps (TF's
autodiff + AdamOptimizer).  Without the file the TF-pinned tests SKIP (the image that builds cadm_amd has no TensorFlow; DESIGN.md section 2
says what the committed goldens do and do not prove); the consumer itself is exercised on a file of the same layout written by the oracle
(`test_consumer_on_an_oracle_written_file`), so a pin dropped in later meets code that has run.

Reference: cadm/dynamics/mlp_cadm_ensemble_cem_dynamics.py:269-317 (losses, train_op), :369-380 (get_context_pred), :571-588 (save / load)."""
import json
import os
from collections import OrderedDict

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PIN = os.environ.get("CADM_TF_PIN") or os.path.join(HERE, "golden", "tf_pin.npz")
STAT_KEYS = ("obs", "act", "delta", "cp_obs", "cp_act", "back_delta")
BATCH_KEYS = ("obs", "act", "delta", "obs_next", "back_delta", "cp_obs", "cp_act")
LOSS_MAP = (("mse", "mse_loss"), ("back_mse", "back_mse_loss"), ("recon", "recon_loss"), ("loss", "loss"))


class Pin:
    def __init__(self, path):
        z = np.load(path, allow_pickle=False)
        self.z = z
        self.cfg = json.loads(str(z["config_json"]))
        self.dataset = str(z["dataset"])
        self.E, self.B, self.D, self.A, self.Hh = (int(z[k]) for k in ("E", "B", "D", "A", "Hh"))
        n = int(z["n_params"])
        self.params = [z["param_%03d" % i] for i in range(n)]
        self.after = {s: [z["after%d_param_%03d" % (s, i)] for i in range(n)] for s in (1, 3) if "after%d_param_000" % s in z}
        self.stats = OrderedDict((k + sfx, np.asarray(z["stat_" + k + sfx], np.float64)) for k in STAT_KEYS for sfx in ("_mean", "_std"))
        self.batch = {k: z["batch_" + k] for k in BATCH_KEYS}
        self.losses = {k[5:]: float(z[k]) for k in z.files if k.startswith("loss_")}

    def train_cfg(self):
        c = self.cfg
        return dict(deterministic=bool(c["deterministic"]), back_coeff=float(c["back_coeff"]), weight_decay_coeff=float(c["weight_decay_coeff"]),
                    weight_decays=tuple(c["weight_decays"]), context_weight_decays=tuple(c["context_weight_decays"]),
                    n_hidden=len(c["hidden_sizes"]), n_cp_hidden=len(c["context_hidden_sizes"]), hidden_nonlinearity=c["hidden_nonlinearity"])


def _split(params):
    from cadm_amd import checkpoint
    info = checkpoint.describe(params)
    out, pos = {"context_model": None, "ff_model": None, "backward_model": None}, 0
    for net, names in info["nets"]:
        out[net] = OrderedDict((nm, np.asarray(params[pos + i])) for i, nm in enumerate(names))
        pos += len(names)
    return out


def oracle_quantities(pin, dtype="float32", steps=(1, 3)):
    """everything the exporter wrote, recomputed by the oracle from the pin's parameters / statistics / batch"""
    import torch
    from oracle import nets as onets
    from oracle import train as otrain
    tdt = getattr(torch, dtype)
    nets = _split(pin.params)
    cfg = pin.train_cfg()
    t = lambda d: otrain.to_torch(d, tdt) if d is not None else None
    st = t(pin.stats)
    batch = {k: torch.tensor(np.asarray(v), dtype=tdt) for k, v in pin.batch.items()}
    out = {}
    cp_np = onets.cast_params(nets["context_model"], np.dtype(dtype).type)
    st_np = onets.cast_stats(pin.stats, np.dtype(dtype).type)
    ctx = onets.context_forward(cp_np, pin.z["ctx_cp_obs"].astype(dtype), pin.z["ctx_cp_act"].astype(dtype), st_np)      # [E, m, C]
    out["ctx_pred"] = np.asarray(ctx)
    L = otrain.train_losses(pin.dataset, t(nets["ff_model"]), t(nets["backward_model"]), t(nets["context_model"]), st, batch, cfg)
    out["losses"] = {k: float(L[k]) for k, _ in LOSS_MAP}
    # Adam steps with the oracle's TF1 Adam on autograd gradients
    live = {k: (otrain.to_torch(v, tdt, requires_grad=True) if v is not None else None) for k, v in nets.items()}
    lr, b1, b2, eps = (float(x) for x in pin.z["adam"])
    opt = otrain.TF1Adam(lr=lr, b1=b1, b2=b2, eps=eps)
    out["after"] = {}
    for s in range(1, max(steps) + 1):
        loss = otrain.train_losses(pin.dataset, live["ff_model"], live["backward_model"], live["context_model"], st, batch, cfg)["loss"]
        opt.step(live, otrain.grads_of(loss, live))
        if s in steps:
            out["after"][s] = [p.detach().numpy().copy() for net in ("context_model", "ff_model", "backward_model") if live[net] is not None
                               for p in live[net].values()]
    return out


def compare(pin, got, what, loss_rtol=1e-5, elem_rtol=1e-5):
    """the north_star bar: 1e-5 relative -- scalars as they are, tensors against max(|ref|, rms(ref)) elementwise"""
    from helpers import assert_close
    assert_close(got["ctx_pred"], pin.z["ctx_pred"], elem_rtol, what + ": get_context_pred")
    for ok, gk in LOSS_MAP:
        if gk in pin.losses:
            np.testing.assert_allclose(got["losses"][ok], pin.losses[gk], rtol=loss_rtol, err_msg="%s: %s" % (what, gk))
    for s, plist in got.get("after", {}).items():
        if s not in pin.after:
            continue
        names = [str(x) for x in pin.z["param_names"]]
        lr = float(pin.z["adam"][0])
        for nm, a, b, p0 in zip(names, plist, pin.after[s], pin.params):
            # parameters at the 1e-5 bar.  Adam's first steps move EVERY weight by ~lr whatever its gradient's size (m / sqrt(v) = +-1): an
            # element whose gradient is within roundoff of zero is ill-conditioned -- its step can land anywhere in [-lr, lr] -- so at most
            # 0.5 % of a tensor may miss the bar, each by no more than 2 % of the step size.
            a64, b64 = np.asarray(a, np.float64), np.asarray(b, np.float64)
            scale = max(float(np.sqrt(np.mean(b64 * b64))), 1e-30)
            miss = np.abs(a64 - b64) > elem_rtol * np.maximum(np.abs(b64), scale)
            assert miss.sum() <= max(1, int(0.005 * miss.size)), "%s: %s after %d steps: %d/%d elements beyond 1e-5" % (what, nm, s, miss.sum(), miss.size)
            assert np.abs(a64 - b64).max() <= 2e-2 * lr * s, "%s: %s after %d steps: worst |diff| %.3e" % (what, nm, s, np.abs(a64 - b64).max())


needs_pin = pytest.mark.skipif(not os.path.exists(PIN), reason="no TensorFlow pin (tools/tf_pin_export.py writes one on a machine with TF 1.15 + "
                               "the reference; drop it at tests/golden/tf_pin.npz or set $CADM_TF_PIN)")


@needs_pin
def test_oracle_matches_tensorflow():
    pin = Pin(PIN)
    assert str(pin.z["tf_version"]).startswith("1.15"), "not a TensorFlow-1.15 export"
    compare(pin, oracle_quantities(pin), "oracle vs TF %s" % pin.z["tf_version"])


def hip_quantities(pin, steps=(1, 3)):
    import torch
    from cadm_amd.dynamics.mlp_cadm_ensemble_cem_dynamics import MLPEnsembleCEMDynamicsModel
    from cadm_amd.envs import make_env_spec
    c = pin.cfg
    model = MLPEnsembleCEMDynamicsModel(
        "dyn_model", make_env_spec(pin.dataset), hidden_sizes=tuple(c["hidden_sizes"]), hidden_nonlinearity=c["hidden_nonlinearity"],
        batch_size=pin.B, learning_rate=float(c["learning_rate"]), normalize_input=True, n_forwards=int(c["horizon"]),
        n_candidates=int(c["n_candidates"]), ensemble_size=pin.E, n_particles=int(c["n_particles"]), use_cem=True,
        deterministic=bool(c["deterministic"]), weight_decays=tuple(c["weight_decays"]), weight_decay_coeff=float(c["weight_decay_coeff"]),
        cp_hidden_sizes=tuple(c["context_hidden_sizes"]), context_weight_decays=tuple(c["context_weight_decays"]),
        context_out_dim=int(c["context_out_dim"]), history_length=pin.Hh, future_length=int(c["future_length"]), state_diff=bool(c["state_diff"]),
        back_coeff=float(c["back_coeff"]))
    model.engine.load_params_list(pin.params)
    eng = model.engine
    eng.set_stats(pin.stats)
    out = {"ctx_pred": eng.context_forward(pin.z["ctx_cp_obs"], pin.z["ctx_cp_act"]).cpu().numpy()}
    model._ensure_train()
    batch = {k: eng._t(v) for k, v in pin.batch.items()}
    l = eng.train_step(batch, train=False).cpu().numpy()
    out["losses"] = {"mse": float(l[0]), "back_mse": float(l[1]), "recon": float(l[2]), "loss": float("nan")}
    out["after"] = {}
    for s in range(1, max(steps) + 1):
        eng.train_step(batch, train=True)
        if s in steps:
            torch.cuda.synchronize()
            out["after"][s] = [np.asarray(a) for a in eng.params_list()]
    return out


@needs_pin
@pytest.mark.gpu
def test_hip_matches_tensorflow(gpu):
    pin = Pin(PIN)
    got = hip_quantities(pin)
    got["losses"].pop("loss")                 # (the device step reports mse / back_mse / recon; the regularised total is the oracle's to check)
    pin.losses.pop("loss", None)
    compare(pin, got, "HIP vs TF %s" % pin.z["tf_version"])


def _write_oracle_pin(path, dataset="halfcheetah", E=3, B=16, seed=5):
    """a file of the exporter's LAYOUT whose numbers come from the fp64 oracle -- NOT a TensorFlow pin (tf_version says so)"""
    import torch
    from cadm_amd import synth
    from cadm_amd.engine import ctx_param_names, dyn_param_names
    Hh = 10
    prob = synth.make_problem(env=dataset, context=True, E=E, m=3, hidden_sizes=(200,) * 4, Hh=Hh, trained_like=True, with_back=True, seed=seed)
    D, A, P = prob["D"], prob["A"], prob["P"]
    params, names = [], []
    for net, key, pn in (("context_model", "cp", ctx_param_names(3)), ("ff_model", "ff", dyn_param_names(4)), ("backward_model", "back", dyn_param_names(4))):
        for nm in pn:
            params.append(np.asarray(prob[key][nm], np.float32))
            names.append("dyn_model/%s/x/%s:0" % (net, nm))
    cfg = dict(hidden_sizes=[200] * 4, hidden_nonlinearity="swish", deterministic=False, weight_decays=[0.000025, 0.00005, 0.000075, 0.000075, 0.0001],
               weight_decay_coeff=1.0, context_hidden_sizes=[256, 128, 64], context_weight_decays=[0.000025, 0.00005, 0.000075], context_out_dim=10,
               back_coeff=0.5, learning_rate=0.001, horizon=30, n_candidates=200, n_particles=E * 2, future_length=10, state_diff=1)
    out = {"config_json": np.array(json.dumps(cfg)), "tf_version": np.array("oracle-fp64 (NOT TensorFlow)"), "dataset": np.array(dataset),
           "D": D, "A": A, "P": P, "E": E, "B": B, "Hh": Hh, "n_params": len(params), "param_names": np.array(names), "adam": np.array([0.001, 0.9, 0.999, 1e-8])}
    for i, a in enumerate(params):
        out["param_%03d" % i] = a
    st = prob["stats"]
    for k in STAT_KEYS:
        out["stat_%s_mean" % k] = np.asarray(st[k + "_mean"], np.float32)
        out["stat_%s_std" % k] = np.asarray(st[k + "_std"], np.float32)
    for k, v in synth.make_train_batch(prob, B=B, seed=seed + 1).items():
        out["batch_" + k] = np.asarray(v, np.float32)
    out["ctx_cp_obs"], out["ctx_cp_act"] = np.asarray(prob["cp_obs"], np.float32), np.asarray(prob["cp_act"], np.float32)
    np.savez(path, **out, ctx_pred=np.zeros(1), loss_mse_loss=0.0)
    pin = Pin(path)
    q = oracle_quantities(pin, "float64")
    out["ctx_pred"] = q["ctx_pred"]
    for ok, gk in LOSS_MAP:
        out["loss_" + gk] = np.float64(q["losses"][ok])
    for s, plist in q["after"].items():
        for i, a in enumerate(plist):
            out["after%d_param_%03d" % (s, i)] = a
    np.savez(path, **out)


def test_consumer_on_an_oracle_written_file(tmp_path):
    """The loader, the architecture inference from a bare variable list, the oracle's loss / context / Adam replay and the comparison code
    all run -- on a file in the exporter's layout written by the fp64 oracle, against which the fp32 oracle must hold the 1e-5 bar."""
    path = str(tmp_path / "oracle_pin.npz")
    _write_oracle_pin(path)
    pin = Pin(path)
    assert not str(pin.z["tf_version"]).startswith("1.15")
    compare(pin, oracle_quantities(pin, "float32"), "fp32 oracle vs fp64 oracle")


@pytest.mark.gpu
def test_hip_consumer_on_an_oracle_written_file(gpu, tmp_path):
    path = str(tmp_path / "oracle_pin.npz")
    _write_oracle_pin(path, E=5, B=32)
    pin = Pin(path)
    got = hip_quantities(pin)
    got["losses"].pop("loss")
    pin.losses.pop("loss", None)
    compare(pin, got, "HIP vs fp64 oracle")
