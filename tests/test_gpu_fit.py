"""`fit` end to end against the oracle on an INJECTED index stream ("fixed permutation, bootstrap_idx,
per-epoch shuffles"): the HIP model and `oracle.train.fit_reference` consume the same recorded draws, so every batch holds
the same rows and the two training runs can be compared step by step
(/root/reference/cadm/dynamics/mlp_cadm_ensemble_cem_dynamics.py:442-513)."""
import numpy as np
import pytest

from cadm_amd.dynamics.mlp_cadm_ensemble_cem_dynamics import (FitIndexStream, MLPEnsembleCEMDynamicsModel,
                                                              RecordingIndexStream, ReplayIndexStream)
from cadm_amd.envs import make_env_spec
from oracle import envs as oenvs
from oracle import train as otrain

pytestmark = pytest.mark.gpu

WD = (0.000025, 0.00005, 0.000075, 0.000075, 0.0001)
CWD = (0.000025, 0.00005, 0.000075)


def _windows(rng, N, D=18, A=6, Hh=10, F=10):
    obs = rng.standard_normal((N, F * D))
    act = rng.uniform(-1, 1, (N, F * A))
    obs_next = obs + 0.05 * rng.standard_normal((N, F * D))
    fb = np.ones((N, F))
    fb[rng.integers(0, N, N // 4), rng.integers(1, F, N // 4)] = 0        # ragged futures
    return dict(obs=obs, act=act, obs_next=obs_next, cp_obs=0.1 * rng.standard_normal((N, D * Hh)),
                cp_act=rng.uniform(-1, 1, (N, A * Hh)), future_bool=fb)


@pytest.mark.parametrize("back_coeff", [0.5, 0.0])
def test_fit_matches_oracle_on_injected_index_stream(gpu, back_coeff):
    E, B, epochs = 5, 64, 2
    model = MLPEnsembleCEMDynamicsModel("dyn", make_env_spec("halfcheetah"), hidden_nonlinearity="swish", batch_size=B,
                                        n_forwards=5, n_candidates=64, ensemble_size=E, n_particles=5, use_cem=True,
                                        weight_decays=WD, weight_decay_coeff=1.0, context_weight_decays=CWD, state_diff=1,
                                        back_coeff=back_coeff, normalize_input=True, valid_split_ratio=0.2, seed=4)
    eng = model.engine
    start = {net: {k: v.detach().cpu().numpy().astype(np.float64) for k, v in eng.nets[net].items()} for net in eng.net_names()}
    data = _windows(np.random.default_rng(2), 30)
    rec = RecordingIndexStream(FitIndexStream(np.random.default_rng(9)))
    model.fit(epochs=epochs, index_stream=rec, **data)
    got_train = np.concatenate(model.last_fit_trace["train"]).astype(np.float64)
    got_valid = np.asarray(model.last_fit_trace["valid"], np.float64)
    kinds = [k for k, _ in rec.log]
    assert kinds == ["permutation", "bootstrap"] + ["epoch_order"] * epochs      # the reference's np.random call order
    assert got_train.shape[0] >= 3 * epochs, "need >= 3 batches per epoch, got %d steps" % got_train.shape[0]

    cfg = dict(deterministic=False, back_coeff=back_coeff, weight_decay_coeff=1.0, weight_decays=WD,
               context_weight_decays=CWD, n_hidden=4, n_cp_hidden=3, state_diff=True, discrete=False)
    nets = dict(ff=start["ff_model"], back=start.get("backward_model"), cp=start["context_model"])
    ref = otrain.fit_reference(oenvs.make_env("halfcheetah"), "halfcheetah", nets, data, ReplayIndexStream(rec.log), cfg,
                               epochs=epochs, batch_size=B, valid_split_ratio=0.2)
    ref_train, ref_valid = np.asarray(ref["train"]), np.asarray(ref["valid"])
    assert ref_train.shape == got_train.shape and ref_valid.shape == got_valid.shape
    # losses per step: <= 2e-3 relative (fp32 HIP vs fp64 oracle over an Adam trajectory)
    scale = np.maximum(np.abs(ref_train), 1e-3)
    err = np.abs(got_train - ref_train) / scale
    assert err.max() <= 2e-3, "per-step losses off by %.2e at step %d" % (err.max(), int(err.max(1).argmax()))
    assert (np.abs(got_valid - ref_valid) / np.maximum(np.abs(ref_valid), 1e-3)).max() <= 2e-3
    # the statistics the model pushed to the device are the reference's (float64 host math)
    for k, v in zip(("obs_mean", "obs_std", "act_mean", "act_std", "delta_mean", "delta_std", "cp_obs_mean", "cp_obs_std",
                     "cp_act_mean", "cp_act_std", "back_delta_mean", "back_delta_std"), model.get_normalization_stats()):
        np.testing.assert_allclose(np.asarray(v, np.float64), ref["stats"][k], rtol=1e-12, atol=0)
    # weights after the run: within the Adam band (every update is <= lr per step; fp32-vs-fp64 sign flips of tiny
    # gradients move single entries by up to ~lr, the bulk agrees far tighter)
    steps = got_train.shape[0]
    for net, okey in (("ff_model", "ff"), ("context_model", "cp"), ("backward_model", "back")):
        if ref["params"].get(okey) is None or net not in eng.nets:
            continue
        for name, want in ref["params"][okey].items():
            have = eng.nets[net][name].detach().cpu().numpy().astype(np.float64)
            d = np.abs(have - want)
            assert d.max() <= 2.0 * 1e-3 * steps * 0.25 + 1e-6, "%s/%s: max |dW| %.2e" % (net, name, d.max())
            assert np.sqrt(np.mean(d ** 2)) <= 2e-5 + 2e-3 * np.sqrt(np.mean((want - start[net][name]) ** 2)), \
                "%s/%s: rms weight difference %.2e vs rms update %.2e" % (net, name, np.sqrt(np.mean(d ** 2)),
                                                                          np.sqrt(np.mean((want - start[net][name]) ** 2)))
