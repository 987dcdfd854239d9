"""Host logic of `fit` (reference cadm/dynamics/mlp_cadm_ensemble_cem_dynamics.py:382-569) on CPU: dataset growth,
train/validation split, `_preprocess_inputs`, bootstrap + shuffle_rows batches, rolling-average early stop.
The HIP engine is replaced by a recorder; the arithmetic it would do is covered by the -m gpu tests."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from cadm_amd.dynamics.mlp_cadm_ensemble_cem_dynamics import MLPEnsembleCEMDynamicsModel
from oracle import train as otrain


class _Space:
    def __init__(self, n):
        self.shape = (n,)


class _Env:
    """duck-typed env: only what fit() touches"""
    observation_space, action_space, proc_observation_space_dims = _Space(3), _Space(2), 3

    @staticmethod
    def targ_proc(obs, next_obs):
        return next_obs - obs

    @staticmethod
    def obs_preproc(obs):
        return obs


class _Recorder:
    """Stands in for HipEngine: remembers what the host loop asked for, replays scripted validation losses."""
    device = torch.device("cpu")

    def __init__(self, v_recon):
        self.v_recon, self.train_batches, self.valid_calls, self.stats, self.repacked = list(v_recon), [], 0, None, 0

    def _t(self, x, dtype=torch.float32):
        return torch.as_tensor(np.asarray(x), dtype=dtype)

    def train_configure(self, *a, **k):
        pass

    def set_stats(self, stats):
        self.stats = stats

    def train_step(self, batch, train=True):
        if train:
            self.train_batches.append({k: v.clone() for k, v in batch.items()})
            return torch.tensor([1.0, 2.0, 3.0])
        v = self.v_recon[min(self.valid_calls, len(self.v_recon) - 1)]
        self.valid_calls += 1
        return torch.tensor([0.5, 0.25, v])

    def train_step_rows(self, dev, F, row_w, row_f, idx, train=True):
        """what cadm_train_step_rows reads: per-step tensors [N, F, .] at (w, f), history tensors at w"""
        w, f = row_w[idx], row_f[idx]
        N = dev["obs"].shape[0]
        batch = {k: dev[k].view(N, F, -1)[w, f] for k in ("obs", "act", "delta", "obs_next", "back_delta")}
        batch["cp_obs"], batch["cp_act"] = dev["cp_obs"][w], dev["cp_act"][w]
        return self.train_step(batch, train)

    def repack(self):
        self.repacked += 1


def _model(E, batch_size, v_recon, Hh=2, F=3):
    m = object.__new__(MLPEnsembleCEMDynamicsModel)
    m.env, m._dataset, m.normalization = _Env(), None, None
    m.obs_space_dims, m.action_space_dims, m.proc_obs_space_dims, m.history_length, m.future_length = 3, 2, 3, Hh, F
    m.ensemble_size, m.batch_size, m.valid_split_ratio, m.rolling_average_persitency = E, batch_size, 0.2, 0.9
    m.normalize_input, m.state_diff, m.discrete, m.back_coeff = True, False, False, 0.5
    m.learning_rate, m.weight_decays, m.context_weight_decays, m.weight_decay_coeff = 1e-3, (0,) * 5, (0,) * 4, 0.0
    m.seed, m._call, m._train_ready, m._stats_dirty = 0, 0, False, True
    m.engine = _Recorder(v_recon)
    return m


def _data(N, F=3, Hh=2, seed=0):
    r = np.random.default_rng(seed)
    fb = (r.uniform(size=(N, F)) < 0.8).astype(np.float64)
    fb[:, 0] = 1.0
    return dict(obs=r.standard_normal((N, 3 * F)), act=r.standard_normal((N, 2 * F)), obs_next=r.standard_normal((N, 3 * F)),
                cp_obs=r.standard_normal((N, 3 * Hh)), cp_act=r.standard_normal((N, 2 * Hh)), future_bool=fb)


@pytest.mark.parametrize("E", [1, 4])
def test_split_preprocess_bootstrap_batches(E):
    N, B = 50, 16
    d = _data(N)
    m = _model(E, B, v_recon=[5.0] * 10)
    m.fit(epochs=2, rng=np.random.default_rng(3), **d)
    eng = m.engine
    # split (:441-443): int(N * 0.2) validation windows, the rest train; rows = windows exploded by future_bool (:676-696)
    rng = np.random.default_rng(3)
    perm = rng.permutation(N)
    n_valid = int(N * 0.2)
    tr, va = perm[n_valid:], perm[:n_valid]
    delta = (d["obs_next"].reshape(-1, 3) - d["obs"].reshape(-1, 3)).reshape(N, -1)
    want = otrain.preprocess_inputs(d["obs"][tr], d["act"][tr], delta[tr], d["cp_obs"][tr], d["cp_act"][tr], d["future_bool"][tr],
                                    d["obs_next"][tr], -delta[tr], D=3, A=2, Hh=2, F=3)
    n_train = int(d["future_bool"][tr].sum())
    assert want[0].shape[0] == n_train
    nb = int(np.ceil(n_train / B))
    assert len(eng.train_batches) == 2 * nb and eng.valid_calls == 2 and eng.repacked == 1
    # every batch is [E, <=B, .]; over one epoch member e sees exactly its bootstrap multiset (a permutation of the
    # identity for E == 1, :465-467); rows come from the preprocessed train set
    train_obs = torch.as_tensor(want[0], dtype=torch.float32)
    for ep in range(2):
        batches = eng.train_batches[ep * nb:(ep + 1) * nb]
        assert all(b["obs"].shape[0] == E and b["obs"].shape[2] == 3 for b in batches)
        assert [b["obs"].shape[1] for b in batches] == [B] * (nb - 1) + [n_train - B * (nb - 1)]
        for e in range(E):
            seen = torch.cat([b["obs"][e] for b in batches])
            assert seen.shape[0] == n_train
            # each seen row is a row of the train set
            dist = torch.cdist(seen.double(), train_obs.double()).min(1).values
            assert float(dist.max()) < 1e-6
            if E == 1:
                assert sorted(map(tuple, seen.numpy().round(5))) == sorted(map(tuple, train_obs.numpy().round(5)))
                # all seven tensors of a batch row belong to the SAME (window, future offset) row, history untiled
                names = ("obs", "act", "delta", "obs_next", "back_delta", "cp_obs", "cp_act")
                joint = np.concatenate([torch.cat([b[k][e] for b in batches]).numpy() for k in names], axis=1)
                ref = np.concatenate([np.asarray(w_, dtype=np.float32) for w_ in want], axis=1)
                assert sorted(map(tuple, joint.round(4))) == sorted(map(tuple, ref.round(4)))
    # epoch 2 reshuffles the SAME bootstrap multiset (shuffle_rows, :472-474)
    for e in range(E):
        e0 = torch.cat([b["obs"][e] for b in eng.train_batches[:nb]]).numpy().round(5)
        e1 = torch.cat([b["obs"][e] for b in eng.train_batches[nb:]]).numpy().round(5)
        assert sorted(map(tuple, e0)) == sorted(map(tuple, e1)) and not np.array_equal(e0, e1)
    # normalisation statistics were computed over the WHOLE dataset's first-step columns (:428-433) and pushed
    assert eng.stats is not None
    np.testing.assert_allclose(eng.stats["obs_mean"], d["obs"][:, :3].mean(0), rtol=1e-12)
    # the dataset grows across fit() calls (:404-426)
    m.fit(epochs=1, rng=np.random.default_rng(4), **_data(20, seed=1))
    assert m._dataset["obs"].shape[0] == N + 20


@pytest.mark.parametrize("v_recon", [
    [1.0, 0.9, 0.8, 0.7, 0.6, 0.5],                 # keeps improving: runs all epochs
    [1.0, 1.0, 1.0, 1.0, 3.0, 3.0, 3.0, 3.0],       # rises: stops once the rolling average turns up
    [-1.0, -1.2, -1.1, -0.2, -0.1, 0.5, 0.5],       # negative losses use the /1.5, /2 initialisation (:547-549)
    [2.0, 5.0, 5.0],
])
def test_early_stop_matches_reference_rule(v_recon):
    m = _model(2, 8, v_recon)
    logged = {}
    from cadm_amd.utils import log as logger
    old = logger.logkv
    logger.logkv = lambda k, v: logged.__setitem__(k, v)
    try:
        m.fit(epochs=len(v_recon), rng=np.random.default_rng(0), log_tabular=True, **_data(30))
    finally:
        logger.logkv = old
    last, _ = otrain.early_stop_trace(v_recon, 0.9)
    assert m.engine.valid_calls == last + 1
    assert logged["Epochs"] == last


def test_no_validation_split_runs_all_epochs():
    m = _model(2, 8, [9.0])
    m.fit(epochs=3, valid_split_ratio=0.0, rng=np.random.default_rng(0), **_data(12))
    assert m.engine.valid_calls == 0 and len(m.engine.train_batches) == 3 * int(np.ceil(m._dataset["future_bool"].sum() / 8))


def test_shape_asserts():
    m = _model(2, 8, [1.0])
    d = _data(10)
    d["act"] = d["act"][:, :-1]
    with pytest.raises(AssertionError):
        m.fit(epochs=1, **d)
