"""MLPEnsembleCEMDynamicsModel (Vanilla DM / PE-TS, no context) -- MI355X drop-in for
/root/reference/cadm/dynamics/mlp_ensemble_cem_dynamics.py:11.

Same constructor kwargs (:25-45) and methods: get_action(obs, cem_init_mean, cem_init_var)
(:191-207), fit(obs, act, obs_next, ...) (:209-323), save/load (:325-341).  Shares the HIP
path of the CaDM model with context_dim = 0 and no backward model.
"""
from collections import OrderedDict

import numpy as np

from .mlp_cadm_ensemble_cem_dynamics import MLPEnsembleCEMDynamicsModel as _CaDMModel


class MLPEnsembleCEMDynamicsModel(_CaDMModel):
    def __init__(self, name, env, hidden_sizes=(200, 200, 200, 200), hidden_nonlinearity="swish",
                 output_nonlinearity=None, batch_size=128, learning_rate=0.001, normalize_input=True,
                 optimizer=None, valid_split_ratio=0.2, rolling_average_persitency=0.99, n_forwards=30,
                 n_candidates=2500, ensemble_size=5, n_particles=20, use_cem=False, deterministic=False,
                 weight_decays=(0., 0., 0., 0., 0.), weight_decay_coeff=0.0,
                 reference_quirks=True, seed=0, device=None, process_group=None, check_replicated_calls=2,
                 check_replicated_every=256, engine_lib=None):
        super().__init__(name, env, hidden_sizes=hidden_sizes, hidden_nonlinearity=hidden_nonlinearity,
                         output_nonlinearity=output_nonlinearity, batch_size=batch_size, learning_rate=learning_rate,
                         normalize_input=normalize_input, optimizer=optimizer, valid_split_ratio=valid_split_ratio,
                         rolling_average_persitency=rolling_average_persitency, n_forwards=n_forwards,
                         n_candidates=n_candidates, ensemble_size=ensemble_size, n_particles=n_particles,
                         use_cem=use_cem, deterministic=deterministic, weight_decays=weight_decays,
                         weight_decay_coeff=weight_decay_coeff, cp_hidden_sizes=(), context_weight_decays=(),
                         context_out_dim=0, history_length=0, future_length=1, state_diff=False, back_coeff=0.0,
                         reference_quirks=reference_quirks, seed=seed, device=device, process_group=process_group,
                         check_replicated_calls=check_replicated_calls, check_replicated_every=check_replicated_every,
                         engine_lib=engine_lib)

    def get_action(self, obs, cem_init_mean=None, cem_init_var=None):
        return super().get_action(obs, None, None, cem_init_mean, cem_init_var)

    def predict(self, obs, act, return_std=False):
        return super().predict(obs, act, None, None, return_std=return_std)

    def get_context_pred(self, *a, **k):
        raise AttributeError("the vanilla model has no context encoder")

    def compute_normalization(self, obs, act, delta, *unused):
        """reference :343-351: the vanilla model keeps obs / delta / act statistics only.  (The extra positional
        arguments are what the shared `fit` passes for the CaDM model; a vanilla model has no history / backward net.)"""
        assert obs.shape[0] == delta.shape[0] == act.shape[0]
        proc_obs = self.env.obs_preproc(obs)
        n = OrderedDict()
        n["obs"] = (np.mean(proc_obs, axis=0), np.std(proc_obs, axis=0))
        n["delta"] = (np.mean(delta, axis=0), np.std(delta, axis=0))
        n["act"] = (np.mean(act, axis=0), np.std(act, axis=0))
        self.normalization = n
        self._stats_dirty = True

    def get_normalization_stats(self):
        """reference :353-373: (obs_mean, obs_std, act_mean, act_std, delta_mean, delta_std)."""
        return self._stats12()[:6]

    def fit(self, obs, act, obs_next, epochs=1000, compute_normalization=True, valid_split_ratio=None,
            rolling_average_persitency=None, verbose=False, log_tabular=False, max_logging=5000, rng=None, index_stream=None):
        """reference :209-323 (single-step samples, no history window)."""
        N = obs.shape[0]
        D, A = self.obs_space_dims, self.action_space_dims
        assert obs.ndim == 2 and obs.shape[1] == D
        assert obs_next.ndim == 2 and obs_next.shape[1] == D
        assert act.ndim == 2 and act.shape[1] == A
        return super().fit(obs, act, obs_next, np.zeros((N, 0)), np.zeros((N, 0)), np.ones((N, 1)), epochs=epochs,
                           compute_normalization=compute_normalization, valid_split_ratio=valid_split_ratio,
                           rolling_average_persitency=rolling_average_persitency, verbose=verbose,
                           log_tabular=log_tabular, max_logging=max_logging, rng=rng, index_stream=index_stream)
