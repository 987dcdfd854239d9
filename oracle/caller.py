"""Oracle restatement of the samplers' per-env bookkeeping around get_action (test infrastructure only).

Follows /root/reference/cadm/samplers/sampler.py: warm start :52-57,118-120; history window :95-97,165-178;
done handling :193-202.  Pinned to the reference's own Sampler.obtain_samples run (tests/golden/make_sampler_golden.py ->
sampler_golden.npz, tests/test_sampler_golden.py)."""
import numpy as np


class SamplerState:
    def __init__(self, num_envs, horizon, obs_dim, act_dim, history_length, state_diff):
        self.H, self.D, self.A, self.Hh, self.state_diff = horizon, obs_dim, act_dim, history_length, state_diff
        self.prev_sol = np.tile(0., [num_envs, horizon, act_dim])                      # :52
        self.init_var = np.tile(np.square(2) / 16, [num_envs, horizon, act_dim])       # :53
        self.state_counts = [0] * num_envs                                             # :92
        self.history_state = np.zeros((num_envs, obs_dim * history_length))            # :96
        self.history_act = np.zeros((num_envs, act_dim * history_length))              # :97

    def after_plan(self, cem_solutions):                                               # :118-120
        self.prev_sol[:, :-1] = cem_solutions[:, 1:].copy()
        self.prev_sol[:, -1:] = 0.
        return cem_solutions[:, 0].copy()

    def after_step(self, obses, actions, next_obses, dones):                           # :141-202
        D, A, Hh = self.D, self.A, self.Hh
        for idx in range(len(obses)):
            observation, action = obses[idx], actions[idx]
            if self.state_counts[idx] < Hh:
                c = self.state_counts[idx]
                self.history_state[idx][c * D:(c + 1) * D] = (next_obses[idx] - observation) if self.state_diff else observation
                self.history_act[idx][c * A:(c + 1) * A] = action
            else:
                self.history_state[idx][:-D] = self.history_state[idx][D:]
                self.history_state[idx][-D:] = (next_obses[idx] - observation) if self.state_diff else observation
                self.history_act[idx][:-A] = self.history_act[idx][A:]
                self.history_act[idx][-A:] = action
            if dones[idx]:
                self.prev_sol[idx] = 0.                                                 # reset_cem :55-57
                self.state_counts[idx] = 0
                self.history_state[idx] = np.zeros((D * Hh))
                self.history_act[idx] = np.zeros((A * Hh))
            else:
                self.state_counts[idx] += 1
