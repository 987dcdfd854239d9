"""MPCController -- the thin dispatcher between the samplers and the dynamics model
(/root/reference/cadm/policies/mpc_controller.py:6-90), plus the CEM warm-start step the
samplers apply around it (/root/reference/cadm/samplers/sampler.py:52-53,118-120)."""
import numpy as np


class MPCController(object):
    def __init__(self, name, env, dynamics_model, reward_model=None, discount=1, use_cem=False,
                 n_candidates=1024, horizon=10, num_rollouts=10, context=False):
        self.name = name
        self.env = env
        self.dynamics_model = dynamics_model
        self.reward_model = reward_model
        self.discount = discount
        self.n_candidates = n_candidates
        self.horizon = horizon
        self.use_cem = use_cem
        self.context = context
        self.num_rollouts = num_rollouts

    @property
    def vectorized(self):
        return True

    def _plan(self, observations, cp=None, init=None):
        """The one dispatch point: history (cp_obs, cp_act) goes to the model only for a context model, the CEM warm
        start (mean, var) only when planning with CEM; random shooting takes neither."""
        args = [observations]
        if self.context:
            args += list(cp if cp is not None else (None, None))
        if self.use_cem:
            args += list(init if init is not None else (None, None))
        return self.dynamics_model.get_action(*args)

    # ---- the reference's entry points (mpc_controller.py:43-90), all thin views of _plan ----
    def get_action(self, observation, init_mean=None, init_var=None):
        observation = observation[None] if observation.ndim == 1 else observation
        return self._plan(observation, None, (init_mean, init_var)), dict()

    def get_actions(self, observations, cp_obs=None, cp_act=None, init_mean=None, init_var=None):
        return self._plan(observations, (cp_obs, cp_act), (init_mean, init_var)), dict()

    def get_rs_gpu_action(self, observations, cp_obs=None, cp_act=None):
        assert not self.use_cem
        return self._plan(observations, (cp_obs, cp_act))

    def get_cem_gpu_action(self, observations, init_mean, init_var, cp_obs=None, cp_act=None):
        assert self.use_cem
        return self._plan(observations, (cp_obs, cp_act), (init_mean, init_var))


class CEMWarmStart(object):
    """prev_sol / init_var bookkeeping of the samplers (sampler.py:50-57,118-120; samplers/utils.py:70-78)."""

    def __init__(self, num_rollouts, horizon, act_dim):
        self.prev_sol = np.tile(0., [num_rollouts, horizon, act_dim])
        self.init_var = np.tile(np.square(2) / 16, [num_rollouts, horizon, act_dim])   # = 0.25

    def reset(self, idx):                      # sampler.py:55-57
        self.prev_sol[idx] = 0.

    def step(self, cem_solutions):             # sampler.py:118-120
        """Shift the plan left by one step, zero the tail, return the action to execute."""
        self.prev_sol[:, :-1] = cem_solutions[:, 1:].copy()
        self.prev_sol[:, -1:] = 0.
        return cem_solutions[:, 0].copy()
