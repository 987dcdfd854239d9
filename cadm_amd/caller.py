"""Device-resident caller state for the planner.

The reference's samplers keep, per vectorised env, the CEM warm start `prev_sol` / `init_var`
(/root/reference/cadm/samplers/sampler.py:50-57,118-120) and the history window that becomes
`cp_obs` / `cp_act` (:95-97,165-178,193-202; evaluation twin samplers/utils.py:70-109) in numpy and
feed them to `policy.get_actions` every step.  `DevicePlannerState` keeps the same state in HBM and
drives the HIP planner directly, so a batched simulator can step without host round trips for this
bookkeeping: `act(obs)` = one planner call + warm-start shift, `observe(...)` = ring-buffer update.
"""
import numpy as np
import torch

from ._lib import ptr


class DevicePlannerState:
    def __init__(self, model, num_envs):
        self.model = model
        eng = self.eng = model.engine
        self.m = int(num_envs)
        H, A, D, Hh = eng.H, eng.A, eng.D, eng.Hh
        dev = eng.device
        self.prev_sol = torch.zeros((self.m, H, A), dtype=torch.float32, device=dev)          # sampler.py:52
        self.init_var = torch.full((self.m, H, A), 0.25, dtype=torch.float32, device=dev)     # sampler.py:53 (2^2 / 16)
        self.hist_obs = torch.zeros((self.m, D * max(Hh, 1)), dtype=torch.float32, device=dev)
        self.hist_act = torch.zeros((self.m, A * max(Hh, 1)), dtype=torch.float32, device=dev)
        self.counts = torch.zeros((self.m,), dtype=torch.int32, device=dev)
        self.action = torch.zeros((self.m, A), dtype=torch.float32, device=dev)
        self.context = eng.C > 0

    def act(self, obs):
        """One MPC step: CEM plan from the warm start, then shift the plan (sampler.py:109-120).
        obs [m,D] (numpy or device tensor) -> device tensor [m,A] (the action to execute)."""
        model, eng = self.model, self.eng
        model._push_stats()
        obs = eng._t(obs)
        plan = eng.cem_plan(obs, self.hist_obs if self.context else None, self.hist_act if self.context else None,
                            self.prev_sol, self.init_var, model.n_candidates, seed=model.seed, call=model._next_call())
        eng._check(eng.lib.cadm_warm_start_shift(eng._ctx, ptr(plan), self.m, ptr(self.prev_sol), ptr(self.action), eng.stream),
              "cadm_warm_start_shift")
        return self.action

    def observe(self, obs, action, next_obs, done=None):
        """History update after the env step (sampler.py:165-178) and per-env reset on done (:193-200)."""
        eng = self.eng
        if not self.context:
            return
        obs, action, next_obs = eng._t(obs), eng._t(action), eng._t(next_obs)
        d = None if done is None else eng._t(np.asarray(done, dtype=np.int32) if not isinstance(done, torch.Tensor) else done,
                                             dtype=torch.int32)
        eng._check(eng.lib.cadm_history_update(eng._ctx, ptr(obs), ptr(next_obs), ptr(action), ptr(d), self.m,
                                          int(bool(self.model.state_diff)), ptr(self.counts), ptr(self.hist_obs),
                                          ptr(self.hist_act), ptr(self.prev_sol), eng.stream), "cadm_history_update")

    def reset(self):
        for t in (self.prev_sol, self.hist_obs, self.hist_act, self.counts):
            t.zero_()
