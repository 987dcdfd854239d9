"""CPU stand-in for cadm_amd.engine.HipEngine built on the oracle: same primitive interface, torch CPU
tensors.  Lets the host-side orchestration (cadm_amd/planner.py: candidate sharding + the per-iteration
all-gather) run under gloo without a GPU.  Test infrastructure only."""
import numpy as np
import torch

from oracle import nets as onets
from oracle import philox as ophilox
from oracle import planner as oplanner


class OracleEngine:
    def __init__(self, o, prob, p, H, deterministic=False, num_cem_iters=5, num_elites=50, dtype=np.float32):
        self.o, self.prob, self.p, self.H = o, prob, p, H
        self.E, self.D, self.A, self.C = prob["E"], prob["D"], prob["A"], prob["C"]
        self.discrete = prob["discrete"]
        self.deterministic = deterministic
        self.num_cem_iters, self.num_elites = num_cem_iters, num_elites
        self.dt = dtype
        self.device = torch.device("cpu")

    def _t(self, x, dtype=torch.float32):
        return torch.as_tensor(np.asarray(x), dtype=dtype)

    def _n(self, t):
        return np.asarray(t, dtype=self.dt)

    def context_forward(self, cp_obs, cp_act):
        return self._t(onets.context_forward(self.o["cp"], self._n(cp_obs), self._n(cp_act), self.o["st"]))

    def sample_actions(self, mean, var, n_global, z=None, seed=0, call=0, it=0, cand_offset=0, n_local=None):
        m = mean.shape[0]
        zz = self._n(z) if z is not None else ophilox.truncated_normals(seed, call, it, m, n_global, self.H, self.A)
        a = oplanner.sample_actions(self._n(mean), self._n(var), zz)
        if n_local is not None and (cand_offset, n_local) != (0, n_global):
            # a rank of a sharded planner draws only its own shard: everything else in the buffer is poison (nothing may read it)
            keep = a[:, cand_offset:cand_offset + n_local].copy()
            a = np.full_like(a, np.nan)
            a[:, cand_offset:cand_offset + n_local] = keep
        return self._t(a)

    def sample_uniform(self, m, n_global, seed=0, call=0):
        return self._t(ophilox.rs_uniforms(seed, call, m, n_global, self.H, self.A)), None

    def rollout_returns(self, obs, ctx_vec, actions, eps=None, norm_actions=True, seed=0, call=0, it=0, cand_offset=0,
                        n_local=None, **_):
        acts = self._n(actions)
        m, n_global = acts.shape[:2]
        n_local = n_global - cand_offset if n_local is None else n_local
        if eps is None:
            e = ophilox.eps_normals(seed, call, it, m, n_global, self.p, self.H, self.D, cand_offset, cand_offset + n_local)
        else:
            e = self._n(eps)
        T = None if ctx_vec is None else oplanner.context_table_indexed(self._n(ctx_vec), it)
        rows = oplanner.rollout_indexed(self.o["env"], self.o["ff"], self.o["st"], self._n(obs), T,
                                        acts[:, cand_offset:cand_offset + n_local], e.astype(self.dt), self.E, self.p,
                                        self.deterministic, norm_actions=norm_actions)
        return self._t(rows)

    def particle_mean(self, rows):
        return self._t(oplanner.particle_mean(self._n(rows)))

    def _ungather(self, cand, G):
        c = self._n(cand)                       # [G, m, n_local] -> [m, G * n_local]
        return np.concatenate([c[g] for g in range(G)], axis=1)

    def cem_refit(self, cand, actions, mean, var, G=1, want_elites=False, regen=None):
        if regen is not None:      # the elites' sequences drawn again by global candidate id from the distribution this iteration was sampled from
            seed, call, it = regen
            m, n_global = actions.shape[:2]
            zz = ophilox.truncated_normals(seed, call, it, m, n_global, self.H, self.A)
            actions = self._t(oplanner.sample_actions(self._n(mean), self._n(var), zz))
        nm, nv, idx = oplanner.elite_refit(self._n(mean), self._n(var), self._n(actions), self._ungather(cand, G), self.num_elites)
        mean.copy_(self._t(nm))
        var.copy_(self._t(nv))
        return self._t(idx, torch.int32) if want_elites else None

    def rs_select(self, cand, actions, G=1):
        c = self._ungather(cand, G)
        best = np.argmax(c, axis=1)
        a = self._n(actions)
        return self._t(a[np.arange(a.shape[0]), best, 0]), self._t(best, torch.int32)
