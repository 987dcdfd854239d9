#!/usr/bin/env python3
"""Golden stream for SURVEY.md 8f-1 (the caller's per-env bookkeeping around get_action), produced by the REFERENCE ITSELF.

`Sampler.obtain_samples` (/root/reference/cadm/samplers/sampler.py:60-214) is pure numpy; it is imported unchanged and run
with a deterministic toy env (vectorised by the reference's own `IterativeEnvExecutor`) and a recording toy policy.  What
is recorded is exactly what the sampler FEEDS the policy at every step -- warm-start mean (`prev_sol`), `init_var`, the
history windows `cp_obs` / `cp_act` -- next to the observations, the plans the policy returned, the actions the sampler
took, and the dones; plus the paths it finally returns.  The packages the module chain merely imports (`pyprind`, `gym`,
...) are satisfied with placeholders; `pyprind.ProgBar` is a do-nothing bar.

Run in the build container only:   python tests/golden/make_sampler_golden.py      -> tests/golden/sampler_golden.npz
"""
import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_f2_golden as f2  # noqa: E402   (placeholder machinery)

REF = "/root/reference"
CASES = {   # name: (num_envs, max_path_length, horizon, D, A, history_length, state_diff, done schedule: env -> episode lengths)
    "diff_h3": (3, 7, 4, 5, 2, 3, True, {0: [3, 7], 1: [7], 2: [5, 2]}),
    "plain_h2": (2, 6, 3, 4, 3, 2, False, {0: [6], 1: [2, 4, 6]}),
    # half-cheetah shapes (D = 18, A = 6, history 10, horizon 10): the device kernels are replayed on this one
    "hc_shape": (3, 14, 10, 18, 6, 10, True, {0: [14], 1: [12, 3], 2: [4, 11]}),
}


class ToyEnv:
    """Deterministic dynamics; `done` after a scheduled number of steps of the current episode (per env copy)."""

    def __init__(self, D, A):
        self.D, self.A = D, A
        self.action_space = types.SimpleNamespace(shape=(A,))
        self.k, self.schedule, self.episode, self.t = 0, [], 0, 0

    def reset(self):
        self.episode += 1
        self.t = 0
        self.obs = np.sin(np.arange(self.D) * 0.7 + self.k + 0.31 * self.episode)
        return self.obs.copy()

    def step(self, a):
        self.t += 1
        self.obs = 0.9 * self.obs + 0.1 * np.cos(np.arange(self.D) + a.sum()) + 0.05 * self.k
        limit = self.schedule[min(self.episode - 1, len(self.schedule) - 1)]
        return self.obs.copy(), float(self.obs[0]), self.t >= limit, {}


def run_case(name):
    n_env, L, H, D, A, Hh, state_diff, sched = CASES[name]
    for mod in ("tensorflow", "pyprind", "gym", "gym.spaces", "mujoco_py", "baselines", "tensorboardX", "mpi4py"):
        if mod not in sys.modules:
            try:
                importlib.import_module(mod)
            except Exception:
                sys.modules[mod] = f2._Anything(mod)
    bar = types.ModuleType("pyprind")
    bar.ProgBar = type("ProgBar", (), {"__init__": lambda self, *a, **k: None, "update": lambda self, *a, **k: None,
                                       "stop": lambda self: None})
    sys.modules["pyprind"] = bar
    if REF not in sys.path:
        sys.path.insert(0, REF)
    S = importlib.import_module("cadm.samplers.sampler")            # the reference sampler, unchanged
    rec = {k: [] for k in ("obses", "init_mean", "init_var", "cp_obs", "cp_act", "plans")}

    class Policy:
        def get_actions(self, obses, init_mean=None, init_var=None, cp_obs=None, cp_act=None):
            for k, v in (("obses", obses), ("init_mean", init_mean), ("init_var", init_var), ("cp_obs", cp_obs), ("cp_act", cp_act)):
                rec[k].append(np.array(v, np.float64))
            t = len(rec["plans"])
            base = np.tanh(np.asarray(obses)[:, :1, None] + 0.3 * np.arange(H)[None, :, None] + 0.7 * np.arange(A)[None, None, :] + 0.11 * t)
            plan = np.clip(0.5 * base + 0.5 * init_mean, -1, 1)
            rec["plans"].append(plan.copy())
            return plan, []

    env = ToyEnv(D, A)
    smp = S.Sampler(env, Policy(), num_rollouts=n_env, max_path_length=L, n_parallel=1, use_cem=True, horizon=H, context=True,
                    state_diff=state_diff, history_length=Hh)
    for k, e in enumerate(smp.vec_env.envs):                        # (deep copies of `env`: give each its identity and schedule)
        e.k, e.schedule = k, sched[k]
    # the dones / next observations are not handed to the policy: capture them at the executor (the reference's own object)
    steps = {k: [] for k in ("actions", "next_obses", "dones")}
    inner = smp.vec_env.step

    def step(actions):
        out = inner(actions)
        steps["actions"].append(np.array(actions, np.float64))
        steps["next_obses"].append(np.array(out[0], np.float64))
        steps["dones"].append(np.array(out[2], bool))
        return out

    smp.vec_env.step = step
    paths = smp.obtain_samples()
    res = {name + "/meta": np.array([n_env, L, H, D, A, Hh, int(state_diff)], np.int64)}
    for k, v in {**rec, **steps}.items():
        res[name + "/" + k] = np.stack(v)
    res[name + "/path_lengths"] = np.array([len(p["rewards"]) for p in paths], np.int64)
    for k in ("observations", "actions", "cp_obs", "cp_act"):
        res[name + "/paths_" + k] = np.concatenate([p[k] for p in paths], axis=0)
    return res


def main():
    out = {}
    for name in CASES:
        out.update(run_case(name))
    np.savez_compressed(os.path.join(HERE, "sampler_golden.npz"), **out)
    print({k: v.shape for k, v in out.items() if k.startswith("diff_h3")})


if __name__ == "__main__":
    main()
