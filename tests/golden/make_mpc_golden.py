#!/usr/bin/env python3
"""Golden table for the MPC controller's dispatch, produced by the REFERENCE ITSELF.

`MPCController` (/root/reference/cadm/policies/mpc_controller.py:6-90) is imported unchanged and driven through its entry
points for every (context, use_cem) combination with a recording dynamics model; the arguments it hands to
`dynamics_model.get_action` are identified by sentinel values (0 = observations, 1 = cp_obs, 2 = cp_act, 3 = init_mean,
4 = init_var, -1 = None).   Run in the build container only:  python tests/golden/make_mpc_golden.py -> mpc_golden.npz"""
import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_f2_golden as f2  # noqa: E402   (placeholder machinery)

REF = "/root/reference"


def sentinel(i, shape):
    return np.full(shape, float(i))


def ident(x):
    return -1 if x is None else int(np.asarray(x).flat[0])


def main():
    for mod in ("tensorflow", "pyprind", "gym", "gym.spaces", "mujoco_py", "baselines", "tensorboardX", "mpi4py"):
        if mod not in sys.modules:
            try:
                importlib.import_module(mod)
            except Exception:
                sys.modules[mod] = f2._Anything(mod)
    sys.path.insert(0, REF)
    M = importlib.import_module("cadm.policies.mpc_controller")         # the reference controller, unchanged
    out = {}
    for context in (False, True):
        for use_cem in (False, True):
            calls = []

            class Model:
                def get_action(self, *args):
                    calls.append([ident(a) for a in args])
                    return np.zeros((2, 3))

            env = types.SimpleNamespace(reward=lambda *a: 0.0, observation_space=None, action_space=None)
            ctl = M.MPCController("policy", env, Model(), use_cem=use_cem, context=context)
            obs, cpo, cpa, mean, var = (sentinel(i, (2, 3)) for i in range(5))
            ctl.get_actions(obs, cp_obs=cpo, cp_act=cpa, init_mean=mean, init_var=var)
            ctl.get_action(sentinel(0, (3,)), init_mean=mean, init_var=var)
            key = "ctx%d_cem%d" % (context, use_cem)
            out[key + "/get_actions"] = np.array(calls[0], np.int64)
            out[key + "/get_action"] = np.array(calls[1], np.int64)
            out[key + "/get_action_obs_ndim"] = np.int64(2)             # (a 1-D observation is promoted to [1, D], :44-45)
    np.savez_compressed(os.path.join(HERE, "mpc_golden.npz"), **out)
    print({k: v.tolist() for k, v in out.items() if v.ndim})


if __name__ == "__main__":
    main()
