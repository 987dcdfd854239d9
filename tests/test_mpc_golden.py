"""`MPCController`'s dispatch pinned to the reference's own class: tests/golden/mpc_golden.npz records which arguments
/root/reference/cadm/policies/mpc_controller.py hands to `dynamics_model.get_action` from every entry point and every
(context, use_cem) combination (tests/golden/make_mpc_golden.py); the drop-in controller must hand over the same ones."""
import os

import numpy as np
import pytest

from cadm_amd.policies.mpc_controller import MPCController

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mpc_golden.npz"))


@pytest.mark.parametrize("context", [False, True])
@pytest.mark.parametrize("use_cem", [False, True])
def test_controller_dispatch_matches_the_reference(context, use_cem):
    calls = []

    class Model:
        def get_action(self, *args):
            calls.append([-1 if a is None else int(np.asarray(a).flat[0]) for a in args] + [np.asarray(args[0]).ndim])
            return np.zeros((2, 3))

    ctl = MPCController("policy", object(), Model(), use_cem=use_cem, context=context)
    obs, cpo, cpa, mean, var = (np.full((2, 3), float(i)) for i in range(5))
    out, info = ctl.get_actions(obs, cp_obs=cpo, cp_act=cpa, init_mean=mean, init_var=var)
    assert out.shape == (2, 3) and info == {}
    ctl.get_action(np.full((3,), 0.0), init_mean=mean, init_var=var)
    key = "ctx%d_cem%d" % (context, use_cem)
    assert calls[0][:-1] == GOLD[key + "/get_actions"].tolist()
    assert calls[1][:-1] == GOLD[key + "/get_action"].tolist()
    assert calls[1][-1] == int(GOLD[key + "/get_action_obs_ndim"])       # a 1-D observation is promoted to [1, D]
