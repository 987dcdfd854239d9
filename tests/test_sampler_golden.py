"""The caller's per-env bookkeeping around get_action pinned to the reference's own `Sampler`:
tests/golden/sampler_golden.npz records what /root/reference/cadm/samplers/sampler.py's `obtain_samples`, run unchanged on a
toy env and a recording policy (tests/golden/make_sampler_golden.py), FED the policy at every step -- warm-start mean,
init_var, history windows -- together with the plans, actions, next observations and dones of that run.  The oracle's
restatement (oracle/caller.py), the host-side `CEMWarmStart` and the device kernels must reproduce that stream."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "sampler_golden.npz"))
CASES = sorted({k.split("/")[0] for k in GOLD.files})


def stream(case):
    g = lambda k: GOLD[case + "/" + k]
    n_env, L, H, D, A, Hh, sd = (int(x) for x in g("meta"))
    return dict(n_env=n_env, L=L, H=H, D=D, A=A, Hh=Hh, state_diff=bool(sd), T=g("obses").shape[0], **{
        k: g(k) for k in ("obses", "init_mean", "init_var", "cp_obs", "cp_act", "plans", "actions", "next_obses", "dones")})


@pytest.mark.parametrize("case", CASES)
def test_oracle_caller_reproduces_the_reference_sampler(case):
    from cadm_amd.policies.mpc_controller import CEMWarmStart
    from oracle.caller import SamplerState
    s = stream(case)
    ref = SamplerState(s["n_env"], s["H"], s["D"], s["A"], s["Hh"], s["state_diff"])
    warm = CEMWarmStart(s["n_env"], s["H"], s["A"])
    assert s["dones"].any() and (~s["dones"]).any() and s["T"] > s["Hh"] + 1          # episode ends and the shift path both occur
    for t in range(s["T"]):
        np.testing.assert_array_equal(ref.prev_sol, s["init_mean"][t])
        np.testing.assert_array_equal(ref.init_var, s["init_var"][t])
        np.testing.assert_array_equal(ref.history_state, s["cp_obs"][t])
        np.testing.assert_array_equal(ref.history_act, s["cp_act"][t])
        np.testing.assert_array_equal(warm.prev_sol, s["init_mean"][t])
        np.testing.assert_array_equal(warm.init_var, s["init_var"][t])
        act = ref.after_plan(s["plans"][t])
        np.testing.assert_array_equal(act, s["actions"][t])
        np.testing.assert_array_equal(warm.step(s["plans"][t]), s["actions"][t])
        ref.after_step(s["obses"][t], act, s["next_obses"][t], s["dones"][t])
        for i in np.flatnonzero(s["dones"][t]):
            warm.reset(i)
        if t + 1 < s["T"]:       # the executor hands the NEXT policy call the reset observation of a finished env
            np.testing.assert_array_equal(s["obses"][t + 1], s["next_obses"][t])


@pytest.mark.gpu
def test_device_caller_kernels_reproduce_the_reference_sampler(gpu):
    """cadm_warm_start_shift / cadm_history_update driven with the reference run's plans and transitions (half-cheetah shapes)."""
    import torch
    from cadm_amd.caller import DevicePlannerState
    from cadm_amd._lib import check, ptr
    from test_gpu_model import CaDMModel, _cadm_kwargs
    s = stream("hc_shape")
    model = CaDMModel(**_cadm_kwargs(normalize_input=False, n_candidates=64, n_forwards=s["H"], history_length=s["Hh"],
                                    state_diff=int(s["state_diff"])))
    dev = DevicePlannerState(model, s["n_env"])
    eng = dev.eng
    assert (eng.D, eng.A, eng.H, eng.Hh) == (s["D"], s["A"], s["H"], s["Hh"])
    f32 = lambda x: np.asarray(x, np.float32)
    for t in range(s["T"]):
        np.testing.assert_array_equal(dev.prev_sol.cpu().numpy(), f32(s["init_mean"][t]))
        np.testing.assert_array_equal(dev.init_var.cpu().numpy(), f32(s["init_var"][t]))
        # (state differences: the device subtracts float32 observations, the reference float64 ones)
        np.testing.assert_allclose(dev.hist_obs.cpu().numpy(), f32(s["cp_obs"][t]), rtol=0, atol=5e-7)
        np.testing.assert_array_equal(dev.hist_act.cpu().numpy(), f32(s["cp_act"][t]))
        plan = eng._t(f32(s["plans"][t]))
        check(eng.lib.cadm_warm_start_shift(eng._ctx, ptr(plan), dev.m, ptr(dev.prev_sol), ptr(dev.action), eng.stream),
              "cadm_warm_start_shift")
        np.testing.assert_array_equal(dev.action.cpu().numpy(), f32(s["actions"][t]))
        dev.observe(f32(s["obses"][t]), f32(s["actions"][t]), f32(s["next_obses"][t]), s["dones"][t])
