"""The rollout kernel exists in several flavours -- the cooperative kernel with one or two 16-row tiles per workgroup
(rollout_xdl.h: XC<.., MT>) and the wave-tile kernel for large batches (rollout_wt.h: every wave its own tile, 8 or 4 per
workgroup); the launcher cuts a batch between them by size.  A row's arithmetic is the same in all of them, so they must agree
BIT FOR BIT on any input; this forces each flavour (developer library: cadm_dev_set_rollout, 1 / 2 = cooperative, 3 / 4 =
wave-tile) on ragged problem sizes: row counts that are not a multiple of 16, odd tile counts, single-tile members, more tiles
than one round of workgroups holds, every noise mode."""
import numpy as np
import pytest
import torch

from cadm_amd import _lib, synth
from helpers import make_engine

pytestmark = pytest.mark.gpu


def _run(eng, prob, ctx, acts, eps, flavour, **kw):
    eng.dev_set_rollout("xdl", row_tiles=int(flavour))
    try:
        rows, traj = eng.rollout_returns(prob["obs"], ctx, acts, eps=eps, want_traj=True, **kw)
        torch.cuda.synchronize()
        return rows.cpu().numpy(), traj.cpu().numpy()
    finally:
        eng.dev_set_rollout("xdl", row_tiles=0)


@pytest.mark.parametrize("env,context,E,p,n,m", [
    ("halfcheetah", True, 5, 20, 7, 1),        # 28 rows per member: 2 tiles, the second ragged
    ("halfcheetah", True, 5, 20, 13, 2),       # 104 rows per member: 7 tiles (odd)
    ("halfcheetah", False, 5, 5, 3, 1),        # 3 rows per member: one partial tile
    ("slim_humanoid", True, 5, 20, 9, 1),      # two pair slots of state per thread
    ("pendulum", True, 5, 5, 50, 3),
    ("cartpole", True, 5, 5, 33, 1),
    ("ant", True, 5, 20, 11, 1),               # four pair slots per lane in the wave-tile kernel, two layer-0 chunks
    ("halfcheetah", True, 1, 4, 2100, 4),      # 33600 rows of one member: 2100 tiles > one 8-tile round of 256 workgroups
])
def test_one_and_two_row_tiles_per_workgroup_agree_bitwise(gpu, env, context, E, p, n, m):
    H = 12
    prob = synth.make_problem(env=env, context=context, E=E, m=m, H=H, trained_like=True, seed=21)
    eng = make_engine(prob, p=p, lib=_lib.load_dev())
    rng = np.random.default_rng(4)
    acts = eng._t(rng.uniform(-1, 1, (m, n, H, prob["A"])).astype(np.float32))
    ctx = eng.context_forward(prob["cp_obs"], prob["cp_act"]) if context else None
    eps = torch.randn((H, m, n, p, prob["D"]), device=eng.device)
    obs_rows = rng.standard_normal((m, n, p, prob["D"])).astype(np.float32)      # per-row initial observations (the teacher-forced form)
    for kw, e in (({}, eps), ({"seed": 5, "call": 3, "it": 1}, None), ({"obs_rows": obs_rows, "seed": 9, "call": 1, "it": 0}, None)):
        # injected noise; device Philox (odd iteration: Q2); per-row initial observations
        r1, t1 = _run(eng, prob, ctx, acts, e, "1", **kw)
        assert np.isfinite(r1).all()
        for flavour in ("2", "3", "4"):
            r2, t2 = _run(eng, prob, ctx, acts, e, flavour, **kw)
            np.testing.assert_array_equal(r2, r1, err_msg="flavour " + flavour)
            np.testing.assert_array_equal(t2, t1, err_msg="flavour " + flavour)
    eng.close()


def test_launcher_cut_between_flavours_changes_nothing(gpu):
    """Sizes at which the launcher's plan mixes flavours (wave-tile rounds + a cooperative remainder): same bits as one flavour."""
    H = 6
    for n in (260, 700, 1300):      # 20 rows per candidate and 5 members: 65 / 175 / 325 tiles per member
        prob = synth.make_problem(env="halfcheetah", context=True, E=5, m=1, H=H, trained_like=True, seed=n)
        eng = make_engine(prob, p=20, lib=_lib.load_dev())
        acts = eng._t(np.random.default_rng(n).uniform(-1, 1, (1, n, H, prob["A"])).astype(np.float32))
        ctx = eng.context_forward(prob["cp_obs"], prob["cp_act"])
        kw = {"seed": 11, "call": 2, "it": 0}
        r0, t0 = _run(eng, prob, ctx, acts, None, "0", **kw)      # the launcher's own plan
        r1, t1 = _run(eng, prob, ctx, acts, None, "1", **kw)
        np.testing.assert_array_equal(r0, r1)
        np.testing.assert_array_equal(t0, t1)
        eng.close()


def test_deterministic_model_both_flavours(gpu):
    prob = synth.make_problem(env="halfcheetah", context=False, E=1, m=1, H=8, trained_like=True, seed=2)
    eng = make_engine(prob, p=1, deterministic=True, lib=_lib.load_dev())
    acts = eng._t(np.random.default_rng(0).uniform(-1, 1, (1, 45, 8, prob["A"])).astype(np.float32))
    r1, t1 = _run(eng, prob, None, acts, None, "1")
    for flavour in ("2", "3", "4"):
        r2, t2 = _run(eng, prob, None, acts, None, flavour)
        np.testing.assert_array_equal(r2, r1, err_msg="flavour " + flavour)
        np.testing.assert_array_equal(t2, t1, err_msg="flavour " + flavour)
    eng.close()


def test_one_hidden_layer_all_flavours_agree(gpu):
    """One hidden layer (round 5: the reference accepts any hidden_sizes tuple, dynamics.py:28): layer 0 feeds the heads directly -- no
    hidden-to-hidden sweep, no resident hidden fragments, and the side jobs that ride in the hidden-layer-1 sweep (the next step's action
    features) need another home.  A JIT-built geometry over SEVERAL steps: all four flavours bit-identical, and the trajectory within the
    multi-step band of the fp32 oracle."""
    from helpers import assert_close, oracle_problem
    from oracle import nets as onets
    from oracle import planner as oplanner
    H, m, n, p = 5, 1, 37, 20
    prob = synth.make_problem(env="halfcheetah", context=True, E=5, m=m, H=H, hidden_sizes=(200,), trained_like=True, seed=23)
    eng = make_engine(prob, p=p, lib=_lib.load_dev())
    rng = np.random.default_rng(8)
    acts_np = rng.uniform(-1, 1, (m, n, H, prob["A"])).astype(np.float32)
    acts = eng._t(acts_np)
    ctx = eng.context_forward(prob["cp_obs"], prob["cp_act"])
    eps = torch.randn((H, m, n, p, prob["D"]), device=eng.device)
    for kw, e in (({}, eps), ({"seed": 5, "call": 3, "it": 1}, None)):
        r1, t1 = _run(eng, prob, ctx, acts, e, "1", **kw)
        assert np.isfinite(r1).all()
        for flavour in ("2", "3", "4"):
            r2, t2 = _run(eng, prob, ctx, acts, e, flavour, **kw)
            np.testing.assert_array_equal(r2, r1, err_msg="flavour " + flavour)
            np.testing.assert_array_equal(t2, t1, err_msg="flavour " + flavour)
        if e is not None:
            o = oracle_problem(prob, np.float32)
            T = oplanner.context_table_indexed(onets.context_forward(o["cp"], o["cp_obs"], o["cp_act"], o["st"]), 0)
            r_ref, t_ref = oplanner.rollout_indexed(o["env"], o["ff"], o["st"], o["obs"], T, acts_np, e.cpu().numpy(), 5, p, False, return_traj=True)
            assert_close(t1, t_ref, 5e-5, "5-step trajectory of a one-hidden-layer net vs the fp32 oracle")
            assert_close(r1, r_ref, 5e-5, "returns")
    eng.close()
