"""Batched context inference: `get_context_pred` on thousands of history rows per call -- the PPO consumer
(/root/reference/cadm/model_free/ppo_cadm.py:155-162, dynamics.py:369-380) -- runs the GEMM-shaped `context_batched_kernel`
(csrc/context.hip: 16/32 rows of one member per workgroup, weights reused across rows, fp32 MFMA).  Parity against the oracle
at PPO scale (m = 2048), at ragged row counts around the tile and dispatch boundaries, for layer widths that are not multiples
of 4 / 16 / 64, for both input layouts, and agreement with the planner's per-row kernel on the same rows."""
import numpy as np
import pytest

from cadm_amd import synth
from cadm_amd.dynamics.mlp_cadm_ensemble_cem_dynamics import MLPEnsembleCEMDynamicsModel
from cadm_amd.envs import make_env_spec
from helpers import assert_close, make_engine, oracle_problem
from oracle import nets as onets

pytestmark = pytest.mark.gpu


def _histories(prob, m, seed):
    rng = np.random.default_rng(seed)
    return 0.1 * rng.standard_normal((m, prob["D"] * prob["Hh"])), rng.uniform(-1, 1, (m, prob["A"] * prob["Hh"]))


@pytest.mark.parametrize("m", [2048, 48, 49, 63, 65, 1000])
def test_batched_context_matches_oracle_and_the_per_row_kernel(gpu, m):
    prob = synth.make_problem(env="halfcheetah", trained_like=True, seed=3)
    eng = make_engine(prob, p=5)
    cp_obs, cp_act = _histories(prob, m, 100 + m)
    got = eng.context_forward(cp_obs, cp_act).cpu().numpy()                     # m >= 48 rows per member: the batched kernel
    o = oracle_problem(prob, np.float32)
    ref = onets.context_forward(o["cp"], cp_obs.astype(np.float32), cp_act.astype(np.float32), o["st"])
    assert got.shape == (prob["E"], m, prob["C"])
    # fp32 MFMA = an fmaf chain over k: fp32 roundoff against numpy's float32 matmul (pure relative 1e-5 on |ref| >= 0.25 rms)
    d = np.abs(got - ref)
    big = np.abs(ref) >= 0.25 * np.sqrt(np.mean(ref ** 2))
    assert (d[big] / np.abs(ref[big])).max() <= 1e-5 and d.max() <= 1e-5 * np.abs(ref).max()
    # the same rows through the planner's per-row kernel (calls of < 48 rows), chunk by chunk: same numbers up to summation order
    step = 40
    per_row = np.concatenate([eng.context_forward(cp_obs[i:i + step], cp_act[i:i + step]).cpu().numpy() for i in range(0, min(m, 200), step)], axis=1)
    assert_close(got[:, :per_row.shape[1]], per_row, 2e-6, "batched vs per-row kernel")
    # training-graph layout: inputs already [E, m, .] with DIFFERENT rows per member
    rng = np.random.default_rng(7)
    bo = 0.1 * rng.standard_normal((prob["E"], m, prob["D"] * prob["Hh"]))
    ba = rng.uniform(-1, 1, (prob["E"], m, prob["A"] * prob["Hh"]))
    got_bs = eng.context_forward(bo, ba, bs=True).cpu().numpy()
    ref_bs = onets.context_forward_bs(o["cp"], bo.astype(np.float32), ba.astype(np.float32), o["st"])
    assert_close(got_bs, ref_bs, 1e-5, "batched context encoder, [E,m,.] inputs")
    eng.close()


def test_context_across_the_kernel_dispatch_threshold(gpu):
    """cadm_context_forward switches kernels at 48 histories per member (per-row fmaf chain below, fp32-MFMA GEMM chain from
    there on; csrc/context.hip: CADM_CONTEXT_BATCHED_MIN_ROWS).  The two sum in different orders, so the context vector of the SAME
    history is not bitwise the same in a 47-row and a 48-row call -- documented (numeric envelope) and pinned here: the
    first 47 rows of both calls agree to 2e-6 of the tensor's scale, each call is bitwise reproducible, and within one kernel a row's
    result does not depend on how many other rows the call carries (40 vs 47 rows: bit-equal)."""
    prob = synth.make_problem(env="halfcheetah", trained_like=True, seed=11)
    eng = make_engine(prob, p=5)
    cp_obs, cp_act = _histories(prob, 48, 4711)
    a47 = eng.context_forward(cp_obs[:47], cp_act[:47]).cpu().numpy()
    a48 = eng.context_forward(cp_obs, cp_act).cpu().numpy()
    assert_close(a48[:, :47], a47, 2e-6, "context of the same 47 histories: batched (m = 48) vs per-row (m = 47) kernel")
    np.testing.assert_array_equal(eng.context_forward(cp_obs[:47], cp_act[:47]).cpu().numpy(), a47)
    np.testing.assert_array_equal(eng.context_forward(cp_obs, cp_act).cpu().numpy(), a48)
    np.testing.assert_array_equal(eng.context_forward(cp_obs[:40], cp_act[:40]).cpu().numpy(), a47[:, :40])
    eng.close()


@pytest.mark.parametrize("env,E,Hh,cp_sizes,C,m", [
    ("pendulum", 5, 1, (8, 6), 3, 100),                # input width 4, layers narrower than one 64-unit group, odd widths
    ("halfcheetah", 3, 3, (320, 100, 30), 10, 77),     # a layer wider than 256; widths not multiples of 4 / 64 (scalar weight loads)
    ("ant", 7, 2, (64,), 7, 513),                      # one hidden layer, odd output width, E = 7
    ("slim_humanoid", 2, 10, (256, 128, 64), 10, 300),  # input width 620: the widest LDS input tile of the reference envs
    ("halfcheetah", 5, 10, (1024, 512), 16, 64),       # too wide for two row tiles in LDS: the one-tile flavour
])
def test_batched_context_shapes(gpu, env, E, Hh, cp_sizes, C, m):
    prob = synth.make_problem(env=env, E=E, trained_like=True, seed=5, Hh=Hh, cp_hidden_sizes=cp_sizes, C=C)
    eng = make_engine(prob, p=E)
    cp_obs, cp_act = _histories(prob, m, 9)
    got = eng.context_forward(cp_obs, cp_act).cpu().numpy()
    o = oracle_problem(prob, np.float32)
    ref = onets.context_forward(o["cp"], cp_obs.astype(np.float32), cp_act.astype(np.float32), o["st"])
    assert got.shape == (E, m, C)
    assert_close(got, ref, 1e-5, "batched context encoder vs fp32 oracle")
    eng.close()


def test_get_context_pred_at_ppo_scale(gpu):
    """Through the drop-in class (dynamics.py:369-380): 2048 histories -> [E, 2048, C], against the fp64 oracle."""
    prob = synth.make_problem(env="halfcheetah", trained_like=True, seed=11)
    model = MLPEnsembleCEMDynamicsModel("dyn_model", make_env_spec("halfcheetah"), hidden_nonlinearity="swish", n_forwards=5, n_candidates=64,
                                        ensemble_size=5, n_particles=5, use_cem=True, state_diff=1, normalize_input=False)
    model.engine.set_net("context_model", prob["cp"])
    model.engine.set_stats(prob["stats"])
    cp_obs, cp_act = _histories(prob, 2048, 12)
    got = model.engine.context_forward(cp_obs, cp_act).cpu().numpy()
    o = oracle_problem(prob, np.float64)
    ref = onets.context_forward(o["cp"], cp_obs, cp_act, o["st"])
    assert got.shape == (5, 2048, 10) and got.dtype == np.float32
    d = np.abs(got - ref)
    big = np.abs(ref) >= 0.25 * np.sqrt(np.mean(ref ** 2))
    assert (d[big] / np.abs(ref[big])).max() <= 1e-5, (d[big] / np.abs(ref[big])).max()
    out = model.get_context_pred(cp_obs, cp_act)          # the class's own call (its statistics: identity with normalize_input=False)
    assert out.shape == (5, 2048, 10) and np.isfinite(out).all()
