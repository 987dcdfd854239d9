"""GPU parity of the fused training step (forward, losses, hand-written backward GEMMs, TF1 Adam)
against the oracle (torch autograd in fp64 / fp32 + the TF1 Adam closed form)."""
import numpy as np
import pytest
import torch

from cadm_amd import synth
from helpers import assert_close, make_engine, rel_err
from oracle import train as otrain

pytestmark = pytest.mark.gpu

WD = (0.000025, 0.00005, 0.000075, 0.000075, 0.0001)      # run_cadm_pets.py:223
CWD = (0.000025, 0.00005, 0.000075)                        # run_cadm_pets.py:232


def _cfg(prob, det, back_coeff):
    return dict(deterministic=det, back_coeff=back_coeff, weight_decay_coeff=1.0, weight_decays=WD,
                context_weight_decays=CWD, n_hidden=len(prob["hidden_sizes"]), n_cp_hidden=len(prob["cp_hidden_sizes"]))


def _oracle_nets(prob, dtype, requires_grad=True):
    ff = otrain.to_torch(prob["ff"], dtype, requires_grad)
    back = otrain.to_torch(prob["back"], dtype, requires_grad) if prob.get("back") is not None else None
    cp = otrain.to_torch(prob["cp"], dtype, requires_grad) if prob["cp"] is not None else None
    st = otrain.to_torch(prob["stats"], dtype)
    return ff, back, cp, st


def _dev_batch(eng, batch, context, with_back):
    keys = ["obs", "act", "delta"] + (["obs_next", "back_delta"] if with_back else []) + (["cp_obs", "cp_act"] if context else [])
    return {k: eng._t(batch[k]) for k in keys}


CASES = [  # env, context, with_back, det, E, B
    ("halfcheetah", True, True, False, 5, 256),
    ("halfcheetah", True, True, False, 3, 37),       # ragged batch (tile tails)
    ("halfcheetah", False, False, True, 1, 32),      # vanilla deterministic (cfg1 / run_pets default batch)
    ("halfcheetah", False, False, False, 5, 64),     # vanilla PE-TS
    ("slim_humanoid", True, True, False, 5, 96),
    ("ant", True, False, False, 2, 50),              # CaDM without backward model (back_coeff = 0)
]


@pytest.mark.parametrize("env,context,with_back,det,E,B", CASES)
def test_losses_match_oracle(gpu, env, context, with_back, det, E, B):
    prob = synth.make_problem(env=env, context=context, E=E, trained_like=True, with_back=with_back, seed=21)
    eng = make_engine(prob, p=E, deterministic=det)
    bc = 0.5 if with_back else 0.0
    eng.train_configure(1e-3, WD, CWD, 1.0, bc, max_batch=B)
    batch = synth.make_train_batch(prob, B=B, seed=2)
    got = eng.train_step(_dev_batch(eng, batch, context, with_back), train=False).cpu().numpy()
    for dt, tol in ((torch.float32, 2e-5), (torch.float64, 5e-5)):
        ff, back, cp, st = _oracle_nets(prob, dt, False)
        tb = {k: torch.tensor(v, dtype=dt) for k, v in batch.items()}
        ref = otrain.train_losses(env, ff, back, cp, st, tb, _cfg(prob, det, bc))
        want = np.array([float(ref["mse"]), float(ref["back_mse"]), float(ref["recon"])])
        np.testing.assert_allclose(got, want, rtol=tol, atol=tol, err_msg="losses vs %s oracle" % dt)


@pytest.mark.parametrize("env,context,with_back,det,E,B", CASES)
def test_gradients_via_linearised_adam(gpu, env, context, with_back, det, E, B):
    """With beta1 = beta2 = 0, lr = eps = 1e6 the TF1 Adam update is  w -= g / (|g|/1e6 + 1) ~= g, so
    one fused step exposes every gradient of the hand-written backward pass as (w_before - w_after).
    Compared against torch.autograd on the fp64 oracle."""
    prob = synth.make_problem(env=env, context=context, E=E, trained_like=True, with_back=with_back, seed=22)
    eng = make_engine(prob, p=E, deterministic=det)
    bc = 0.5 if with_back else 0.0
    eng.train_configure(1e6, WD, CWD, 1.0, bc, max_batch=B, beta1=0.0, beta2=0.0, epsilon=1e6)
    batch = synth.make_train_batch(prob, B=B, seed=3)
    before = {n: {k: v.clone() for k, v in eng.nets[n].items()} for n in eng.net_names()}
    eng.train_step(_dev_batch(eng, batch, context, with_back), train=True)
    ff, back, cp, st = _oracle_nets(prob, torch.float64)
    tb = {k: torch.tensor(v, dtype=torch.float64) for k, v in batch.items()}
    out = otrain.train_losses(env, ff, back, cp, st, tb, _cfg(prob, det, bc))
    grads = otrain.grads_of(out["loss"], {"ff_model": ff, "backward_model": back, "context_model": cp})
    checked = 0
    for net in eng.net_names():
        for name, w0 in before[net].items():
            g_ref = grads[net][name]
            g_hip = (w0 - eng.nets[net][name]).cpu().numpy().astype(np.float64)
            if g_ref is None:      # TF skips variables without a gradient: they must not move
                assert np.abs(g_hip).max() == 0.0, "%s/%s moved although it has no gradient" % (net, name)
                continue
            g_ref = g_ref.numpy()
            scale = max(np.abs(g_ref).max(), 1e-12)
            # fp32 forward/backward + recovering g from a difference of O(0.1) weights: 2e-3 of the tensor's scale
            err = np.abs(g_hip - g_ref).max() / scale
            assert err < 2e-3, "%s/%s gradient off: rel-to-max err %.3e (max |g| %.3e)" % (net, name, err, scale)
            checked += 1
    assert checked >= 10


def test_adam_trajectory_matches_tf1_semantics(gpu):
    """10 real Adam steps on a fixed batch: losses and parameters track the fp32 oracle
    (torch autograd + TF1 Adam closed form)."""
    env, E, B = "halfcheetah", 5, 128
    prob = synth.make_problem(env=env, context=True, E=E, trained_like=True, with_back=True, seed=30)
    eng = make_engine(prob, p=E)
    eng.train_configure(1e-3, WD, CWD, 1.0, 0.5, max_batch=B)
    batch = synth.make_train_batch(prob, B=B, seed=4)
    dev = _dev_batch(eng, batch, True, True)
    ff, back, cp, st = _oracle_nets(prob, torch.float64)
    tb = {k: torch.tensor(v, dtype=torch.float64) for k, v in batch.items()}
    nets = {"ff_model": ff, "backward_model": back, "context_model": cp}
    opt = otrain.TF1Adam(1e-3)
    hip_losses, ref_losses = [], []
    for step in range(10):
        hip_losses.append(eng.train_step(dev, train=True).cpu().numpy())
        out = otrain.train_losses(env, ff, back, cp, st, tb, _cfg(prob, False, 0.5))
        ref_losses.append([float(out["mse"]), float(out["back_mse"]), float(out["recon"])])
        opt.step(nets, otrain.grads_of(out["loss"], nets))
    hip_losses, ref_losses = np.array(hip_losses), np.array(ref_losses)
    assert ref_losses[-1, 2] < ref_losses[0, 2], "oracle loss did not decrease"
    np.testing.assert_allclose(hip_losses, ref_losses, rtol=2e-3, atol=2e-3)
    for net in eng.net_names():
        for name, w in eng.nets[net].items():
            ref = nets[net][name].detach().numpy()
            moved = np.abs(ref - prob[{"ff_model": "ff", "backward_model": "back", "context_model": "cp"}[net]][name]).max()
            if moved == 0:
                continue
            # Adam's first steps are +-lr per element; a sign flip of a ~0 gradient costs at most ~2 lr per step
            assert np.abs(w.cpu().numpy() - ref).max() <= 0.15 * moved + 2e-4, "%s/%s diverged from TF1 Adam" % (net, name)


def test_train_then_plan_uses_updated_weights(gpu):
    """After training steps the planner must see the new weights (streams re-packed)."""
    prob = synth.make_problem(env="halfcheetah", context=True, E=5, m=1, H=5, trained_like=True, with_back=True, seed=9)
    eng = make_engine(prob, p=5, H=5)
    eng.train_configure(1e-2, WD, CWD, 1.0, 0.5, max_batch=64)
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, (1, 8, 5, 6))
    eps = rng.standard_normal((5, 1, 8, 5, 18))
    ctx = eng.context_forward(prob["cp_obs"], prob["cp_act"])
    r0 = eng.rollout_returns(prob["obs"], ctx, acts, eps=eps).cpu().numpy()
    batch = synth.make_train_batch(prob, B=64, seed=5)
    dev = _dev_batch(eng, batch, True, True)
    for _ in range(5):
        eng.train_step(dev, train=True)
    ctx = eng.context_forward(prob["cp_obs"], prob["cp_act"])
    r1 = eng.rollout_returns(prob["obs"], ctx, acts, eps=eps).cpu().numpy()   # repacks lazily
    assert rel_err(r1, r0) > 1e-4
    # and equals a fresh engine loaded with the trained parameters
    prob2 = dict(prob)
    prob2["ff"] = {k: v.cpu().numpy() for k, v in eng.nets["ff_model"].items()}
    prob2["cp"] = {k: v.cpu().numpy() for k, v in eng.nets["context_model"].items()}
    prob2["back"] = {k: v.cpu().numpy() for k, v in eng.nets["backward_model"].items()}
    eng2 = make_engine(prob2, p=5, H=5)
    ctx2 = eng2.context_forward(prob["cp_obs"], prob["cp_act"])
    r2 = eng2.rollout_returns(prob["obs"], ctx2, acts, eps=eps).cpu().numpy()
    np.testing.assert_array_equal(r1, r2)


def test_train_step_rows_equals_gathered_batch(gpu):
    """cadm_train_step_rows (rows addressed as (window, future offset) inside the kernels) == cadm_train_step on the
    batch gathered the way the reference's _preprocess_inputs + bootstrap indexing would (dynamics.py:676-696,:478-503)."""
    env, E, B, N, F, Hh = "halfcheetah", 5, 48, 40, 3, 10
    prob = synth.make_problem(env=env, context=True, E=E, trained_like=True, with_back=True, seed=41)
    r = np.random.default_rng(5)
    D, A = 18, 6
    ds = dict(obs=r.standard_normal((N, F * D)), act=r.standard_normal((N, F * A)), delta=r.standard_normal((N, F * D)),
              obs_next=r.standard_normal((N, F * D)), back_delta=r.standard_normal((N, F * D)),
              cp_obs=0.1 * r.standard_normal((N, D * Hh)), cp_act=r.uniform(-1, 1, (N, A * Hh)))
    fb = r.uniform(size=(N, F)) < 0.8
    w, f = np.nonzero(fb)
    idx = r.integers(0, w.shape[0], size=(E, B))
    results = []
    for mode in ("rows", "gathered"):
        eng = make_engine(prob, p=E)
        eng.train_configure(1e-3, WD, CWD, 1.0, 0.5, max_batch=B)
        dev = {k: eng._t(v) for k, v in ds.items()}
        tw, tf_, ti = (torch.as_tensor(x, device=eng.device) for x in (w, f, idx))
        for _ in range(3):
            if mode == "rows":
                losses = eng.train_step_rows(dev, F, tw, tf_, ti, train=True)
            else:
                ww, ff = tw[ti], tf_[ti]
                batch = {k: dev[k].view(N, F, -1)[ww, ff] for k in ("obs", "act", "delta", "obs_next", "back_delta")}
                batch["cp_obs"], batch["cp_act"] = dev["cp_obs"][ww], dev["cp_act"][ww]
                losses = eng.train_step({k: v.contiguous() for k, v in batch.items()}, train=True)
        results.append((losses.cpu().numpy(), {n: {k: v.cpu().numpy() for k, v in eng.nets[n].items()} for n in eng.net_names()}))
    np.testing.assert_array_equal(results[0][0], results[1][0])
    for n in results[0][1]:
        for k in results[0][1][n]:
            np.testing.assert_array_equal(results[0][1][n][k], results[1][1][n][k], err_msg="%s/%s" % (n, k))


SHAPES = [  # env, E, B, hidden_sizes, cp_hidden_sizes: every load shape of the chain kernels, odd widths, ragged tiles, E > 8
    ("pendulum", 2, 1, (128,) * 2, (64,)),                 # single row; D = 3 (odd heads), tiny K0
    ("slim_humanoid", 3, 17, (256,) * 3, (100, 50)),       # D = 45 (odd), cp widths not multiples of 4 / 16
    ("ant", 9, 33, (128,) * 2, (256, 128, 64)),            # E > 8: two members per XCD slot; K0 = 45 (odd)
    ("cartpole", 7, 40, (136,) * 2, (72, 36)),             # widths that are multiples of 4 but not of 16 / 32
    ("halfcheetah", 4, 50, (200,) * 4, (30, 22)),          # E = 4: two XCDs per member; even-only cp widths (b64 but not b128)
    ("halfcheetah", 2, 20, (512,) * 2, (256, 128, 64)),    # widest compiled hidden width: 16 weight blocks per k loop
]


@pytest.mark.parametrize("env,E,B,hid,cph", SHAPES)
def test_chain_kernel_shapes(gpu, env, E, B, hid, cph):
    prob = synth.make_problem(env=env, context=True, E=E, hidden_sizes=hid, cp_hidden_sizes=cph, trained_like=True,
                              with_back=True, seed=50)
    wd, cwd = WD[:len(hid)] + (WD[-1],), CWD[:len(cph)] + (CWD[-1],)
    cfg = dict(deterministic=False, back_coeff=0.5, weight_decay_coeff=1.0, weight_decays=wd, context_weight_decays=cwd,
               n_hidden=len(hid), n_cp_hidden=len(cph))
    batch = synth.make_train_batch(prob, B=B, seed=6)
    tb = {k: torch.tensor(v, dtype=torch.float64) for k, v in batch.items()}
    # losses
    eng = make_engine(prob, p=E)
    eng.train_configure(1e-3, wd, cwd, 1.0, 0.5, max_batch=B)
    got = eng.train_step(_dev_batch(eng, batch, True, True), train=False).cpu().numpy()
    ff, back, cp, st = _oracle_nets(prob, torch.float64, False)
    ref = otrain.train_losses(env, ff, back, cp, st, tb, cfg)
    want = np.array([float(ref["mse"]), float(ref["back_mse"]), float(ref["recon"])])
    np.testing.assert_allclose(got, want, rtol=5e-5, atol=5e-5)
    # gradients through the linearised Adam step (see test_gradients_via_linearised_adam)
    eng = make_engine(prob, p=E)
    eng.train_configure(1e6, wd, cwd, 1.0, 0.5, max_batch=B, beta1=0.0, beta2=0.0, epsilon=1e6)
    before = {n: {k: v.clone() for k, v in eng.nets[n].items()} for n in eng.net_names()}
    eng.train_step(_dev_batch(eng, batch, True, True), train=True)
    ff, back, cp, st = _oracle_nets(prob, torch.float64)
    out = otrain.train_losses(env, ff, back, cp, st, tb, cfg)
    grads = otrain.grads_of(out["loss"], {"ff_model": ff, "backward_model": back, "context_model": cp})
    for net in eng.net_names():
        for name, w0 in before[net].items():
            g_ref = grads[net][name]
            g_hip = (w0 - eng.nets[net][name]).cpu().numpy().astype(np.float64)
            if g_ref is None:
                assert np.abs(g_hip).max() == 0.0, "%s/%s moved although it has no gradient" % (net, name)
                continue
            g_ref = g_ref.numpy()
            scale = max(np.abs(g_ref).max(), 1e-12)
            err = np.abs(g_hip - g_ref).max() / scale
            assert err < 2e-3, "%s/%s gradient off: rel-to-max err %.3e (max |g| %.3e)" % (net, name, err, scale)


def test_first_adam_step_moves_every_trained_parameter_by_lr(gpu):
    """Oracle-independent pin of the TF1 Adam epilogue: from zero moments the first update is
    lr * sqrt(1-b2)/(1-b1) * (1-b1) g / (sqrt((1-b2) g^2) + eps) = lr * sign(g) * |g| / (|g| + eps / sqrt(1-b2)),
    i.e. every parameter with a gradient well above 3e-7 moves by lr (the context encoder's first layer sees gradients of a
    few 1e-6: 0.9 lr), none by more, and parameters without a gradient
    (output_logvar bias of the backward net, dynamics.py:213-240) stay put."""
    env, E, B, lr = "halfcheetah", 5, 128, 1e-3
    prob = synth.make_problem(env=env, context=True, E=E, trained_like=True, with_back=True, seed=60)
    eng = make_engine(prob, p=E)
    eng.train_configure(lr, WD, CWD, 1.0, 0.5, max_batch=B)
    batch = synth.make_train_batch(prob, B=B, seed=7)
    before = {n: {k: v.clone() for k, v in eng.nets[n].items()} for n in eng.net_names()}
    eng.train_step(_dev_batch(eng, batch, True, True), train=True)
    for net in eng.net_names():
        for name, w0 in before[net].items():
            d = (eng.nets[net][name] - w0).abs().cpu().numpy().ravel()
            if net == "backward_model" and name in ("output_logvar_bias", "max_logvar", "min_logvar"):
                assert d.max() == 0.0, "%s/%s has no gradient and must not move" % (net, name)
                continue
            assert d.max() <= lr * (1 + 1e-3), "%s/%s moved by %.3e > lr" % (net, name, d.max())   # fp32 rounding of w - lr at |w| ~ 10
            assert np.median(d) >= 0.8 * lr, "%s/%s: median |step| %.3e, expected ~lr" % (net, name, np.median(d))


def test_uncompiled_context_width_trains_and_plans(gpu):
    """The training chains take any layer width at run time; the planner's rollout kernel is instantiated per geometry: a
    context width the library does not carry (C = 7) is built on demand (cadm_amd.jit) at the first planner call."""
    prob = synth.make_problem(env="halfcheetah", context=True, E=3, C=7, trained_like=True, with_back=True, seed=70)
    eng = make_engine(prob, p=3)
    assert not eng.lib.cadm_rollout_builtin(eng._ctx)
    eng.train_configure(1e-3, WD, CWD, 1.0, 0.5, max_batch=20)
    batch = synth.make_train_batch(prob, B=20, seed=8)
    l = eng.train_step(_dev_batch(eng, batch, True, True), train=True)
    assert torch.isfinite(l).all()
    eng.repack()
    plan = eng.cem_plan(prob["obs"], prob["cp_obs"], prob["cp_act"], prob["init_mean"], prob["init_var"], 64)
    assert torch.isfinite(plan).all() and plan.abs().max() <= 1.0
