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


def _dev_engine(prob, p, **kw):
    """An engine on the DEVELOPER library: the product's objects plus the hook that reads Adam's moment buffers."""
    from cadm_amd import _lib
    return make_engine(prob, p=p, lib=_lib.load_dev(), **kw)


def _grad_report(g_hip, g_ref):
    """(max |d| / max |ref|,  max elementwise |d| / |ref| over |ref| >= 0.25 rms(ref))"""
    scale = max(np.abs(g_ref).max(), 1e-300)
    d = np.abs(g_hip - g_ref)
    big = np.abs(g_ref) >= 0.25 * np.sqrt(np.mean(g_ref ** 2))
    return d.max() / scale, (d[big] / np.abs(g_ref[big])).max() if big.any() else 0.0


def _check_gradients(eng, grads, what):
    """Every gradient tensor the fused step produced, read back DIRECTLY (Adam's first moment after one step with beta1 = 0 is
    the gradient, exactly: m = 0 m + 1 g) and compared element by element with torch.autograd on the fp64 oracle:
    <= 1e-5 of the tensor's max, and <= 1e-4 PURE relative on every element with |g| >= 0.25 rms (fp32 forward / backward over a
    256-row batch: roundoff-scale bars; a sign or scale error in any slice of any tensor fails them)."""
    checked, worst = 0, (0.0, 0.0, "")
    for net in eng.net_names():
        for name in eng.nets[net]:
            g_ref = grads[net][name]
            if g_ref is None:      # TF skips variables without a gradient (backward model's logvar head, dynamics.py:213-240)
                continue
            g_hip = eng.dev_read_adam_moment(net, name).cpu().numpy().astype(np.float64)
            assert np.isfinite(g_hip).all()
            to_max, pure = _grad_report(g_hip, g_ref.numpy())
            assert to_max <= 1e-5, "%s %s/%s gradient: max err %.2e of the tensor's max" % (what, net, name, to_max)
            assert pure <= 1e-4, "%s %s/%s gradient: pure relative err %.2e on |g| >= 0.25 rms" % (what, net, name, pure)
            if to_max > worst[0]:
                worst = (to_max, pure, "%s/%s" % (net, name))
            checked += 1
    print("%s: %d gradient tensors, worst %.2e of max (pure-relative %.2e) at %s" % ((what, checked) + worst))
    return checked


# chain kernel: 8 waves x 1 workgroup per CU (latency) / 4 waves x 3 per CU (throughput); + 16 + 64: the whole large-batch path forced at these
# small batches -- work items spread over all XCDs, one forward launch per net (context vector read back), ONE pass of the summed context
# gradient down the encoder (not bit-identical to one pass per net: this is its gradient test)
@pytest.mark.parametrize("flavour", [8, 4, 4 + 16 + 64, 8 + 16 + 64])
@pytest.mark.parametrize("env,context,with_back,det,E,B", CASES)
def test_gradients_elementwise_vs_fp64_autograd(gpu, env, context, with_back, det, E, B, flavour):
    prob = synth.make_problem(env=env, context=context, E=E, trained_like=True, with_back=with_back, seed=22)
    eng = _dev_engine(prob, E, deterministic=det)
    eng._check(eng.lib.cadm_dev_set_train_flavour(eng._ctx, flavour), "cadm_dev_set_train_flavour")
    bc = 0.5 if with_back else 0.0
    eng.train_configure(1e-3, WD, CWD, 1.0, bc, max_batch=B, beta1=0.0)
    batch = synth.make_train_batch(prob, B=B, seed=3)
    before = {n: {k: v.clone() for k, v in eng.nets[n].items()} for n in eng.net_names()}
    eng.train_step(_dev_batch(eng, batch, context, with_back), train=True)
    ff, back, cp, st = _oracle_nets(prob, torch.float64)
    tb = {k: torch.tensor(v, dtype=torch.float64) for k, v in batch.items()}
    out = otrain.train_losses(env, ff, back, cp, st, tb, _cfg(prob, det, bc))
    grads = otrain.grads_of(out["loss"], {"ff_model": ff, "backward_model": back, "context_model": cp})
    assert _check_gradients(eng, grads, "%s E=%d B=%d" % (env, E, B)) >= 8
    for net in eng.net_names():       # variables without a gradient must not move (TF's minimize skips them)
        for name, w0 in before[net].items():
            if grads[net][name] is None:
                assert torch.equal(w0, eng.nets[net][name]), "%s/%s moved although it has no gradient" % (net, name)


def test_chain_kernel_flavours_agree_bitwise(gpu):
    """Round 5: the chain kernel exists with 8 waves per workgroup (one workgroup per CU: the reference's batch of 256) and with 4 (three
    workgroups per CU: large batches; the launcher picks by the number of work items).  A row's arithmetic -- operand order of every
    MFMA chain, epilogues, the loss phase's reductions -- is the same in both: six training steps from the same state end in bit-identical
    weights, moments and losses, at a batch size on either side of the switch -- and under either mapping of work items to XCDs."""
    for B in (64, 600, 1100):          # (1100: the launcher's own pick is the 4-wave flavour with the items spread)
        prob = synth.make_problem(env="halfcheetah", context=True, E=5, trained_like=True, with_back=True, seed=29)
        batch = synth.make_train_batch(prob, B=B, seed=5)
        end = {}
        # (+ 16: the large-batch path -- work items spread over all XCDs, the launcher's mapping once a member's items exceed one round of its XCD,
        #  and one forward launch per net, the backward model's reading the context vector back instead of recomputing it --, + 32: member-affine, joint)
        for fl in (8, 4, 0, 8 + 16, 4 + 16, 4 + 32):
            eng = _dev_engine(prob, 5)
            eng._check(eng.lib.cadm_dev_set_train_flavour(eng._ctx, fl), "cadm_dev_set_train_flavour")
            eng.train_configure(1e-3, WD, CWD, 1.0, 0.5, max_batch=B)
            dev = _dev_batch(eng, batch, True, True)
            losses = [eng.train_step(dev, train=True).cpu().numpy() for _ in range(6)]
            ev = eng.train_step(dev, train=False).cpu().numpy()
            end[fl] = (losses, ev, {n: {k: v.cpu().numpy() for k, v in eng.nets[n].items()} for n in eng.net_names()})
            eng.close()
        for fl in (4, 0, 8 + 16, 4 + 16, 4 + 32):
            for a, b in zip(end[8][0], end[fl][0]):
                np.testing.assert_array_equal(a, b)
            np.testing.assert_array_equal(end[8][1], end[fl][1])
            for n in end[8][2]:
                for k in end[8][2][n]:
                    np.testing.assert_array_equal(end[8][2][n][k], end[fl][2][n][k], err_msg="B=%d flavour %d %s/%s" % (B, fl, n, k))


def test_large_batch_path_switches_at_a_documented_batch_size(gpu):
    """the one-pass context backward of large batches is not bit-identical to the per-net passes, so the batch size at which the
    step switches to it must not depend on the device's CU count.  The rule is evaluated for 256 CUs (csrc/train.hip: CADM_LARGE_BATCH_CUS):
    5 members + backward model switch between B = 1840 and B = 1856.  Observed through the result: the launcher's own pick (flavour 0) is
    bit-identical to the forced joint path (+ 32 + 128) at 1840 and to the forced large-batch path (+ 16 + 64) at 1856 -- and the two forced
    paths differ from each other at both (else the test would prove nothing)."""
    prob = synth.make_problem(env="halfcheetah", context=True, E=5, trained_like=True, with_back=True, seed=31)
    for B, expect in ((1840, 32 + 128), (1856, 16 + 64)):
        batch = synth.make_train_batch(prob, B=B, seed=6)
        end = {}
        for fl in (0, 32 + 128, 16 + 64):
            eng = _dev_engine(prob, 5)
            eng._check(eng.lib.cadm_dev_set_train_flavour(eng._ctx, fl), "cadm_dev_set_train_flavour")
            eng.train_configure(1e-3, WD, CWD, 1.0, 0.5, max_batch=B)
            dev = _dev_batch(eng, batch, True, True)
            for _ in range(2):
                eng.train_step(dev, train=True)
            end[fl] = {n: {k: v.cpu().numpy() for k, v in eng.nets[n].items()} for n in eng.net_names()}
            eng.close()
        same = lambda a, b: all(np.array_equal(end[a][n][k], end[b][n][k]) for n in end[a] for k in end[a][n])
        assert not same(32 + 128, 16 + 64), "B=%d: the forced paths agree bit for bit -- nothing is being told apart" % B
        assert same(0, expect), "B=%d: the launcher did not take the %s path" % (B, "joint" if expect == 32 + 128 else "large-batch")


def test_adam_steps_match_tf1_semantics_elementwise(gpu):
    """6 real Adam steps (lr 1e-3, beta 0.9 / 0.999, eps 1e-8).  Before every step the oracle is set to the device's state -- weights
    and both moment buffers, read through the developer hook -- takes the fp64 autograd gradient there and applies the TF1 closed
    form (ApplyAdam: lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); w -= lr_t m / (sqrt(v) + eps)); the device's step must land on the same
    weights ELEMENT BY ELEMENT, every element of every tensor: |dw - dw_ref| <= 2e-4 lr + |du/dg| * (1e-5 max|g|) + ulp(w)/2.
    The middle term is the gradient bar of test_gradients_elementwise_vs_fp64_autograd carried through the update u(g) = lr_t m /
    (sqrt(v) + eps): Adam's update is a sign-like function of the gradient history, so for an element whose gradients are at
    roundoff level du/dg is large (in TF as much as here) and the bound says by how much; everywhere else it vanishes against
    2e-4 lr.  No element is excluded."""
    env, E, B, lr = "halfcheetah", 5, 128, 1e-3
    prob = synth.make_problem(env=env, context=True, E=E, trained_like=True, with_back=True, seed=30)
    eng = _dev_engine(prob, E)
    eng.train_configure(lr, WD, CWD, 1.0, 0.5, max_batch=B)
    batch = synth.make_train_batch(prob, B=B, seed=4)
    dev = _dev_batch(eng, batch, True, True)
    tb = {k: torch.tensor(v, dtype=torch.float64) for k, v in batch.items()}
    st = otrain.to_torch(prob["stats"], torch.float64)
    names = {"ff_model": "ff", "backward_model": "back", "context_model": "cp"}
    b1, b2, eps = 0.9, 0.999, 1e-8
    losses, loose, total, worst = [], 0, 0, 0.0
    for step in range(1, 7):
        # (the backward model's logvar bounds are never trained -- no gradient, no Adam slot: dynamics.py:213-240)
        untrained = lambda n, k: n == "backward_model" and k in ("max_logvar", "min_logvar")
        mom = lambda n, k, second: (np.zeros(tuple(eng.nets[n][k].shape)) if untrained(n, k)
                                    else eng.dev_read_adam_moment(n, k, second=second).cpu().numpy().astype(np.float64))
        state = {n: {k: (v.cpu().numpy().astype(np.float64), mom(n, k, False), mom(n, k, True)) for k, v in eng.nets[n].items()}
                 for n in eng.net_names()}
        nets = {n: otrain.to_torch({k: w for k, (w, _, _) in state[n].items()}, torch.float64, True) for n in eng.net_names()}
        out = otrain.train_losses(env, nets["ff_model"], nets["backward_model"], nets["context_model"], st, tb, _cfg(prob, False, 0.5))
        grads = otrain.grads_of(out["loss"], nets)
        got = eng.train_step(dev, train=True).cpu().numpy()
        losses.append(got)
        np.testing.assert_allclose(got, [float(out["mse"].detach()), float(out["back_mse"].detach()), float(out["recon"].detach())],
                                   rtol=5e-5, atol=5e-5)
        lr_t = lr * np.sqrt(1.0 - b2 ** step) / (1.0 - b1 ** step)
        for n in eng.net_names():
            for k, (w0, m0, v0) in state[n].items():
                w1 = eng.nets[n][k].cpu().numpy().astype(np.float64)
                if grads[n][k] is None:
                    assert np.array_equal(w1, w0), "%s/%s moved although it has no gradient" % (n, k)
                    continue
                g = grads[n][k].numpy()
                m1, v1 = b1 * m0 + (1 - b1) * g, b2 * v0 + (1 - b2) * g * g
                dw_ref = -lr_t * m1 / (np.sqrt(v1) + eps)
                dw = w1 - w0
                assert np.abs(dw).max() <= 1.001 * lr_t * max(1.0, np.abs(m1 / (np.sqrt(v1) + eps)).max()) + np.abs(w0).max() * 2.0 ** -22 + 1e-7     # no element runs away
                sv = np.sqrt(v1)
                dudg = lr_t * np.abs((1 - b1) / (sv + eps) - m1 * (1 - b2) * g / (np.maximum(sv, 1e-300) * (sv + eps) ** 2))
                bar = 2e-4 * lr + dudg * 1e-5 * np.abs(g).max() + np.abs(w0) * 2.0 ** -23
                err = np.abs(dw - dw_ref)
                loose += int((bar > 2e-3 * lr).sum())
                total += g.size
                worst = max(worst, float((err / lr).max()))
                assert (err <= bar).all(), "step %d %s/%s: update off by %.2e lr at the worst element (its bar %.2e lr)" % (
                    step, n, k, (err / lr)[np.argmax(err - bar)], (bar / lr)[np.argmax(err - bar)])
    assert losses[-1][2] < losses[0][2], "training loss did not decrease"
    # (informational: ~10 % of the elements have |g| < 1e-2 max|g| of their tensor, where the propagated term exceeds 2e-3 lr)
    assert loose <= 0.25 * total, "%d of %d element-updates had a propagated bar above 2e-3 lr" % (loose, total)
    print("adam: 6 steps, %d element-updates, worst deviation %.2e lr; %d (%.2f %%) were ill-conditioned enough for a bar above 2e-3 lr"
          % (total, worst, loose, 100.0 * loose / total))


@pytest.mark.parametrize("B", [256, 37, 1024])
def test_eval_losses_equal_the_training_steps_reduction(gpu, B):
    """An evaluation step hands its loss partials to the LAST-arriving workgroup inside the forward launch (device-coherent sc1
    stores, drained, then an agent-scope atomic; sc1 loads on the reader -- train.hip, closing phase); a training step leaves
    them to a workgroup of the NEXT launch (kernel boundary).  Same partials, two hand-offs: the sums must agree to fp32
    reduction-order roundoff.  Two very different batches alternate for 300 rounds, so a partial read stale (the previous
    round's) would move the sum by orders of magnitude more than the bar; lr = 0 keeps the weights where they are."""
    env, E = "halfcheetah", 5
    prob = synth.make_problem(env=env, context=True, E=E, trained_like=True, with_back=True, seed=61)
    eng = make_engine(prob, p=E)
    eng.train_configure(0.0, WD, CWD, 1.0, 0.5, max_batch=B)
    b1 = synth.make_train_batch(prob, B=B, seed=8)
    b2 = synth.make_train_batch(prob, B=B, seed=9)
    for k in ("delta", "back_delta"):
        b2[k] = 7.0 * b2[k]
    devs = [_dev_batch(eng, b, True, True) for b in (b1, b2)]
    want = [eng.train_step(d, train=True).cpu().numpy() for d in devs]          # reduction across the kernel boundary
    assert abs(want[0][2] - want[1][2]) > 0.5 * abs(want[0][2])                  # the two batches are told apart easily
    outs = []
    for r in range(300):
        outs.append(eng.train_step(devs[r & 1], train=False))                    # no sync in between: back-to-back launches
    outs = torch.stack(outs).cpu().numpy()
    for r in range(300):
        np.testing.assert_allclose(outs[r], want[r & 1], rtol=3e-6, atol=0, err_msg="evaluation step %d" % r)
    again = [eng.train_step(d, train=True).cpu().numpy() for d in devs]
    for w, a in zip(want, again):
        np.testing.assert_array_equal(w, a)                                      # lr = 0: nothing moved, the reduction is deterministic


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


@pytest.mark.parametrize("flavour", [None, 4 + 16 + 64])      # the product's launch plan / the large-batch plan forced (developer library)
def test_train_step_rows_equals_gathered_batch(gpu, flavour):
    """cadm_train_step_rows (rows addressed as (window, future offset) inside the kernels) == cadm_train_step on the
    batch gathered the way the reference's _preprocess_inputs + bootstrap indexing would (dynamics.py:676-696,:478-503).
    Also under the large-batch launch plan (one forward launch per net with the context vector read back, one pass of the summed
    context gradient): its extra input tiles address workspace rows, not dataset rows."""
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
        eng = make_engine(prob, p=E) if flavour is None else _dev_engine(prob, E)
        if flavour is not None:
            eng._check(eng.lib.cadm_dev_set_train_flavour(eng._ctx, flavour), "cadm_dev_set_train_flavour")
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
    # gradients, read back directly (see test_gradients_elementwise_vs_fp64_autograd)
    eng = _dev_engine(prob, E)
    eng.train_configure(1e-3, wd, cwd, 1.0, 0.5, max_batch=B, beta1=0.0)
    eng.train_step(_dev_batch(eng, batch, True, True), train=True)
    ff, back, cp, st = _oracle_nets(prob, torch.float64)
    out = otrain.train_losses(env, ff, back, cp, st, tb, cfg)
    grads = otrain.grads_of(out["loss"], {"ff_model": ff, "backward_model": back, "context_model": cp})
    _check_gradients(eng, grads, "%s E=%d B=%d hid=%s" % (env, E, B, hid))


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
