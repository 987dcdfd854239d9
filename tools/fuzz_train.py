#!/usr/bin/env python3
"""Randomised check of the fused training step against the oracle (developer tool; needs a GPU):
python tools/fuzz_train.py [seconds] [seed]

Every round draws a random configuration (env kind, context encoder and its widths, backward model, deterministic /
probabilistic, ensemble size, ragged batch size, hidden width incl. widths the planner is not compiled for, history length) and
checks (i) the forward losses [mse, back_mse, recon] against the fp64 oracle, (ii) every gradient tensor of the hand-written
backward pass -- read back directly (Adam's first moment after one step with beta1 = 0, developer library) -- element by
element against torch.autograd on the oracle (<= 1e-5 of the tensor's max), and that variables without a gradient do not move.
Hidden widths are drawn per layer now and then (zero-padded in the engine).  The chain kernel's flavour (8 or 4 waves per workgroup) is forced at random in the gradient leg.  Same bars as
tests/test_gpu_train.py."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from cadm_amd import _lib, synth
from oracle import train as otrain

ENVS = ["halfcheetah", "cripple_halfcheetah", "ant", "slim_humanoid", "pendulum", "cartpole"]
WD = (0.000025, 0.00005, 0.000075, 0.000075, 0.0001)


def main():
    t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    rounds, worst_l, worst_g, kinks = 0, 0.0, 0.0, 0
    while time.time() < t_end:
        env = ENVS[rng.integers(len(ENVS))]
        context = bool(rng.integers(2))
        with_back = bool(context and rng.integers(2))
        det = bool(rng.integers(3) == 0)
        E, B = int(rng.choice([1, 2, 3, 5])), int(rng.integers(1, 300))
        hid = int(rng.choice([64, 96, 128, 200, 256]))
        hids = (hid,) * 4 if rng.integers(4) else tuple(int(rng.choice([48, 64, 96, 128, 200])) for _ in range(4))      # unequal now and then
        ncp = int(rng.integers(1, 4))
        cph = tuple(int(rng.choice([8, 24, 64, 130])) for _ in range(ncp))
        Hh = int(rng.integers(1, 6))
        prob = synth.make_problem(env=env, context=context, E=E, trained_like=True, with_back=with_back, hidden_sizes=hids,
                                  cp_hidden_sizes=cph, Hh=Hh, seed=int(rng.integers(1 << 30)))
        cwd = tuple(0.00003 * (i + 1) for i in range(ncp + 1))
        cfg = dict(deterministic=det, back_coeff=0.5 if with_back else 0.0, weight_decay_coeff=1.0, weight_decays=WD,
                   context_weight_decays=cwd, n_hidden=4, n_cp_hidden=ncp)
        tag = "%s ctx=%d back=%d det=%d E=%d B=%d hid=%s cp=%s Hh=%d" % (env, context, with_back, det, E, B, hids, cph, Hh)
        batch = synth.make_train_batch(prob, B=B, seed=int(rng.integers(1 << 30)))
        keys = ["obs", "act", "delta"] + (["obs_next", "back_delta"] if with_back else []) + (["cp_obs", "cp_act"] if context else [])
        nets64 = lambda rg: (otrain.to_torch(prob["ff"], torch.float64, rg),
                             otrain.to_torch(prob["back"], torch.float64, rg) if with_back else None,
                             otrain.to_torch(prob["cp"], torch.float64, rg) if context else None)
        st = otrain.to_torch(prob["stats"], torch.float64)
        tb = {k: torch.tensor(v, dtype=torch.float64) for k, v in batch.items()}
        # (i) forward losses
        eng = synth.make_engine(prob, p=E, deterministic=det)
        eng.train_configure(1e-3, WD, cwd, 1.0, cfg["back_coeff"], max_batch=B)
        got = eng.train_step({k: eng._t(batch[k]) for k in keys}, train=False).cpu().numpy()
        ff, back, cp = nets64(False)
        ref = otrain.train_losses(env, ff, back, cp, st, tb, cfg)
        want = np.array([float(ref["mse"]), float(ref["back_mse"]), float(ref["recon"])])
        el = float(np.max(np.abs(got - want) / (np.abs(want) + 1.0)))
        assert el <= 1e-4, "%s: losses %r vs %r" % (tag, got, want)
        worst_l = max(worst_l, el)
        eng.close()
        # (ii) gradients, read back directly
        eng = synth.make_engine(prob, p=E, deterministic=det, lib=_lib.load_dev())
        eng.train_configure(1e-3, WD, cwd, 1.0, cfg["back_coeff"], max_batch=B, beta1=0.0)
        # 0 = the launcher's pick; 4 / 8 waves per workgroup forced; + 16: large-batch mapping (work items spread over all XCDs, one forward
        # launch per net with the context vector read back), + 32: member-affine mapping, joint launch (round 5)
        # + 64: one pass of the summed context gradient down the encoder (the large-batch backward path)
        flavour = int(rng.choice([0, 4, 8, 4 + 16, 8 + 16, 4 + 32, 4 + 16 + 64, 8 + 64]))
        eng._check(eng.lib.cadm_dev_set_train_flavour(eng._ctx, flavour), "cadm_dev_set_train_flavour")
        tag += " flavour=%d" % flavour
        before = {n: {k: v.clone() for k, v in eng.nets[n].items()} for n in eng.net_names()}
        eng.train_step({k: eng._t(batch[k]) for k in keys}, train=True)
        def off_tensors(thr):
            """[(net, name, err)] of the gradient tensors further than 1e-5 of their max from fp64 autograd (context relu kink at thr)"""
            ff, back, cp = nets64(True)
            out = otrain.train_losses(env, ff, back, cp, st, tb, dict(cfg, cp_relu_threshold=thr))
            grads = otrain.grads_of(out["loss"], {"ff_model": ff, "backward_model": back, "context_model": cp})
            bad, worst = [], 0.0
            for net in eng.net_names():
                true = eng.param_shapes_true(net)
                for name, w0 in before[net].items():
                    g_ref = grads[net][name]
                    if g_ref is None:
                        assert torch.equal(w0, eng.nets[net][name]), "%s: %s/%s moved although it has no gradient" % (tag, net, name)
                        continue
                    g_hip = eng.dev_read_adam_moment(net, name).cpu().numpy().astype(np.float64)[tuple(slice(0, k) for k in true[name])]
                    g_ref = g_ref.numpy()
                    err = float(np.abs(g_hip - g_ref).max() / max(np.abs(g_ref).max(), 1e-300))
                    if err > 1e-5:
                        bad.append((net, name, err))
                    else:
                        worst = max(worst, err)
            return bad, worst
        bad, w = off_tensors(0.0)
        if bad and context and all(net == "context_model" for net, _, _ in bad):
            # Context-encoder pre-activations within fp32 roundoff of 0: the fp32 forward pass and the fp64 oracle put such a relu unit on
            # different sides of its kink, and one batch row's contribution to the gradient differs (one configuration in a few hundred;
            # two units in one configuration now and then).  Accepted only if the gradient matches the oracle with each such unit
            # (|z| within 3e-6 of the layer's rms in fp64; the six closest) forced to one side or the other -- every combination is tried.
            import itertools
            _, _, cp64 = nets64(False)
            zs = otrain.context_preacts(cp64, st, tb, cfg)
            # (fp32 roundoff of these dot products is ~1e-7 of the layer's rms pre-activation: units within 3e-6 of it, the six closest at most)
            cands = []
            for l, z in enumerate(zs):
                bar = 3e-6 * max(1.0, float(z.pow(2).mean().sqrt()))
                cands += [(float(z[tuple(idx)].abs()) / bar, (l,) + tuple(int(v) for v in idx)) for idx in torch.nonzero(z.abs() < bar)]
            units = [u for _, u in sorted(cands)[:6]]
            if os.environ.get("FUZZ_DIAG"):
                print("   units near the kink:", sorted(cands)[:8], flush=True)
            if 0 < len(units) <= 6:
                for signs in itertools.product((2e-5, -2e-5), repeat=len(units)):
                    thr = {l: torch.zeros_like(z) for l, z in enumerate(zs)}
                    for (l, e_, b_, u_), sg in zip(units, signs):
                        thr[l][e_, b_, u_] = sg
                    bad2, w2 = off_tensors(thr)
                    if not bad2:
                        print("relu kink: %s -- %s off by %.1e against relu'(0 +- roundoff); exact with the %d unit(s) %s forced %s"
                              % (tag, bad[0][1], bad[0][2], len(units), units, ["off" if x > 0 else "on" for x in signs]), flush=True)
                        bad, w, kinks = [], w2, kinks + 1
                        break
        assert not bad, "%s: gradient off: %s" % (tag, ", ".join("%s/%s %.3e of the tensor's max" % b for b in bad))
        worst_g = max(worst_g, w)
        eng.close()
        rounds += 1
    print("train fuzz OK: %d random configurations (%d with a context relu unit on its kink, see above); worst loss deviation %.2e, worst gradient "
          "deviation %.2e of the tensor's scale" % (rounds, kinks, worst_l, worst_g))


if __name__ == "__main__":
    main()
