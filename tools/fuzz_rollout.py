#!/usr/bin/env python3
"""Randomised differential test of the rollout kernels (developer tool; needs a GPU):  python tools/fuzz_rollout.py [seconds] [seed] [variant library]

Every round draws a random problem (env kind, context / vanilla, hidden width, ensemble size, particles, candidates, batch of
envs m, horizon, noise mode, CEM iteration parity) and checks
  * the one-tile and the two-tile flavour of the cooperative kernel and both wave-tile flavours agree BIT FOR BIT (returns and trajectories),
  * both agree with the fp32-MFMA comparison kernel of round 1 to 2e-5 of the trajectory scale,
  * a second launch reproduces the first bit for bit.
The hazards inline-asm MFMAs can hide from the compiler are timing- and allocation-dependent; this sweeps many
instantiations and launch shapes that the fixed test cases do not."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from cadm_amd import _lib, synth

ENVS = ["halfcheetah", "cripple_halfcheetah", "ant", "slim_humanoid", "pendulum", "cartpole"]


def run(eng, prob, ctx, acts, eps, flavour, **kw):
    if flavour:
        eng.dev_set_rollout("xdl", row_tiles=int(flavour))
    if os.environ.get("CADM_FUZZ_VERBOSE"):
        print("   flavour %r eps=%s" % (flavour, "injected" if eps is not None else "device"), flush=True)
    rows, traj = eng.rollout_returns(prob["obs"], ctx, acts, eps=eps, want_traj=True, **kw)
    torch.cuda.synchronize()
    return rows.cpu().numpy(), traj.cpu().numpy()


def fuzz(seconds=60.0, seed=0, hids=(128, 200, 200, 256, 512), lib_path=None):
    """-> (rounds, worst xdl-vs-fp32 deviation); raises AssertionError on the first failing problem."""
    t_end = time.time() + seconds
    rng = np.random.default_rng(seed)
    rounds, worst = 0, 0.0
    while time.time() < t_end:
        env = ENVS[rng.integers(len(ENVS))]
        context = bool(rng.integers(2))
        hid = int(rng.choice(list(hids)))
        E = int(rng.choice([1, 2, 5]))
        p = E * int(rng.integers(1, 5))
        m, n, H = int(rng.integers(1, 4)), int(rng.integers(1, 120)), int(rng.integers(1, 12))
        if rng.integers(5) == 0:      # a big batch now and then: full workgroups and several rounds of the wave-tile kernel, mixed launcher plans
            n, H = int(rng.integers(500, 4000)), int(rng.integers(1, 4))
        det = bool(rng.integers(4) == 0)
        prob = synth.make_problem(env=env, context=context, E=E, m=m, H=H, hidden_sizes=(hid,) * 4, trained_like=True,
                                  seed=int(rng.integers(1 << 30)))
        discrete = prob["discrete"]
        A, D = prob["A"], prob["D"]
        if discrete:
            raw = rng.integers(0, A, (m, n, H))
            acts = np.eye(A, dtype=np.float32)[raw]
        else:
            acts = rng.uniform(-1, 1, (m, n, H, A)).astype(np.float32)
        kw = dict(norm_actions=not discrete, it=int(rng.integers(2)))
        out = {}
        if os.environ.get("CADM_FUZZ_VERBOSE"):      # (a kernel that faults takes the process down: say what is about to run)
            print("fuzz: %s ctx=%d hid=%d E=%d p=%d m=%d n=%d H=%d det=%d" % (env, context, hid, E, p, m, n, H, det), flush=True)
        for kind in ("f32", "xdl"):
            eng = synth.make_engine(prob, p=p, deterministic=det, lib=_lib.load_dev(lib_path) if lib_path else _lib.load_dev())
            if kind == "f32":
                eng.dev_set_rollout("f32")
            ctx = eng.context_forward(prob["cp_obs"], prob["cp_act"]) if context else None
            a_dev = eng._t(acts)
            mode = int(rng.integers(2)) if kind == "f32" else mode
            eps = None
            if not det:
                if kind == "f32":
                    eps_np = rng.standard_normal((H, m, n, p, D)).astype(np.float32) if mode else None
                eps = None if eps_np is None else eng._t(eps_np)
            extra = dict(kw, seed=7, call=3) if (eps is None and not det) else kw
            if kind == "f32":
                try:
                    out["f32"] = run(eng, prob, ctx, a_dev, eps, None, **extra)
                except Exception as exc:            # the comparison kernel does not fit every geometry in LDS (wide layers)
                    if "LDS" not in str(exc):
                        raise
                    out["f32"] = None
            else:
                out["mt1"] = run(eng, prob, ctx, a_dev, eps, "1", **extra)
                out["mt2"] = run(eng, prob, ctx, a_dev, eps, "2", **extra)
                if hid <= 256:      # the wave-tile kernel (rollout_wt.h) exists for 3-product widths
                    out["wt8"] = run(eng, prob, ctx, a_dev, eps, "3", **extra)
                    out["wt4"] = run(eng, prob, ctx, a_dev, eps, "4", **extra)
                out["mt1b"] = run(eng, prob, ctx, a_dev, eps, "1", **extra)
                out["plan"] = run(eng, prob, ctx, a_dev, eps, "0", **extra)      # whatever mix of flavours the launcher picks
            eng.close()
        tag = "%s ctx=%d hid=%d E=%d p=%d m=%d n=%d H=%d det=%d" % (env, context, hid, E, p, m, n, H, det)
        for a, b in (("mt1", "mt2"), ("mt1", "mt1b"), ("mt1", "wt8"), ("mt1", "wt4"), ("mt1", "plan")):
            if b not in out:
                continue
            for x, y in zip(out[a], out[b]):
                assert np.array_equal(x, y, equal_nan=True), "%s: %s vs %s differ" % (tag, a, b)
        if out["f32"] is None or (not det and eps_np is None and hid == 128):
            rounds += 1       # (the round-1 comparison kernel draws a different device noise stream at HID = 128; the production
            continue          #  kernel's stream is pinned to the oracle's by tests/test_gpu_planner.py at every width)
        t32, tx = out["f32"][1], out["mt1"][1]
        ok = np.isfinite(t32) & np.isfinite(tx)
        scale = max(float(np.sqrt(np.mean(t32[ok] ** 2))), 1e-6) if ok.any() else 1.0
        err = float(np.abs(t32[ok] - tx[ok]).max() / scale) if ok.any() else 0.0
        worst = max(worst, err)
        assert err <= 2e-4 * max(1, H), "%s: xdl vs fp32 kernel %.2e" % (tag, err)
        rounds += 1
    return rounds, worst


def main():
    # (argv[3]: a variant build of the developer library, tools/build_variant.sh -- built for HID 200 only unless told otherwise)
    lib_path = os.path.join(ROOT, "cadm_amd", sys.argv[3]) if len(sys.argv) > 3 else None
    rounds, worst = fuzz(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0,
                         hids=(200,) if lib_path else (128, 200, 200, 256, 512), lib_path=lib_path)
    print("fuzz OK: %d random problems; worst xdl-vs-fp32 trajectory deviation %.2e of the rms" % (rounds, worst))


if __name__ == "__main__":
    main()
