"""How far inside the 1e-5 parity bar the split-f16 rollout kernel sits -- measured, not asserted in prose.

Shared by tests/test_gpu_precision.py (asserts), tools/precision_report.py (writes profiles/r3_precision.md) and the
`parity` block of bench.py's cpu_baseline leg.  Test infrastructure: uses the oracle as the checker.

The production kernel evaluates every fp32 product as three f16 MFMA products of 2-way split operands
(cadm_amd/csrc/xdl_geo.h); the developer library's fp32-MFMA kernel (csrc/dev/rollout_f32.h) evaluates the same rows
with fp32 operands.  Both are compared with the fp64 numpy oracle (truth) and the fp32 numpy oracle ("what TF-CPU
computes up to summation order") on the same inputs: BASELINE cfg2's full size (ens=5, part=20, cand=200), trained-like
weights (reference core/utils.py:341-365, :441-472).
"""
import numpy as np

from cadm_amd import synth
from oracle import envs as oenvs
from oracle import nets as onets
from oracle import planner as oplanner


def _oracle(prob, dt):
    o = dict(env=oenvs.make_env(prob["env"]), ff=onets.cast_params(prob["ff"], dt),
             cp=None if prob["cp"] is None else onets.cast_params(prob["cp"], dt), st=onets.cast_stats(prob["stats"], dt))
    for k in ("obs", "cp_obs", "cp_act"):
        o[k] = prob[k].astype(dt)
    return o


def err_stats(got, ref):
    """Error of `got` against `ref` (fp64 truth): normwise max|d|/max|ref|, rms-scaled max|d|/rms(ref), and the PURE
    elementwise relative error over the elements with |ref| >= 0.25 rms (no floor)."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    d = np.abs(got - ref)
    rms = max(float(np.sqrt(np.mean(ref * ref))), 1e-300)
    big = np.abs(ref) >= 0.25 * rms
    return dict(max_rel=float(d.max() / max(np.abs(ref).max(), 1e-300)), max_over_rms=float(d.max() / rms),
                pure_rel_big=float((d[big] / np.abs(ref[big])).max()) if big.any() else 0.0, n=int(ref.size), n_big=int(big.sum()))


def oracle_rollout(prob, dt, actions, eps, p, det, obs_rows=None, it=0, **kw):
    o = _oracle(prob, dt)
    T = None
    if prob["cp"] is not None:
        T = oplanner.context_table_indexed(onets.context_forward(o["cp"], o["cp_obs"], o["cp_act"], o["st"]), it)
    return oplanner.rollout_indexed(o["env"], o["ff"], o["st"], o["obs"], T, actions.astype(dt), None if eps is None else eps.astype(dt),
                                    prob["E"], p, det, obs_rows=None if obs_rows is None else obs_rows.astype(dt), return_traj=True, **kw)


def measure(engines, cfgname="cfg2", seed=77, H=30, n_traj=24, hidden=200, n=None, mutate=None, obs_scale=None):
    """engines: dict name -> HipEngine factory(prob, p, H, det).  Returns {kernel name: {...}, "fp32_oracle": {...}}:
      one_step_*   every row of the FULL configuration advanced one teacher-forced step from its own random state
      traj_*       `n_traj` candidates x all particles over the whole horizon (next observation after every step)
    mutate(prob): applied to both synthetic problems (statistics / weights of the widened-envelope cases); obs_scale [D]: the
    observations (one-step rows and the trajectory's start) are multiplied by it (raw, un-normalised magnitudes)."""
    cfg = synth.CONFIGS[cfgname]
    E, p, n, det = cfg["E"], cfg["p"], n or cfg["n"], cfg["deterministic"]
    hs = (hidden,) * 4
    rng = np.random.default_rng(seed)
    out = {}
    # ---- one step, full size
    prob1 = synth.make_problem(env=cfg["env"], context=cfg["context"], E=E, m=1, H=1, trained_like=True, seed=seed, hidden_sizes=hs)
    if mutate is not None:
        mutate(prob1)
    D, A = prob1["D"], prob1["A"]
    obs_rows = rng.standard_normal((1, n, p, D))
    if obs_scale is not None:
        obs_rows = obs_rows * np.asarray(obs_scale)
        prob1["obs"] = prob1["obs"] * np.asarray(obs_scale)
    act1 = rng.uniform(-1, 1, (1, n, 1, A))
    eps1 = None if det else rng.standard_normal((1, 1, n, p, D))
    r64, t64 = oracle_rollout(prob1, np.float64, act1, eps1, p, det, obs_rows=obs_rows)
    r32, t32 = oracle_rollout(prob1, np.float32, act1, eps1, p, det, obs_rows=obs_rows)
    # (the step itself: next obs - obs is what the network computed; with raw-scale observations the sum is dominated by the exact addend)
    additive = cfg["env"] in ("slim_humanoid", "cartpole", "pendulum") and t64.size == obs_rows.size      # obs_postproc = obs + delta on every dim
    d64 = t64 - obs_rows.reshape(t64.shape) if additive else None
    out["fp32_oracle"] = {"one_step_obs": err_stats(t32, t64), "one_step_reward": err_stats(r32, r64)}
    got1 = {}
    for name, make in engines.items():
        eng = make(prob1, p, 1, det)
        ctx = eng.context_forward(prob1["cp_obs"], prob1["cp_act"]) if cfg["context"] else None
        rows, traj = eng.rollout_returns(prob1["obs"], ctx, act1, eps=eps1, obs_rows=obs_rows, want_traj=True)
        got1[name] = (rows.cpu().numpy(), traj.cpu().numpy())
        out[name] = {"one_step_obs": err_stats(got1[name][1], t64), "one_step_reward": err_stats(got1[name][0], r64),
                     "one_step_obs_vs_fp32_oracle": err_stats(got1[name][1], t32)}
        if additive:      # envs whose postproc is obs + delta on every dim compare the increment too
            out[name]["one_step_delta"] = err_stats(got1[name][1].astype(np.float64) - obs_rows.reshape(t64.shape), d64)
        eng.close()
    # ---- whole horizon, a slice of candidates
    probH = synth.make_problem(env=cfg["env"], context=cfg["context"], E=E, m=1, H=H, trained_like=True, seed=seed + 1, hidden_sizes=hs)
    if mutate is not None:
        mutate(probH)
    if obs_scale is not None:
        probH["obs"] = probH["obs"] * np.asarray(obs_scale)
    actH = rng.uniform(-1, 1, (1, n_traj, H, A))
    epsH = None if det else rng.standard_normal((H, 1, n_traj, p, D))
    R64, T64 = oracle_rollout(probH, np.float64, actH, epsH, p, det)
    R32, T32 = oracle_rollout(probH, np.float32, actH, epsH, p, det)
    out["fp32_oracle"].update({"traj_obs": err_stats(T32, T64), "traj_last_obs": err_stats(T32[-1], T64[-1]), "returns": err_stats(R32, R64)})
    for name, make in engines.items():
        eng = make(probH, p, H, det)
        ctx = eng.context_forward(probH["cp_obs"], probH["cp_act"]) if cfg["context"] else None
        rows, traj = eng.rollout_returns(probH["obs"], ctx, actH, eps=epsH, want_traj=True)
        rows, traj = rows.cpu().numpy(), traj.cpu().numpy()
        out[name].update({"traj_obs": err_stats(traj, T64), "traj_last_obs": err_stats(traj[-1], T64[-1]), "returns": err_stats(rows, R64)})
        eng.close()
    out["_meta"] = dict(config=cfgname, env=cfg["env"], E=E, p=p, n=n, H=H, hidden=hidden, rows_one_step=n * p, traj_candidates=n_traj,
                        weights="trained-like (synth.make_problem(trained_like=True))", seed=seed)
    return out


def product_engine(prob, p, H, det):
    return synth.make_engine(prob, p=p, H=H, deterministic=det)


def f32_engine(prob, p, H, det):
    from cadm_amd import _lib
    eng = synth.make_engine(prob, p=p, H=H, deterministic=det, lib=_lib.load_dev())
    eng.dev_set_rollout("f32")
    return eng


def markdown(res):
    m = res["_meta"]
    lines = ["# Split-f16 rollout kernel: measured precision -- %s, hidden %d" % (m["config"], m.get("hidden", 200)), "",
             "`%s` full size (%s, ens=%d part=%d cand=%d, hidden 4 x %d), %s.  Errors against the **fp64 numpy oracle** on identical inputs;"
             % (m["config"], m["env"], m["E"], m["p"], m["n"], m.get("hidden", 200), m["weights"]),
             "`xdl` = production kernel (3 f16 MFMA products of split operands, fp32 accumulate), `f32mfma` = the developer library's",
             "fp32-operand MFMA kernel, `fp32 oracle` = numpy float32 restatement.  one-step: %d rows, each from its own random state;"
             % m["rows_one_step"],
             "trajectory: %d candidates x %d particles x %d steps.  `max/max` = max|d| / max|ref|; `pure rel` = max elementwise"
             % (m["traj_candidates"], m["p"], m["H"]),
             "|d|/|ref| over elements with |ref| >= 0.25 rms (no floor).", "",
             "| quantity | metric | xdl | f32mfma | fp32 oracle |", "|---|---|---|---|---|"]
    names = [k for k in ("xdl", "f32mfma") if k in res]
    for q, label in (("one_step_obs", "one-step next obs"), ("one_step_reward", "one-step reward"), ("traj_obs", "%d-step trajectory" % m["H"]),
                     ("traj_last_obs", "obs after step %d" % m["H"]), ("returns", "%d-step returns" % m["H"])):
        for met, ml in (("max_rel", "max/max"), ("pure_rel_big", "pure rel")):
            cells = ["%.2e" % res[k][q][met] if k in res else "-" for k in ("xdl", "f32mfma", "fp32_oracle")]
            lines.append("| %s | %s | %s |" % (label, ml, " | ".join(cells)))
    if "xdl" in res:
        lines += ["", "one-step next obs, xdl vs the fp32 oracle: max/max %.2e, pure rel %.2e (the bar: 1e-5)."
                  % (res["xdl"]["one_step_obs_vs_fp32_oracle"]["max_rel"], res["xdl"]["one_step_obs_vs_fp32_oracle"]["pure_rel_big"])]
    return "\n".join(lines) + "\n"
