"""The planner graph's WIRING pinned to the reference's own graph-building code: tests/golden/graph_golden.npz was written
by /root/reference/cadm/dynamics/core/utils.py itself, imported unchanged and executed on a numpy-eager `tensorflow`
stand-in (tests/golden/make_graph_golden.py says exactly what that does and does not prove).  Inputs (weights in
variable-creation order, statistics, observations, random draws in the order the graph consumes them) are regenerated from
tests/golden/graph_inputs.py; the oracle -- both formulations -- and the HIP planner must reproduce the reference's plan."""
import os
import sys
from collections import OrderedDict

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import graph_inputs as gi  # noqa: E402
from oracle import envs as oenvs  # noqa: E402
from oracle import nets as onets  # noqa: E402
from oracle import planner as oplanner  # noqa: E402

GOLD = np.load(os.path.join(HERE, "golden", "graph_golden.npz"))


def rebuild(case):
    """Inputs exactly as the generator fed them to the reference: (cp params, ff params, stats/inputs, z, eps_canonical)."""
    c = gi.CASES[case]
    W = gi.Weights(c["seed"])
    nets = {"context_model": OrderedDict(), "ff_model": OrderedDict()}
    for name, shp in zip(GOLD[case + "/var_names"], GOLD[case + "/var_shapes"]):
        net, pname = str(name).split("/")
        shape = tuple(int(s) for s in str(shp).split(","))
        if pname.endswith(("_weight", "_bias")):
            nets[net][pname] = W.dense(str(name), shape)
        else:       # tf.Variable(np.ones([1, D]) / 2.) / tf.Variable(-np.ones([1, D]) * 10)  (core/utils.py:338-339; :70-71 vanilla)
            key = {"max_log_var": "max_logvar", "min_log_var": "min_logvar"}.get(pname, pname)      # the vanilla builder's spelling
            nets[net][key] = W.plain(str(name), np.ones(shape) / 2.0 if key == "max_logvar" else -np.ones(shape) * 10)
    draws = gi.Draws(c["seed"])
    seq = []
    for kind, shp in zip(GOLD[case + "/draw_kinds"], GOLD[case + "/draw_shapes"]):
        shape = tuple(int(s) for s in str(shp).split(","))
        seq.append((str(kind), draws.truncated(shape) if kind == "truncated_normal" else draws.uniform(shape, -1, 1) if kind == "uniform"
                    else draws.randint(shape, c["A"]) if kind == "randint" else draws.normal(shape)))
    E, p, m, n, H, D = c["E"], c["p"], c["m"], c["n"], c["H"], c["D"]
    train_eps = seq[0][1]
    if c.get("rs"):       # random shooting: one uniform action tensor, then the head noise of every step
        assert [k for k, _ in seq] == ["normal", "randint" if c.get("discrete") else "uniform"] + ["normal"] * H
        eps = np.stack([e.reshape(p, m, n, D).transpose(1, 2, 0, 3) for _, e in seq[2:]])        # [H, m, n, p, D]
        return c, nets, gi.make_inputs(case), seq[1][1], eps, train_eps
    assert [k for k, _ in seq] == ["normal"] + (["truncated_normal"] + ["normal"] * H) * 5     # the graph's draw order
    z, eps = [], []
    for it in range(5):
        blk = seq[1 + it * (H + 1):1 + (it + 1) * (H + 1)]
        z.append(blk[0][1])                                                       # [m, n, H, A]
        # the graph draws the head noise in its member-major layout [E, (p/E)*m*n, D] = reshape of [p, m, n, D]
        eps.append(np.stack([e.reshape(p, m, n, D).transpose(1, 2, 0, 3) for _, e in blk[1:]]))   # [H, m, n, p, D]
    return c, nets, gi.make_inputs(case), np.stack(z), np.stack(eps), train_eps


def test_variable_creation_order_is_the_checkpoint_order():
    """save / load rely on tf.trainable_variables() = creation order (dynamics.py:266,571-588): context_model first, and
    inside a dynamics net hidden_i_{weight,bias}, output_mu_*, output_logvar_*, max_logvar, min_logvar."""
    from cadm_amd.engine import ctx_param_names, dyn_param_names
    names = [str(x) for x in GOLD["hc_cadm_m2/var_names"]]
    want = ["context_model/" + x for x in ctx_param_names(3)] + ["ff_model/" + x for x in dyn_param_names(4)]
    assert names == want


@pytest.mark.parametrize("case", sorted(gi.CASES))
def test_oracle_reproduces_the_reference_graph(case):
    c, nets, inp, z, eps, train_eps = rebuild(case)
    env, st = oenvs.make_env(c["env"]), inp["stats"]
    cp, ff = nets["context_model"] or None, nets["ff_model"]
    feats = [onets.normalize(env.obs_preproc(inp["bs_obs"]), st["obs_mean"], st["obs_std"]),
             onets.normalize(inp["bs_act"], st["act_mean"], st["act_std"])]
    if cp is not None:
        ctx = onets.context_forward(cp, inp["cp_obs"], inp["cp_act"], st)
        np.testing.assert_allclose(ctx, GOLD[case + "/context"], rtol=1e-5, atol=1e-6)
        # training-graph forward on the [E, B, .] batch (core/utils.py:372-381): heads before the noise
        bs_cp = onets.context_forward_bs(cp, inp["bs_cp_obs"], inp["bs_cp_act"], st)
        np.testing.assert_allclose(bs_cp, GOLD[case + "/bs_cp"], rtol=1e-5, atol=1e-6)
        feats.append(bs_cp)
    x = np.concatenate(feats, axis=-1)
    out, mu, lv = onets.dynamics_forward(ff, x, st["delta_mean"], st["delta_std"], train_eps, False)
    np.testing.assert_allclose(mu, GOLD[case + "/train_mu"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(lv, GOLD[case + "/train_logvar"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out, GOLD[case + "/train_output"], rtol=1e-5, atol=1e-5)
    if c.get("rs"):      # the random-shooting planner (core/utils.py:490-561): z = the uniform action tensor [m, n, H, A]
        for form in ("literal", "indexed"):
            if c.get("discrete"):     # z = action indices [m, n, H]; the net sees their one-hot rows un-normalised (core/utils.py:499-500)
                first, _ = oplanner.rs_plan(env, ff, cp, st, inp["obs"], inp["cp_obs"], inp["cp_act"], np.eye(c["A"], dtype=np.float32)[z],
                                            eps, c["E"], c["p"], formulation=form, raw_actions=z)
            else:
                first, _ = oplanner.rs_plan(env, ff, cp, st, inp["obs"], inp["cp_obs"], inp["cp_act"], z, eps, c["E"], c["p"],
                                            formulation=form)
            np.testing.assert_array_equal(first, GOLD[case + "/plan"], err_msg=form)
        return
    # the unrolled CEM planner, both formulations of the oracle (literal tile/transpose/reshape chain and index-mapped)
    big = c["m"] * c["n"] * c["p"] * c["H"] > 500000          # the m = 10 twin of cfg2: the literal chain once is enough on CPU
    for form in (("indexed",) if big else ("literal", "indexed")):
        plan, info, _ = oplanner.cem_plan(env, ff, cp, st, inp["obs"], inp["cp_obs"] if cp is not None else None,
                                          inp["cp_act"] if cp is not None else None, inp["init_mean"], inp["init_var"], z, eps,
                                          c["E"], c["p"], formulation=form, return_info=True)
        check_iterations(case, [i["cand_returns"] for i in info], [i["elites"] for i in info], 1e-5, form)
        err = np.abs(plan - GOLD[case + "/plan"]).max()
        assert err <= 1e-5, "%s formulation: plan differs from the reference graph's by %.2e" % (form, err)


def check_iterations(case, cand_returns, elites, rtol, who):
    """Per CEM iteration, against what the reference's own lines ranked (the input of its tf.nn.top_k, core/utils.py:474-475) and
    kept: candidate returns within rtol of the iteration's return scale, and the SAME 50 elites (as sets: the refit is a mean)."""
    for it, (cand, el) in enumerate(zip(cand_returns, elites)):
        ref, ref_el = GOLD[case + "/cand_returns"][it], GOLD[case + "/elites"][it]
        scale = np.abs(ref).max()
        err = np.abs(np.asarray(cand).reshape(ref.shape) - ref).max() / scale
        assert err <= rtol, "%s: candidate returns of CEM iteration %d differ from the reference graph's by %.2e of their scale" % (who, it, err)
        np.testing.assert_array_equal(np.sort(np.asarray(el).reshape(ref_el.shape), axis=-1), np.sort(ref_el, axis=-1),
                                      err_msg="%s: elite set of CEM iteration %d" % (who, it))
    return err


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(gi.CASES))
def test_hip_planner_reproduces_the_reference_graph(gpu, case):
    _hip_case(case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["hc_cadm_cfg2", "hc_cadm_cfg2_m10"])
@pytest.mark.parametrize("flavour", [2, 3])
def test_hip_planner_reproduces_the_reference_graph_on_every_rollout_flavour(gpu, case, flavour):
    """The launcher picks the large-batch flavours of the rollout kernel (two row tiles per workgroup; the wave-tile kernel) only
    from sizes no reference-executed golden has; forced here (developer library) on the timed geometry's goldens."""
    from cadm_amd import _lib
    _hip_case(case, lib=_lib.load_dev(), flavour=flavour)


def _hip_case(case, lib=None, flavour=0):
    from cadm_amd import planner as hplanner
    from cadm_amd.engine import HipEngine
    c, nets, inp, z, eps, _ = rebuild(case)
    kw = {} if lib is None else {"lib": lib}
    eng = HipEngine(c["env"], c["E"], c["p"], c["D"], c["A"], c["P"], c["C"], c["hidden"], c["H"], history_length=c["Hh"],
                    cp_hidden_sizes=c["cp_hidden"], discrete=bool(c.get("discrete")), **kw)
    if flavour:
        eng.dev_set_rollout("xdl", row_tiles=flavour)
    vanilla = c["C"] == 0
    if not vanilla:
        eng.set_net("context_model", nets["context_model"])
    eng.set_net("ff_model", nets["ff_model"])
    eng.set_stats(inp["stats"])
    if vanilla:
        inp = dict(inp, cp_obs=None, cp_act=None)
    else:
        ctx = eng.context_forward(inp["cp_obs"], inp["cp_act"]).cpu().numpy()
        np.testing.assert_allclose(ctx, GOLD[case + "/context"], rtol=2e-5, atol=2e-6)
    if c.get("rs"):
        if c.get("discrete"):
            first, _ = hplanner.rs_plan(eng, inp["obs"], inp["cp_obs"], inp["cp_act"], c["n"], actions=np.eye(c["A"], dtype=np.float32)[z],
                                        raw=z, eps=eng._t(eps))
            np.testing.assert_array_equal(first.cpu().numpy(), GOLD[case + "/plan"])                      # the chosen action index
        else:
            first, _ = hplanner.rs_plan(eng, inp["obs"], inp["cp_obs"], inp["cp_act"], c["n"], actions=z, eps=eng._t(eps))
            np.testing.assert_array_equal(first.cpu().numpy(), np.clip(GOLD[case + "/plan"], -1.0, 1.0))  # the chosen candidate's action
        eng.close()
        return
    plan, info, _ = hplanner.cem_plan(eng, inp["obs"], inp["cp_obs"], inp["cp_act"], inp["init_mean"], inp["init_var"], c["n"],
                                      z=eng._t(z), eps=eng._t(eps), return_info=True)
    plan = plan.cpu().numpy()
    # every iteration: the candidate returns the reference's own graph code ranked (its top_k input) to 1e-5 of their scale --
    # the rollout kernel over the full horizon, context quirks included -- and the same 50 elites; then the plan itself.  With
    # identical elite sets the plan is a mean of identical injected draws: what remains is the refit's summation order.
    err = check_iterations(case, [i["cand"].cpu().numpy() for i in info], [i["elites"].cpu().numpy() for i in info], 1e-5, "hip")
    ref = np.clip(GOLD[case + "/plan"], -1.0, 1.0)            # get_action's clip (dynamics.py:365-366)
    perr = np.abs(plan - ref).max()
    print("%s: hip vs reference graph: last-iteration returns %.2e of scale, plan max abs err %.2e" % (case, err, perr))
    assert perr <= 1e-5, perr
    eng.close()
