"""Deterministic INPUTS of the graph-wiring golden (shared by tests/golden/make_graph_golden.py, which feeds them to the
reference's graph builder, and by the tests, which feed the same numbers to the oracle and to the HIP path).
Everything here is input data: weights in variable-creation order, statistics, observations, and the random draws in the
order and shapes the reference's graph consumes them."""
import numpy as np

CASES = {   # name: dict(env, E, p, m, n, H, hidden, cp_hidden, C, Hh, seed)
    "hc_cadm_m2": dict(env="halfcheetah", D=18, A=6, P=18, E=5, p=10, m=2, n=56, H=3, hidden=(128,) * 4, cp_hidden=(24, 16, 8),
                       C=10, Hh=3, B=2, seed=101),
    "hc_cadm_m3": dict(env="halfcheetah", D=18, A=6, P=18, E=5, p=5, m=3, n=52, H=4, hidden=(128,) * 4, cp_hidden=(16, 8),
                       C=10, Hh=2, B=2, seed=202),
    # BASELINE.json configs[1] at FULL geometry -- the shape bench.py times (run_scripts/run_cadm_pets.py:118-140,201 defaults:
    # hidden 200 x 4, context encoder 256/128/64 -> 10, history 10, 20 particles, 200 candidates, horizon 30) -- and its twin at the
    # production launch shape m = 10 (10 envs planned for in one call; odd CEM iterations then take quirk Q2's scrambled context)
    "hc_cadm_cfg2": dict(env="halfcheetah", D=18, A=6, P=18, E=5, p=20, m=1, n=200, H=30, hidden=(200,) * 4, cp_hidden=(256, 128, 64),
                         C=10, Hh=10, B=2, seed=2201),
    "hc_cadm_cfg2_m10": dict(env="halfcheetah", D=18, A=6, P=18, E=5, p=20, m=10, n=200, H=30, hidden=(200,) * 4,
                             cp_hidden=(256, 128, 64), C=10, Hh=10, B=2, seed=2210),
    # random shooting (core/utils.py:490-561): the planner branch taken when no CEM initial distribution is fed
    # the vanilla PE-TS twin (create_plus_ensemble_cem_mlp, core/utils.py:5-248): no context encoder
    "hc_vanilla_m2": dict(env="halfcheetah", D=18, A=6, P=18, E=5, p=10, m=2, n=54, H=3, hidden=(128,) * 4, cp_hidden=(), C=0, Hh=2,
                          B=2, seed=606, vanilla=True),
    # the other env kinds' closures (obs_preproc / obs_postproc / tf_reward_fn of cadm/envs/*.py) inside the same graph
    "ant_cadm": dict(env="ant", D=28, A=8, P=27, E=5, p=5, m=2, n=52, H=3, hidden=(128,) * 4, cp_hidden=(16, 8), C=10, Hh=2, B=2, seed=811),
    "humanoid_cadm": dict(env="slim_humanoid", D=45, A=17, P=45, E=5, p=5, m=1, n=52, H=2, hidden=(128,) * 4, cp_hidden=(16, 8), C=10,
                          Hh=2, B=2, seed=812),
    "cripple_cadm": dict(env="cripple_halfcheetah", D=18, A=6, P=18, E=5, p=5, m=2, n=52, H=3, hidden=(128,) * 4, cp_hidden=(16, 8),
                         C=10, Hh=2, B=2, seed=813),
    "pendulum_cadm": dict(env="pendulum", D=3, A=1, P=3, E=5, p=5, m=2, n=52, H=4, hidden=(128,) * 4, cp_hidden=(16, 8), C=10, Hh=2,
                          B=2, seed=814),
    "cartpole_rs": dict(env="cartpole", D=4, A=2, P=4, E=5, p=5, m=2, n=40, H=4, hidden=(128,) * 4, cp_hidden=(16, 8), C=10, Hh=2,
                        B=2, seed=815, rs=True, discrete=True),
    "hc_cadm_rs": dict(env="halfcheetah", D=18, A=6, P=18, E=5, p=10, m=2, n=40, H=3, hidden=(128,) * 4, cp_hidden=(16, 8),
                       C=10, Hh=2, B=2, seed=505, rs=True),
}


def trunc_normal(rng, shape):
    z = rng.standard_normal(shape)
    bad = np.abs(z) >= 2.0
    while bad.any():
        z[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(z) >= 2.0
    return z


class Draws:
    """The graph's random ops, in call order: standard (truncated) normals as float32."""

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed + 7)
        self.log = []

    def truncated(self, shape):
        z = trunc_normal(self.rng, tuple(int(s) for s in shape)).astype(np.float32)
        self.log.append(("truncated_normal", z))
        return z

    def normal(self, shape):
        z = self.rng.standard_normal(tuple(int(s) for s in shape)).astype(np.float32)
        self.log.append(("normal", z))
        return z

    def randint(self, shape, n):
        k = self.rng.integers(0, int(n), tuple(int(s) for s in shape)).astype(np.int32)
        self.log.append(("randint", k))
        return k

    def uniform(self, shape, lo, hi):
        u = self.rng.uniform(lo, hi, tuple(int(s) for s in shape)).astype(np.float32)
        self.log.append(("uniform", u))
        return u


class Weights:
    """tf.compat.v1.get_variable / tf.Variable in creation order.  Initial values are 'trained-like' (gain 2, small random
    biases) so that biases and head scales matter; the reference's initialisers only decide SHAPES here."""

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.vars = []      # (name, array)

    def dense(self, name, shape):
        if name.endswith("_weight"):
            v = trunc_normal(self.rng, shape) * (2.0 / (2.0 * np.sqrt(shape[1])))
        else:
            v = 0.1 * self.rng.standard_normal(shape)
        v = v.astype(np.float32)
        self.vars.append((name, v))
        return v

    def plain(self, name, value):
        v = np.asarray(value, np.float32)
        self.vars.append((name, v))
        return v


def make_inputs(case):
    c = CASES[case]
    rng = np.random.default_rng(c["seed"] + 1)
    D, A, P, Hh, m, H, E, B = c["D"], c["A"], c["P"], c["Hh"], c["m"], c["H"], c["E"], c["B"]
    f32 = lambda x: np.asarray(x, np.float32)
    ms = lambda k: (f32(rng.standard_normal(k)), f32(rng.uniform(0.5, 2.0, k)))
    st = {}
    st["obs_mean"], st["obs_std"] = ms(P)
    st["act_mean"], st["act_std"] = ms(A)
    st["delta_mean"], st["delta_std"] = ms(D)
    st["cp_obs_mean"], st["cp_obs_std"] = f32(np.zeros(D * Hh)), f32(np.ones(D * Hh))     # state_diff (dynamics.py:616-618)
    st["cp_act_mean"], st["cp_act_std"] = ms(A * Hh)
    st["back_delta_mean"], st["back_delta_std"] = ms(D)
    return dict(stats=st, obs=f32(rng.standard_normal((m, D))), cp_obs=f32(0.1 * rng.standard_normal((m, D * Hh))),
                cp_act=f32(rng.uniform(-1, 1, (m, A * Hh))), init_mean=f32(np.zeros((m, H, A))), init_var=f32(np.full((m, H, A), 0.25)),
                bs_obs=f32(rng.standard_normal((E, B, D))), bs_act=f32(rng.uniform(-1, 1, (E, B, A))),
                bs_cp_obs=f32(0.1 * rng.standard_normal((E, B, D * Hh))), bs_cp_act=f32(rng.uniform(-1, 1, (E, B, A * Hh))))


# ---- the training-loss golden (tests/golden/make_loss_golden.py): the reference model's constructor on a bootstrap batch ----
LOSS_CASES = {
    "hc_cadm_prob": dict(env="halfcheetah", D=18, A=6, P=18, E=5, p=5, m=1, n=52, H=2, hidden=(128,) * 4, cp_hidden=(24, 16, 8), C=10,
                         Hh=3, F=2, B=6, seed=303, deterministic=False, back_coeff=0.5, weight_decay_coeff=1.0,
                         weight_decays=(0.000025, 0.00005, 0.000075, 0.000075, 0.0001),
                         context_weight_decays=(0.000025, 0.00005, 0.000075, 0.000075)),
    # the vanilla PE-TS class (cadm/dynamics/mlp_ensemble_cem_dynamics.py): no context, no backward model
    "hc_vanilla_prob": dict(env="halfcheetah", D=18, A=6, P=18, E=5, p=5, m=1, n=52, H=2, hidden=(128,) * 4, cp_hidden=(), C=0,
                            Hh=1, F=1, B=6, seed=707, deterministic=False, back_coeff=0.0, weight_decay_coeff=1.0, vanilla=True,
                            weight_decays=(0.000025, 0.00005, 0.000075, 0.000075, 0.0001), context_weight_decays=()),
    # an env whose obs_preproc drops a dimension (ant: P = D - 1) and one with many dims, through the training graph
    "ant_cadm_prob": dict(env="ant", D=28, A=8, P=27, E=5, p=5, m=1, n=52, H=2, hidden=(128,) * 4, cp_hidden=(16, 8), C=10, Hh=2,
                          F=2, B=5, seed=1010, deterministic=False, back_coeff=0.5, weight_decay_coeff=1.0,
                          weight_decays=(0.000025, 0.00005, 0.000075, 0.000075, 0.0001), context_weight_decays=(0.000025, 0.00005, 0.000075)),
    "humanoid_cadm_prob": dict(env="slim_humanoid", D=45, A=17, P=45, E=5, p=5, m=1, n=52, H=2, hidden=(128,) * 4, cp_hidden=(16, 8),
                               C=10, Hh=2, F=2, B=5, seed=1111, deterministic=False, back_coeff=0.0, weight_decay_coeff=1.0,
                               weight_decays=(0.000025, 0.00005, 0.000075, 0.000075, 0.0001),
                               context_weight_decays=(0.000025, 0.00005, 0.000075)),
    # one member, so that the checkpoint the reference's own save() writes for it stays small (tests/golden/ref_ckpt/)
    "ckpt_e1": dict(env="halfcheetah", D=18, A=6, P=18, E=1, p=1, m=1, n=52, H=2, hidden=(128,) * 4, cp_hidden=(8,), C=10, Hh=2, F=2,
                    B=4, seed=909, deterministic=False, back_coeff=0.5, weight_decay_coeff=1.0, save=True,
                    weight_decays=(0.000025, 0.00005, 0.000075, 0.000075, 0.0001), context_weight_decays=(0.000025, 0.00005)),
    "hc_cadm_det": dict(env="halfcheetah", D=18, A=6, P=18, E=5, p=5, m=1, n=52, H=2, hidden=(128,) * 4, cp_hidden=(16, 8), C=10,
                        Hh=2, F=2, B=5, seed=404, deterministic=True, back_coeff=0.5, weight_decay_coeff=0.5,
                        weight_decays=(0.000025, 0.00005, 0.000075, 0.000075, 0.0001),
                        context_weight_decays=(0.000025, 0.00005, 0.000075)),
}


def make_loss_inputs(case):
    c = LOSS_CASES[case]
    rng = np.random.default_rng(c["seed"] + 1)
    D, A, P, Hh, m, H, E, B = c["D"], c["A"], c["P"], c["Hh"], c["m"], c["H"], c["E"], c["B"]
    f32 = lambda x: np.asarray(x, np.float32)
    ms = lambda k: (f32(rng.standard_normal(k)), f32(rng.uniform(0.5, 2.0, k)))
    st = {}
    st["obs_mean"], st["obs_std"] = ms(P)
    st["act_mean"], st["act_std"] = ms(A)
    st["delta_mean"], st["delta_std"] = ms(D)
    st["cp_obs_mean"], st["cp_obs_std"] = f32(np.zeros(D * Hh)), f32(np.ones(D * Hh))
    st["cp_act_mean"], st["cp_act_std"] = ms(A * Hh)
    st["back_delta_mean"], st["back_delta_std"] = ms(D)
    return dict(stats=st, obs=f32(rng.standard_normal((m, D))), obs_next=f32(rng.standard_normal((m, D))),
                act=f32(rng.uniform(-1, 1, (m, A))), cp_obs=f32(0.1 * rng.standard_normal((m, D * Hh))),
                cp_act=f32(rng.uniform(-1, 1, (m, A * Hh))), init_mean=f32(np.zeros((m, H, A))), init_var=f32(np.full((m, H, A), 0.25)),
                bs_obs=f32(rng.standard_normal((E, B, D))), bs_obs_next=f32(rng.standard_normal((E, B, D))),
                bs_act=f32(rng.uniform(-1, 1, (E, B, A))), bs_delta=f32(rng.standard_normal((E, B, D))),
                bs_back_delta=f32(rng.standard_normal((E, B, D))), bs_cp_obs=f32(0.1 * rng.standard_normal((E, B, D * Hh))),
                bs_cp_act=f32(rng.uniform(-1, 1, (E, B, A * Hh))))


def placeholder_feed(inp, vanilla=False):
    """The constructor's placeholders in creation order (mlp_cadm_ensemble_cem_dynamics.py:108-137; vanilla
    mlp_ensemble_cem_dynamics.py:91-107)."""
    st = inp["stats"]
    if vanilla:
        return [inp["obs"], inp["act"], inp["bs_delta"][0], inp["bs_obs"], inp["bs_act"], inp["bs_delta"],
                st["obs_mean"], st["obs_std"], st["act_mean"], st["act_std"], st["delta_mean"], st["delta_std"],
                inp["init_mean"], inp["init_var"]]
    return [inp["obs"], inp["obs_next"], inp["act"], inp["cp_obs"], inp["cp_act"],
            inp["bs_obs"], inp["bs_obs_next"], inp["bs_act"], inp["bs_delta"], inp["bs_back_delta"], inp["bs_cp_obs"], inp["bs_cp_act"],
            st["obs_mean"], st["obs_std"], st["act_mean"], st["act_std"], st["delta_mean"], st["delta_std"],
            st["cp_obs_mean"], st["cp_obs_std"], st["cp_act_mean"], st["cp_act_std"], st["back_delta_mean"], st["back_delta_std"],
            inp["init_mean"], inp["init_var"]]


def make_fit_inputs(case, N=7):
    """Windowed samples as `fit` receives them (dynamics.py:382-390): [N, F, .] future windows, [N, Hh * .] histories, mask."""
    c = LOSS_CASES[case]
    rng = np.random.default_rng(c["seed"] + 2)
    D, A, Hh, F = c["D"], c["A"], c["Hh"], c["F"]
    fb = (rng.uniform(size=(N, F)) > 0.3).astype(np.float64)
    fb[:, 0] = 1.0
    return dict(obs=rng.standard_normal((N, F, D)), act=rng.uniform(-1, 1, (N, F, A)), delta=rng.standard_normal((N, F, D)),
                cp_obs=0.1 * rng.standard_normal((N, D * Hh)), cp_act=rng.uniform(-1, 1, (N, A * Hh)), future_bool=fb,
                obs_next=rng.standard_normal((N, F, D)), back_delta=rng.standard_normal((N, F, D)))
