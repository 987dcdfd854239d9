"""Env plug-in surface of the hot path.

The reference model object only touches a handful of env attributes
(/root/reference/cadm/dynamics/mlp_cadm_ensemble_cem_dynamics.py:95-102,185-187,401-402,593):
``observation_space.shape[0]``, ``proc_observation_space_dims``, ``action_space.{shape,n}``,
``obs_preproc``, ``obs_postproc``, ``targ_proc``, ``tf_reward_fn()``.  A TF closure cannot
be traced here, so the env CLASS NAME selects a compiled-in env kind (SURVEY.md
Appendix B); unknown envs are rejected.

``EnvSpec`` objects are simulator-free stand-ins with the same duck type, used for
synthetic workloads and by ``fit`` for its host-side (numpy, float64) target
computation -- exactly what the reference does on the host (dynamics.py:399-407).
"""
import numpy as np

from ._lib import ENV_KINDS

# reference env class name -> kind name (cadm/envs/*.py class definitions)
CLASS_TO_KIND = {
    "HalfCheetahEnv": "halfcheetah",
    "CrippleHalfCheetahEnv": "cripple_halfcheetah",
    "AntEnv": "ant",
    "SlimHumanoidEnv": "slim_humanoid",
    "RandomCartPole_Force_Length": "cartpole",
    "ModifiableCartPoleEnv": "cartpole",
    "RandomPendulumAll": "pendulum",
    "ModifiablePendulumEnv": "pendulum",
}


class _Box:
    def __init__(self, dim):
        self.shape = (dim,)
        self.low = -np.ones(dim)
        self.high = np.ones(dim)


class _Discrete:
    def __init__(self, n):
        self.shape = ()
        self.n = n


class EnvSpec:
    """Simulator-free env stand-in: spaces + the four closures (numpy)."""

    def __init__(self, kind):
        if kind not in ENV_KINDS:
            raise ValueError("unknown env kind %r (supported: %s)" % (kind, sorted(ENV_KINDS)))
        self.kind = kind
        D, A, P, discrete = {
            "halfcheetah": (18, 6, 18, False), "cripple_halfcheetah": (18, 6, 18, False),
            "ant": (28, 8, 27, False), "slim_humanoid": (45, 17, 45, False),
            "cartpole": (4, 2, 4, True), "pendulum": (3, 1, 3, False)}[kind]
        self.observation_space = _Box(D)
        self.action_space = _Discrete(A) if discrete else _Box(A)
        self.proc_observation_space_dims = P
        self.cadm_env_kind = kind

    # --- closures (numpy, host side) ---
    def obs_preproc(self, obs):
        if self.kind in ("halfcheetah", "cripple_halfcheetah"):
            return np.concatenate([obs[..., 1:2], np.sin(obs[..., 2:3]), np.cos(obs[..., 2:3]),
                                   obs[..., 3:]], axis=-1)
        if self.kind == "ant":
            return obs[..., 1:]
        return obs

    def obs_postproc(self, obs, pred):
        if self.kind in ("halfcheetah", "cripple_halfcheetah", "ant"):
            return np.concatenate([pred[..., :1], obs[..., 1:] + pred[..., 1:]], axis=-1)
        return obs + pred

    def targ_proc(self, obs, next_obs):
        if self.kind in ("halfcheetah", "cripple_halfcheetah", "ant"):
            return np.concatenate([next_obs[..., :1], next_obs[..., 1:] - obs[..., 1:]], axis=-1)
        return next_obs - obs


def make_env_spec(kind):
    return EnvSpec(kind)


def resolve_env_kind(env):
    """Map an env object (reference env, NormalizedEnv wrapper, or EnvSpec) to a kind name."""
    seen = 0
    e = env
    while e is not None and seen < 8:
        kind = getattr(e, "cadm_env_kind", None)
        if isinstance(kind, str):
            return kind
        for cls in type(e).__mro__:
            if cls.__name__ in CLASS_TO_KIND:
                return CLASS_TO_KIND[cls.__name__]
        nxt = None
        for attr in ("wrapped_env", "_wrapped_env", "env", "unwrapped"):
            cand = e.__dict__.get(attr) if hasattr(e, "__dict__") else None
            if cand is not None and cand is not e:
                nxt = cand
                break
        e = nxt
        seen += 1
    raise ValueError(
        "cannot map env %r to a compiled-in env kind; supported reference env classes: %s"
        % (type(env).__name__, sorted(CLASS_TO_KIND)))
