"""Oracle restatement of the reference's unrolled CEM / random-shooting planner.

Test infrastructure only (see oracle/__init__.py).

Follows /root/reference/cadm/dynamics/core/utils.py:
  CEM block  :411-488  (vanilla twin :118-184)
  RS block   :490-561  (vanilla :186-246)
and /root/reference/cadm/dynamics/mlp_cadm_ensemble_cem_dynamics.py:344-367
(get_action: clip to [-1,1]).

Two independent formulations of the rollout are provided:
  * ``literal``: the tile / transpose([2,0,1,3]) / reshape chain exactly as the
    graph builds it (utils.py:432-472), including the per-iteration re-transpose
    of the context tensor (quirk Q2, :434) and the j % E context tiling (Q1, :435);
  * ``indexed``: per-row index arithmetic (row (mi,ni,j) -> member j // (p/E),
    context T[mi, j % E]) -- the form the HIP kernel implements.
tests/test_oracle_planner.py asserts they agree bit-for-bit in fp64.

All randomness is INJECTED: ``z`` are standard truncated-normal draws
(|z| < 2) for the action sampler (utils.py:429), ``eps`` standard normals for the
Gaussian head (utils.py:365), both in canonical layouts
  z   [iters, m, n, H, A]        eps [iters, H, m, n, p, D].
"""
import numpy as np

from . import nets

NUM_ELITES = 50      # utils.py:391
NUM_CEM_ITERS = 5    # utils.py:392
ALPHA = 0.1          # utils.py:393
LOWER, UPPER = -1.0, 1.0  # utils.py:395-396


# ----------------------------------------------------------------------------
# context layout (quirks Q1 / Q2)
# ----------------------------------------------------------------------------
def context_table_literal(ctx, n_iters):
    """Yield T[it] of shape [m,E,C] exactly as utils.py:434-435 produces it:
    the python variable is re-transposed on every CEM iteration."""
    cur = ctx  # [E,m,C]
    out = []
    E, m, C = ctx.shape
    for _ in range(n_iters):
        cur = np.transpose(cur, (1, 0, 2))          # :434
        out.append(np.reshape(cur, (m, E, C)).copy())  # the [m,1,E,C] reshape of :435
    return out


def context_table_indexed(ctx, it, quirks=True):
    """T [m,E',C] with context[mi,ni,j] = T[mi, j % E]."""
    E, m, C = ctx.shape
    if not quirks:
        raise ValueError("quirks-off layout is per-member, use context_rows_fixed")
    T = np.empty((m, E, C), ctx.dtype)
    for mi in range(m):
        for e in range(E):
            if it % 2 == 0:
                T[mi, e] = ctx[e, mi]
            else:
                f = mi * E + e
                T[mi, e] = ctx[f // m, f % m]
    return T


def member_of_particle(j, p, E):
    return j // (p // E)  # utils.py:445-455 via reshape [E, (p/E)*m*n, .]


# ----------------------------------------------------------------------------
# one trajectory-sampling rollout over the horizon
# ----------------------------------------------------------------------------
def _x_rows(env, st, observation, action_t, p):
    """normalised [m,n,p,P] obs features and [m,n,p,A] action features."""
    nact = nets.normalize(action_t, st["act_mean"], st["act_std"])           # :443
    nact = np.tile(nact[:, :, None, :], (1, 1, p, 1))                         # :444
    nobs = nets.normalize(env.obs_preproc(observation), st["obs_mean"], st["obs_std"])  # :450-451
    return nobs, nact


def rollout_literal(env, dyn, st, obs, T, actions, eps, E, p, deterministic, norm_actions=True,
                    raw_actions=None, return_traj=False):
    """utils.py:431-472 transcribed.  obs [m,D]; T [m,E,C] or None; actions [m,n,H,A];
    eps [H,m,n,p,D].  Returns returns [m,n,p] (before the particle mean)."""
    dt = obs.dtype.type
    m, D = obs.shape
    _, n, H, A = actions.shape
    pe = p // E
    observation = np.tile(obs.reshape(m, 1, 1, D), (1, n, p, 1))              # :432
    returns = np.zeros((m, n, p), obs.dtype)
    if T is not None:
        C = T.shape[-1]
        context = np.tile(T.reshape(m, 1, E, C), (1, n, pe, 1))               # :435
        reshaped_context = np.transpose(context, (2, 0, 1, 3)).reshape(E, pe * m * n, C)  # :436-439
    traj = []
    for t in range(H):
        action = actions[:, :, t]                                             # :442
        if norm_actions:
            nact = nets.normalize(action, st["act_mean"], st["act_std"])      # :443
        else:
            nact = action
        nact = np.tile(nact[:, :, None, :], (1, 1, p, 1))                     # :444
        nact = np.transpose(nact, (2, 0, 1, 3)).reshape(E, pe * m * n, A)     # :445-448
        nobs = nets.normalize(env.obs_preproc(observation), st["obs_mean"], st["obs_std"])
        P = nobs.shape[-1]
        nobs = np.transpose(nobs, (2, 0, 1, 3)).reshape(E, pe * m * n, P)     # :452-455
        if T is not None:
            x = np.concatenate([nobs, nact, reshaped_context], axis=2)        # :458
        else:
            x = np.concatenate([nobs, nact], axis=2)                          # :460
        e_t = np.transpose(eps[t], (2, 0, 1, 3)).reshape(E, pe * m * n, D)
        delta, _, _ = nets.dynamics_forward(dyn, x, st["delta_mean"], st["delta_std"], e_t, deterministic)
        delta = delta.reshape(p, m, n, D)                                     # :463
        delta = np.transpose(delta, (1, 2, 0, 3))                             # :464
        next_observation = env.obs_postproc(observation, delta)               # :466
        if raw_actions is not None:   # discrete RS: reward gets the integer action (:546)
            repeated_action = np.tile(raw_actions[:, :, t][:, :, None], (1, 1, p))
        else:
            repeated_action = np.tile(action[:, :, None, :], (1, 1, p, 1))    # :467
        reward = env.reward(observation, repeated_action, next_observation)   # :469
        returns = returns + reward.astype(obs.dtype)                          # :471
        observation = next_observation                                        # :472
        if return_traj:
            traj.append(observation.copy())
    if return_traj:
        return returns, np.stack(traj)
    return returns


def rollout_indexed(env, dyn, st, obs, T, actions, eps, E, p, deterministic, norm_actions=True,
                    raw_actions=None, return_traj=False, obs_rows=None, ctx_rows=None,
                    hidden_act=nets.swish, out_act=nets.ACTIVATIONS[None], x_transform=None):
    """Index-mapped form: row (mi,ni,j) is evaluated by member j // (p/E) with
    context T[mi, j % E] (or an explicit ``ctx_rows`` [m,p,C]).  ``obs_rows``
    [m,n,p,D] optionally overrides the tiled start state (teacher forcing).
    ``hidden_act`` / ``out_act``: the ctor's nonlinearities (dynamics.py:17-24,104-105).  ``x_transform``
    (tests of the kernels' documented range clamps only) is applied to the assembled network input."""
    m, D = obs.shape
    _, n, H, A = actions.shape
    pe = p // E
    observation = (np.broadcast_to(obs[:, None, None, :], (m, n, p, D)).copy()
                   if obs_rows is None else obs_rows.copy())
    returns = np.zeros((m, n, p), obs.dtype)
    crow = None
    if ctx_rows is not None:
        crow = np.broadcast_to(ctx_rows[:, None, :, :], (m, n, p, ctx_rows.shape[-1]))
    elif T is not None:
        C = T.shape[-1]
        crow = np.empty((m, n, p, C), obs.dtype)
        for j in range(p):
            crow[:, :, j, :] = T[:, j % E, :][:, None, :]
    traj = []
    for t in range(H):
        action = actions[:, :, t]
        nact = nets.normalize(action, st["act_mean"], st["act_std"]) if norm_actions else action
        nact = np.broadcast_to(nact[:, :, None, :], (m, n, p, A))
        nobs = nets.normalize(env.obs_preproc(observation), st["obs_mean"], st["obs_std"])
        feats = [nobs, nact] + ([crow] if crow is not None else [])
        x = np.concatenate(feats, axis=-1)                                   # [m,n,p,K0]
        if x_transform is not None:
            x = x_transform(x)
        delta = np.empty((m, n, p, D), obs.dtype)
        for e in range(E):
            js = slice(e * pe, (e + 1) * pe)
            xe = x[:, :, js, :].reshape(1, -1, x.shape[-1])
            ee = eps[t][:, :, js, :].reshape(1, -1, D)
            dyn_e = {k: (v[e:e + 1] if v.ndim == 3 else v) for k, v in dyn.items()}
            d_e, _, _ = nets.dynamics_forward(dyn_e, xe, st["delta_mean"], st["delta_std"], ee, deterministic,
                                              hidden_act=hidden_act, out_act=out_act)
            delta[:, :, js, :] = d_e.reshape(m, n, pe, D)
        next_observation = env.obs_postproc(observation, delta)
        if raw_actions is not None:
            ra = np.broadcast_to(raw_actions[:, :, t][:, :, None], (m, n, p))
        else:
            ra = np.broadcast_to(action[:, :, None, :], (m, n, p, A))
        reward = env.reward(observation, ra, next_observation)
        returns = returns + reward.astype(obs.dtype)
        observation = next_observation
        if return_traj:
            traj.append(observation.copy())
    if return_traj:
        return returns, np.stack(traj)
    return returns


# ----------------------------------------------------------------------------
# CEM pieces
# ----------------------------------------------------------------------------
def constrained_var(mean, var):  # utils.py:425-426
    dt = mean.dtype.type
    lb_dist, ub_dist = mean - dt(LOWER), dt(UPPER) - mean
    return np.minimum(np.minimum(np.square(lb_dist / dt(2)), np.square(ub_dist / dt(2))), var)


def sample_actions(mean, var, z):  # utils.py:427-429
    """actions [m,n,H,A] = mean + sqrt(constrained_var) * z, z truncated standard normal."""
    cvar = constrained_var(mean, var)
    return mean[:, None] + np.sqrt(cvar)[:, None] * z


def particle_mean(returns):  # utils.py:474
    return np.mean(returns, axis=2, dtype=returns.dtype)


def top_k_indices(ret, k):
    """tf.nn.top_k(sorted=True): descending, ties -> lower index first."""
    return np.argsort(-ret, axis=-1, kind="stable")[..., :k]


def elite_refit(mean, var, actions, cand_returns, num_elites=NUM_ELITES, alpha=ALPHA):
    """utils.py:475-486.  actions [m,n,H,A]; cand_returns [m,n] -> new mean, var [m,H,A]."""
    dt = mean.dtype.type
    idx = top_k_indices(cand_returns, num_elites)                      # :475
    elites = np.take_along_axis(actions, idx[:, :, None, None], axis=1)  # :476-480
    new_mean = np.mean(elites, axis=1, dtype=mean.dtype)               # :482
    new_var = np.mean(np.square(elites - new_mean[:, None]), axis=1, dtype=mean.dtype)  # :483
    mean = mean * dt(alpha) + dt(1 - alpha) * new_mean                 # :485
    var = var * dt(alpha) + dt(1 - alpha) * new_var                    # :486
    return mean, var, idx


def cem_plan(env, dyn, cp, st, obs, cp_obs, cp_act, init_mean, init_var, z, eps, E, p,
             deterministic=False, formulation="indexed", quirks=True, n_iters=NUM_CEM_ITERS,
             num_elites=NUM_ELITES, return_info=False):
    """Full CEM block (utils.py:398-488).  cp=None -> vanilla model (no context)."""
    m = obs.shape[0]
    ctx = None
    if cp is not None:
        ctx = nets.context_forward(cp, cp_obs, cp_act, st)              # :401-406
        tables_lit = context_table_literal(ctx, n_iters)
    mean, var = init_mean, init_var
    info = []
    for it in range(n_iters):
        actions = sample_actions(mean, var, z[it])
        T = None
        ctx_rows = None
        if ctx is not None:
            if formulation == "literal":
                T = tables_lit[it]
            elif quirks:
                T = context_table_indexed(ctx, it)
            else:
                ctx_rows = context_rows_fixed(ctx, p)
        if formulation == "literal":
            assert quirks, "the literal graph has the quirks by construction"
            rets = rollout_literal(env, dyn, st, obs, T, actions, eps[it], E, p, deterministic)
        else:
            rets = rollout_indexed(env, dyn, st, obs, T, actions, eps[it], E, p, deterministic,
                                   ctx_rows=ctx_rows)
        cand = particle_mean(rets)
        mean, var, idx = elite_refit(mean, var, actions, cand, num_elites)
        if return_info:
            info.append(dict(actions=actions, returns=rets, cand_returns=cand, elites=idx,
                             mean=mean.copy(), var=var.copy()))
    if return_info:
        return mean, info, ctx
    return mean


def context_rows_fixed(ctx, p):
    """Quirk-free layout: row (mi, j) uses the context of its OWN member."""
    E, m, C = ctx.shape
    pe = p // E
    out = np.empty((m, p, C), ctx.dtype)
    for j in range(p):
        out[:, j] = ctx[j // pe]
    return out


def rs_plan(env, dyn, cp, st, obs, cp_obs, cp_act, actions, eps, E, p, deterministic=False,
            formulation="indexed", raw_actions=None):
    """Random-shooting block (utils.py:498-561).  ``actions`` [m,n,H,A] are the injected
    U(-1,1) draws (or one-hot rows for discrete envs, fed un-normalised, :500).
    Returns (best first action [m,A] or [m] ints, cand_returns)."""
    T = None
    if cp is not None:
        ctx = nets.context_forward(cp, cp_obs, cp_act, st)
        T = context_table_literal(ctx, 1)[0]      # transposed ONCE (:513), Q1 only
    fn = rollout_literal if formulation == "literal" else rollout_indexed
    rets = fn(env, dyn, st, obs, T, actions, eps, E, p, deterministic,
              norm_actions=raw_actions is None, raw_actions=raw_actions)
    cand = particle_mean(rets)
    best = np.argmax(cand, axis=1)                 # :555 (first maximum)
    m = obs.shape[0]
    if raw_actions is not None:
        return raw_actions[np.arange(m), best, 0], cand
    return actions[np.arange(m), best, 0], cand


def get_action_clip(action, discrete=False):
    """mlp_cadm_ensemble_cem_dynamics.py:365-366."""
    if discrete:
        return action
    dt = action.dtype.type
    return np.minimum(np.maximum(action, dt(-1.0)), dt(1.0))


def warm_start_shift(prev_sol, sol):
    """cadm/samplers/sampler.py:118-120: shift the plan left, zero the tail,
    act with the first step."""
    prev_sol = prev_sol.copy()
    prev_sol[:, :-1] = sol[:, 1:].copy()
    prev_sol[:, -1:] = 0.0
    return prev_sol, sol[:, 0].copy()
