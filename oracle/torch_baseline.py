"""CPU baseline: op-for-op torch-CPU fp32 restatement of the reference's CEM graph.

Test / baseline infrastructure only (see oracle/__init__.py): timed by bench.py's
``cpu_baseline`` leg ("kind": "port") on the GPU node's host cores.  It mirrors how the TF1.15
graph executes (/root/reference/cadm/dynamics/core/utils.py:424-488): batched matmuls on
[E, R/E, K], with the tile / transpose / reshape steps MATERIALISED every time step exactly as
the graph does, sampling with torch's CPU generator.
"""
import torch


def _norm(x, mean, std):
    return (x - mean) / (std + 1e-10)


def _preproc(env, o):
    if env in ("halfcheetah", "cripple_halfcheetah"):
        return torch.cat([o[..., 1:2], torch.sin(o[..., 2:3]), torch.cos(o[..., 2:3]), o[..., 3:]], -1)
    if env == "ant":
        return o[..., 1:]
    return o


def _postproc(env, o, d):
    if env in ("halfcheetah", "cripple_halfcheetah", "ant"):
        return torch.cat([d[..., :1], o[..., 1:] + d[..., 1:]], -1)
    return o + d


def _reward(env, o, a, o2):
    if env in ("halfcheetah", "cripple_halfcheetah"):
        return o[..., 0] - 0.1 * (a * a).sum(-1)
    if env == "ant":
        return o[..., 0] - 0.005 * (a * a).sum(-1) + 0.05
    if env == "slim_humanoid":
        alive = ((o[..., 1] > 1.0) & (o[..., 1] < 2.0)).to(o.dtype)
        return (0.25 / 0.015) * o[..., 22] - 0.1 * (a * a).sum(-1) + 5.0 * alive
    raise NotImplementedError(env)


def _forward(ff, nh, x, dmean, dstd, deterministic):
    for i in range(nh):
        x = torch.baddbmm(ff["hidden_%d_bias" % i], x, ff["hidden_%d_weight" % i])
        x = x * torch.sigmoid(x)
    mu = torch.baddbmm(ff["output_mu_bias"], x, ff["output_mu_weight"])
    dmu = mu * (dstd + 1e-10) + dmean
    if deterministic:
        return dmu
    lv = torch.baddbmm(ff["output_logvar_bias"], x, ff["output_logvar_weight"])
    lv = ff["max_logvar"] - torch.nn.functional.softplus(ff["max_logvar"] - lv)
    lv = ff["min_logvar"] + torch.nn.functional.softplus(lv - ff["min_logvar"])
    return dmu + torch.randn_like(dmu) * torch.exp((lv + 2 * torch.log(dstd)) / 2.0)


@torch.no_grad()
def cem_get_action(prob, n, p, deterministic=False, n_iters=5, num_elites=50, alpha=0.1):
    """One `get_action` (context encoder + 5 CEM iterations).  prob: dict of torch CPU fp32
    tensors prepared by `prepare`."""
    env, E, H, D, A = prob["env"], prob["E"], prob["H"], prob["D"], prob["A"]
    ff, cp, st = prob["ff"], prob["cp"], prob["st"]
    obs = prob["obs"]
    m = obs.shape[0]
    pe = p // E
    bs_cp = None
    if cp is not None:
        x = torch.cat([_norm(prob["cp_obs"][None].repeat(E, 1, 1), st["cp_obs_mean"], st["cp_obs_std"]),
                       _norm(prob["cp_act"][None].repeat(E, 1, 1), st["cp_act_mean"], st["cp_act_std"])], -1)
        for i in range(prob["n_cp_hidden"]):
            x = torch.relu(torch.baddbmm(cp["cp_hidden_%d_bias" % i], x, cp["cp_hidden_%d_weight" % i]))
        bs_cp = torch.baddbmm(cp["cp_output_bias"], x, cp["cp_output_weight"])
    mean, var = prob["init_mean"].clone(), prob["init_var"].clone()
    for _ in range(n_iters):
        cvar = torch.minimum(torch.minimum(((mean + 1.0) / 2) ** 2, ((1.0 - mean) / 2) ** 2), var)
        z = torch.empty((m, n, H, A))
        torch.nn.init.trunc_normal_(z, 0.0, 1.0, -2.0, 2.0)
        actions = mean[:, None] + cvar.sqrt()[:, None] * z
        returns = torch.zeros((m, n, p))
        observation = obs.reshape(m, 1, 1, D).repeat(1, n, p, 1)
        if bs_cp is not None:
            bs_cp = bs_cp.transpose(0, 1).contiguous()
            C = bs_cp.shape[-1]
            context = bs_cp.reshape(m, 1, E, C).repeat(1, n, pe, 1)
            rctx = context.permute(2, 0, 1, 3).reshape(E, pe * m * n, C)
        for t in range(H):
            action = actions[:, :, t]
            nact = _norm(action, st["act_mean"], st["act_std"])[:, :, None, :].repeat(1, 1, p, 1)
            nact = nact.permute(2, 0, 1, 3).reshape(E, pe * m * n, A)
            nobs = _norm(_preproc(env, observation), st["obs_mean"], st["obs_std"])
            nobs = nobs.permute(2, 0, 1, 3).reshape(E, pe * m * n, -1)
            x = torch.cat([nobs, nact] + ([rctx] if bs_cp is not None else []), 2)
            delta = _forward(ff, prob["n_hidden"], x, st["delta_mean"], st["delta_std"], deterministic)
            delta = delta.reshape(p, m, n, D).permute(1, 2, 0, 3)
            nxt = _postproc(env, observation, delta)
            returns = returns + _reward(env, observation, action[:, :, None, :].repeat(1, 1, p, 1), nxt)
            observation = nxt
        ret = returns.mean(2)
        idx = torch.topk(ret, num_elites, dim=1, sorted=True).indices
        elites = torch.gather(actions, 1, idx[:, :, None, None].expand(-1, -1, H, A))
        new_mean = elites.mean(1)
        new_var = ((elites - new_mean[:, None]) ** 2).mean(1)
        mean = mean * alpha + (1 - alpha) * new_mean
        var = var * alpha + (1 - alpha) * new_var
    return mean.clamp(-1.0, 1.0)


def prepare(synth_prob):
    t = lambda v: torch.tensor(v, dtype=torch.float32)
    out = dict(env=synth_prob["env"], E=synth_prob["E"], H=synth_prob["H"], D=synth_prob["D"], A=synth_prob["A"],
               n_hidden=len(synth_prob["hidden_sizes"]), n_cp_hidden=len(synth_prob["cp_hidden_sizes"]))
    out["ff"] = {k: t(v) for k, v in synth_prob["ff"].items()}
    out["cp"] = None if synth_prob["cp"] is None else {k: t(v) for k, v in synth_prob["cp"].items()}
    out["st"] = {k: t(v) for k, v in synth_prob["stats"].items()}
    for k in ("obs", "cp_obs", "cp_act", "init_mean", "init_var"):
        out[k] = t(synth_prob[k])
    return out
