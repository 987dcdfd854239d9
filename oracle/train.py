"""Oracle restatement of the reference training step (losses, gradients, TF1 Adam)
and of ``fit``'s host-side data preparation.

Test infrastructure only (see oracle/__init__.py).

Follows /root/reference/cadm/dynamics/mlp_cadm_ensemble_cem_dynamics.py:
  losses                :269-314   (vanilla twin mlp_ensemble_cem_dynamics.py:148-167)
  optimizer.minimize    :316-317   (tf.compat.v1.train.AdamOptimizer, TF 1.15:
                                    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); var -= lr_t*m/(sqrt(v)+eps))
  _preprocess_inputs    :676-696
  compute_normalization :590-602, get_normalization_stats :604-645
Forward math is written with torch so that torch.autograd supplies the
gradients (fp64 = truth, fp32 = "TF-like").
"""
from collections import OrderedDict

import numpy as np
import torch

EPS_STD = 1e-10


def _norm(x, mean, std):
    return (x - mean) / (std + EPS_STD)


def _softplus_tf(x):
    thr = float(np.log(torch.finfo(x.dtype).eps)) + 2.0
    ex = torch.exp(torch.clamp(x, max=-thr + 1.0))
    out = torch.log1p(ex)
    out = torch.where(x < thr, ex, out)
    out = torch.where(x > -thr, x, out)
    return out


def _swish(x):
    return x * torch.sigmoid(x)


_ACTS = {"swish": _swish, "relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid, None: lambda x: x}   # dynamics.py:17-24


def _mlp(p, x, n_hidden, act=_swish):
    for i in range(n_hidden):
        x = act(torch.matmul(x, p["hidden_%d_weight" % i]) + p["hidden_%d_bias" % i])
    mu = torch.matmul(x, p["output_mu_weight"]) + p["output_mu_bias"]
    lv = torch.matmul(x, p["output_logvar_weight"]) + p["output_logvar_bias"]
    return mu, lv


def _l2(p, wds, prefix_list):
    tot = 0.0
    for name, wd in zip(prefix_list, wds):
        tot = tot + wd * (p[name] ** 2).sum() / 2.0   # tf.nn.l2_loss, utils.py:642
    return tot


def dyn_l2_names(n_hidden):
    return ["hidden_%d_weight" % i for i in range(n_hidden)] + ["output_mu_weight", "output_logvar_weight"]


def dyn_weight_decays(weight_decays, n_hidden):
    """utils.py:319,328,334: hidden_i -> weight_decays[i]; both heads -> weight_decays[-1]."""
    return [weight_decays[i] for i in range(n_hidden)] + [weight_decays[-1], weight_decays[-1]]


def ctx_l2_names(n_cp_hidden):
    return ["cp_hidden_%d_weight" % i for i in range(n_cp_hidden)] + ["cp_output_weight"]


def ctx_weight_decays(context_weight_decays, n_cp_hidden):
    """utils.py:601,610."""
    return [context_weight_decays[i] for i in range(n_cp_hidden)] + [context_weight_decays[-1]]


def to_torch(d, dtype, requires_grad=False):
    out = OrderedDict()
    for k, v in d.items():
        t = torch.tensor(np.asarray(v), dtype=dtype)
        t.requires_grad_(requires_grad)
        out[k] = t
    return out


def preproc_torch(env_name, obs):
    if env_name in ("halfcheetah", "cripple_halfcheetah"):
        return torch.cat([obs[..., 1:2], torch.sin(obs[..., 2:3]), torch.cos(obs[..., 2:3]), obs[..., 3:]], -1)
    if env_name == "ant":
        return obs[..., 1:]
    return obs


def train_losses(env_name, ff, back, cp, st, batch, cfg):
    """Training graph (dynamics.py:269-314).  ff/back/cp are dicts of torch tensors
    (back / cp may be None), st a dict of torch stat vectors, batch a dict with
    obs, act, delta, obs_next, back_delta, cp_obs, cp_act  [E,B,.].
    cfg: deterministic, back_coeff, weight_decay_coeff, weight_decays,
    context_weight_decays, n_hidden, n_cp_hidden.
    Returns dict(loss, mse, back_mse, recon)."""
    nh = cfg["n_hidden"]
    feats = [_norm(preproc_torch(env_name, batch["obs"]), st["obs_mean"], st["obs_std"]),
             _norm(batch["act"], st["act_mean"], st["act_std"])]
    ctx = None
    if cp is not None:
        x = torch.cat([_norm(batch["cp_obs"], st["cp_obs_mean"], st["cp_obs_std"]),
                       _norm(batch["cp_act"], st["cp_act_mean"], st["cp_act_std"])], -1)
        for i in range(cfg["n_cp_hidden"]):
            z = torch.matmul(x, cp["cp_hidden_%d_weight" % i]) + cp["cp_hidden_%d_bias" % i]
            # relu; cfg["cp_relu_threshold"] (tools/fuzz_train.py only, default 0 = the reference's relu) moves the kink by a hair, to tell a
            # pre-activation that rounds to the other side of 0 in fp32 from a wrong gradient
            thr = cfg.get("cp_relu_threshold", 0.0)
            if isinstance(thr, dict):          # {layer: tensor shaped like z}: single units moved to one side of the kink or the other
                thr = thr.get(i, 0.0)
            x = torch.where(z > thr, z, torch.zeros_like(z))
        ctx = torch.matmul(x, cp["cp_output_weight"]) + cp["cp_output_bias"]
        feats.append(ctx)
    act = _ACTS[cfg.get("hidden_nonlinearity", "swish")]
    mu, lv = _mlp(ff, torch.cat(feats, -1), nh, act)
    targ = _norm(batch["delta"], st["delta_mean"], st["delta_std"])

    def red(x):  # reduce_sum_e(reduce_mean_b(reduce_mean_d(.)))  dynamics.py:273-274
        return x.mean(-1).mean(-1).sum()

    mse = red((mu - targ) ** 2)
    l2 = _l2(ff, dyn_weight_decays(cfg["weight_decays"], nh), dyn_l2_names(nh))
    if cp is not None:
        l2 = l2 + _l2(cp, ctx_weight_decays(cfg["context_weight_decays"], cfg["n_cp_hidden"]),
                      ctx_l2_names(cfg["n_cp_hidden"]))
    back_mse = torch.zeros((), dtype=mu.dtype)
    if back is not None and cfg["back_coeff"] > 0.0:
        bfeats = [_norm(preproc_torch(env_name, batch["obs_next"]), st["obs_mean"], st["obs_std"]),
                  _norm(batch["act"], st["act_mean"], st["act_std"])]
        if ctx is not None:
            bfeats.append(ctx)
        bmu, _ = _mlp(back, torch.cat(bfeats, -1), nh, act)
        btarg = _norm(batch["back_delta"], st["back_delta_mean"], st["back_delta_std"])
        back_mse = red((bmu - btarg) ** 2)                                   # :280-281
        l2 = l2 + _l2(back, dyn_weight_decays(cfg["weight_decays"], nh), dyn_l2_names(nh))  # :283,293
    coeff = cfg["weight_decay_coeff"]
    if cfg["deterministic"]:
        recon = mse
        if back is not None and cfg["back_coeff"] > 0.0:
            recon = recon + cfg["back_coeff"] * back_mse                      # :297-300
        loss = recon + l2 * coeff                                             # :301
    else:
        max_lv, min_lv = ff["max_logvar"], ff["min_logvar"]
        lvc = max_lv - _softplus_tf(max_lv - lv)                              # utils.py:356
        lvc = min_lv + _softplus_tf(lvc - min_lv)                             # utils.py:357
        invvar = torch.exp(-lvc)                                              # :303
        mu_loss = red((mu - targ) ** 2 * invvar)                              # :304-305
        var_loss = red(lvc)                                                   # :306-307
        reg = 0.01 * max_lv.sum() - 0.01 * min_lv.sum()                       # :308
        recon = mu_loss + var_loss
        if back is not None and cfg["back_coeff"] > 0.0:
            recon = recon + cfg["back_coeff"] * back_mse                      # :311-312
        loss = recon + reg + l2 * coeff                                       # :314
    return dict(loss=loss, mse=mse, back_mse=back_mse, recon=recon)


def context_preacts(cp, st, batch, cfg):
    """Pre-activations of the context encoder's hidden layers (list of [E,B,width] tensors) -- tools/fuzz_train.py looks for relu units
    within roundoff of their kink."""
    x = torch.cat([_norm(batch["cp_obs"], st["cp_obs_mean"], st["cp_obs_std"]),
                   _norm(batch["cp_act"], st["cp_act_mean"], st["cp_act_std"])], -1)
    out = []
    for i in range(cfg["n_cp_hidden"]):
        z = torch.matmul(x, cp["cp_hidden_%d_weight" % i]) + cp["cp_hidden_%d_bias" % i]
        out.append(z.detach())
        x = torch.relu(z)
    return out


def grads_of(loss, nets):
    """Gradients of ``loss`` w.r.t. every tensor in the given dicts; tensors the
    loss does not depend on get None (TF: such variables are skipped by minimize)."""
    flat = [(n, k, v) for n, d in nets.items() if d is not None for k, v in d.items()]
    gs = torch.autograd.grad(loss, [v for _, _, v in flat], allow_unused=True)
    out = {n: OrderedDict() for n, d in nets.items() if d is not None}
    for (n, k, _), g in zip(flat, gs):
        out[n][k] = g
    return out


class TF1Adam:
    """tf.compat.v1.train.AdamOptimizer (TF 1.15 adam.py / training_ops ApplyAdam)."""

    def __init__(self, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.t = 0
        self.m, self.v = {}, {}

    def step(self, params, grads):
        """params / grads: {net: {name: tensor}}; updates params in place (no autograd)."""
        self.t += 1
        lr_t = self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        with torch.no_grad():
            for n, d in params.items():
                if d is None:
                    continue
                for k, v in d.items():
                    g = grads[n][k]
                    if g is None:
                        continue
                    key = (n, k)
                    if key not in self.m:
                        self.m[key] = torch.zeros_like(v)
                        self.v[key] = torch.zeros_like(v)
                    self.m[key].mul_(self.b1).add_(g, alpha=1 - self.b1)
                    self.v[key].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                    v.sub_(lr_t * self.m[key] / (self.v[key].sqrt() + self.eps))


# ----------------------------------------------------------------------------
# fit()'s host-side data path
# ----------------------------------------------------------------------------
def preprocess_inputs(obs, act, delta, cp_obs, cp_act, future_bool, obs_next, back_delta,
                      D, A, Hh, F):
    """dynamics.py:676-696: explode [N, F*.] samples into rows, tile cp_*, mask by future_bool."""
    fb = future_bool.reshape(-1)
    _obs = obs.reshape((-1, D))
    _act = act.reshape((-1, A))
    _delta = delta.reshape((-1, D))
    _obs_next = obs_next.reshape((-1, D))
    _back_delta = back_delta.reshape((-1, D))
    _cp_obs = np.tile(cp_obs, (1, F)).reshape((-1, D * Hh))
    _cp_act = np.tile(cp_act, (1, F)).reshape((-1, A * Hh))
    k = fb > 0
    return (_obs[k], _act[k], _delta[k], _obs_next[k], _back_delta[k], _cp_obs[k], _cp_act[k])


def compute_normalization(env, obs, act, delta, cp_obs, cp_act, back_delta):
    """dynamics.py:590-602 (population mean/std, float64)."""
    proc = env.obs_preproc(obs)
    n = OrderedDict()
    n["obs"] = (np.mean(proc, axis=0), np.std(proc, axis=0))
    n["delta"] = (np.mean(delta, axis=0), np.std(delta, axis=0))
    n["act"] = (np.mean(act, axis=0), np.std(act, axis=0))
    n["cp_obs"] = (np.mean(cp_obs, axis=0), np.std(cp_obs, axis=0))
    n["cp_act"] = (np.mean(cp_act, axis=0), np.std(cp_act, axis=0))
    n["back_delta"] = (np.mean(back_delta, axis=0), np.std(back_delta, axis=0))
    return n


def normalization_stats(norm, D, A, Hh, discrete, state_diff):
    """dynamics.py:604-645 with normalize_input=True -> dict keyed like nets.STAT_KEYS."""
    s = {}
    s["obs_mean"], s["obs_std"] = norm["obs"]
    s["delta_mean"], s["delta_std"] = norm["delta"]
    if discrete:
        s["act_mean"], s["act_std"] = np.zeros(A), np.ones(A)
    else:
        s["act_mean"], s["act_std"] = norm["act"]
    if state_diff:
        s["cp_obs_mean"], s["cp_obs_std"] = np.zeros(D * Hh), np.ones(D * Hh)
    else:
        s["cp_obs_mean"], s["cp_obs_std"] = norm["cp_obs"]
    if discrete:
        s["cp_act_mean"], s["cp_act_std"] = np.zeros(A * Hh), np.ones(A * Hh)
    else:
        s["cp_act_mean"], s["cp_act_std"] = norm["cp_act"]
    s["back_delta_mean"], s["back_delta_std"] = norm["back_delta"]
    return s


def early_stop_trace(v_recon, persistency):
    """Restatement of the rolling-average early stop of `fit` (reference
    cadm/dynamics/mlp_cadm_ensemble_cem_dynamics.py:544-564): given the per-epoch validation recon losses,
    returns (index of the last epoch that ran, the rolling averages after each of those epochs)."""
    rolling, prev, trace = None, None, []
    last = -1
    for epoch, v in enumerate(v_recon):
        last = epoch
        if rolling is None:                      # :544-549
            rolling, prev = 1.5 * v, 2 * v
            if v < 0:
                rolling, prev = v / 1.5, v / 2
        rolling = persistency * rolling + (1.0 - persistency) * v      # :551-552
        trace.append(rolling)
        if prev < rolling:                       # :554-556
            break
        prev = rolling                           # :564
    return last, trace


def _fit_setup(env, data, stream, E, valid_split_ratio, max_logging, discrete, state_diff):
    """Everything `fit` does before its epoch loop (dynamics.py:399-467): targets, normalisation over the first-step columns,
    window split, `_preprocess_inputs`, bootstrap indices.  -> (stats, train rows, valid rows or None, bootstrap_idx, n_train)"""
    D, A = env.obs_dim, env.act_dim
    obs, act, obs_next = data["obs"], data["act"], data["obs_next"]
    cp_obs, cp_act, future_bool = data["cp_obs"], data["cp_act"], data["future_bool"]
    F = future_bool.shape[1]
    Hh = cp_obs.shape[1] // D if D else 0
    o1, on1 = obs.reshape(-1, D), obs_next.reshape(-1, D)
    delta = env.targ_proc(o1, on1).reshape(-1, F * D)                    # :401,405
    back_delta = env.targ_proc(on1, o1).reshape(-1, F * D)               # :402,407
    norm = compute_normalization(env, obs[:, :D], act[:, :A], delta[:, :D], cp_obs, cp_act, back_delta[:, :D])   # :428-440
    stats_np = normalization_stats(norm, D, A, Hh, discrete, state_diff)
    N = obs.shape[0]
    n_valid = min(int(N * valid_split_ratio), max_logging)              # :443
    perm = np.asarray(stream.permutation(N))                            # :444
    tr, va = perm[n_valid:], perm[:n_valid]
    cols = lambda idx: preprocess_inputs(obs[idx], act[idx], delta[idx], cp_obs[idx], cp_act[idx], future_bool[idx],
                                         obs_next[idx], back_delta[idx], D, A, Hh, F)
    names = ("obs", "act", "delta", "obs_next", "back_delta", "cp_obs", "cp_act")
    train = dict(zip(names, cols(tr)))
    valid = dict(zip(names, cols(va))) if n_valid > 0 else None
    n_train = train["obs"].shape[0]
    if E > 1:
        bidx = np.asarray(stream.bootstrap(E, n_train))                 # :465
    else:
        bidx = np.tile(np.arange(n_train), (E, 1))                      # :467
    return stats_np, train, valid, bidx, n_train


def fit_feed_sequence(env, data, stream, E, epochs, batch_size, valid_split_ratio=0.2, max_logging=5000, discrete=False,
                      state_diff=True):
    """What `fit` FEEDS the graph, without training: (stats, [per epoch: ([train batches], valid batch or None)]); a batch is a dict
    of [E, B, .] arrays.  Same draws from `stream` as `fit_reference` (tests/test_loss_golden.py compares it with the feeds of
    the reference's own fit())."""
    stats, train, valid, bidx, n_train = _fit_setup(env, data, stream, E, valid_split_ratio, max_logging, discrete, state_diff)
    out = []
    for _ in range(epochs):
        order = np.asarray(stream.epoch_order(E, n_train))              # shuffle_rows, :472-474
        bidx = bidx[np.arange(E)[:, None], order]                       # :483
        batches = []
        for b in range(int(np.ceil(n_train / batch_size))):
            idx = bidx[:, b * batch_size:(b + 1) * batch_size]          # :487
            batches.append({k: v[idx] for k, v in train.items()})
        vb = None
        if valid is not None and valid["obs"].shape[0] > 0:
            vidx = np.tile(np.arange(valid["obs"].shape[0]), (E, 1))    # :469-470
            vb = {k: v[vidx] for k, v in valid.items()}
        out.append((batches, vb))
    return stats, out


# ----------------------------------------------------------------------------
# fit(): the whole host loop, on an injected index stream
# ----------------------------------------------------------------------------
def fit_reference(env, env_name, nets, data, stream, cfg, epochs, batch_size, valid_split_ratio=0.2,
                  max_logging=5000, persistency=0.99, dtype=torch.float64, lr=1e-3):
    """Restatement of `MLPEnsembleCEMDynamicsModel.fit` for ONE call on an empty dataset
    (/root/reference/cadm/dynamics/mlp_cadm_ensemble_cem_dynamics.py:382-569): targets (:399-407), dataset columns
    (:409-425), normalisation over the first-step columns (:428-440), window split (:442-453), `_preprocess_inputs`
    (:455-459), bootstrap indices (:463-467), per-epoch `shuffle_rows` (:472-474,483), one Adam step per batch
    (:486-513), validation on identical indices for every member (:469-470,516-537) and the rolling-average early stop
    (:541-564).  Every random draw comes from `stream` (permutation / bootstrap / epoch_order -- the order the reference
    consumes `np.random`).

    env: oracle.envs spec (numpy closures); nets: dict(ff=, back=, cp=) of OrderedDicts of numpy arrays (back / cp may be
    None); data: dict(obs, act, obs_next, cp_obs, cp_act, future_bool) windowed as the reference's fit() takes them;
    cfg: the `train_losses` cfg + state_diff, discrete.
    Returns dict(train=[per-step (mse, back_mse, recon)], valid=[per-epoch], params={net: {name: numpy}}, stats=dict,
    epochs_run)."""
    E = next(iter(nets["ff"].values())).shape[0]
    stats_np, train, valid, bidx, n_train = _fit_setup(env, data, stream, E, valid_split_ratio, max_logging,
                                                       cfg.get("discrete", False), cfg.get("state_diff", True))
    st = {k: torch.tensor(np.asarray(v), dtype=dtype) for k, v in stats_np.items()}
    params = {k: (None if v is None else to_torch(v, dtype, requires_grad=True)) for k, v in nets.items()}
    adam = TF1Adam(lr=lr)
    tb = lambda d, idx: {k: torch.tensor(np.asarray(v[idx]), dtype=dtype) for k, v in d.items()}
    out_train, out_valid = [], []
    rolling, prev = None, None
    epochs_run = 0
    for epoch in range(epochs):
        epochs_run = epoch + 1
        order = np.asarray(stream.epoch_order(E, n_train))              # shuffle_rows, :472-474
        bidx = bidx[np.arange(E)[:, None], order]                       # :483
        for b in range(int(np.ceil(n_train / batch_size))):
            idx = bidx[:, b * batch_size:(b + 1) * batch_size]          # :487  [E, B]
            res = train_losses(env_name, params["ff"], params.get("back"), params.get("cp"), st, tb(train, idx), cfg)
            out_train.append((float(res["mse"].detach()), float(res["back_mse"].detach()), float(res["recon"].detach())))
            adam.step(params, grads_of(res["loss"], params))            # :508-510
        if valid is not None and valid["obs"].shape[0] > 0:
            vidx = np.tile(np.arange(valid["obs"].shape[0]), (E, 1))    # :469-470
            with torch.no_grad():
                res = train_losses(env_name, params["ff"], params.get("back"), params.get("cp"), st, tb(valid, vidx), cfg)
            v = float(res["recon"])
            out_valid.append((float(res["mse"]), float(res["back_mse"]), v))
            if rolling is None:                                         # :544-549
                rolling, prev = 1.5 * v, 2 * v
                if v < 0:
                    rolling, prev = v / 1.5, v / 2
            rolling = persistency * rolling + (1.0 - persistency) * v   # :551-552
            if prev < rolling:                                          # :554-556
                break
        prev = rolling                                                  # :564
    final = {k: (None if v is None else OrderedDict((n, t.detach().numpy().copy()) for n, t in v.items())) for k, v in params.items()}
    return dict(train=out_train, valid=out_valid, params=final, stats=stats_np, epochs_run=epochs_run)
