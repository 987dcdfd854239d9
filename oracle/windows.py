"""Oracle restatement of the reference's training-set builder for the context model
(`ModelSampleProcessor.process_samples`, /root/reference/cadm/samplers/model_sample_processor.py:23-127).

Test infrastructure only (see oracle/__init__.py).  PINNED: tests/golden/f2_windows.npz holds outputs of the reference's
own code (tests/golden/make_f2_golden.py) and tests/test_oracle_windows.py checks this restatement against them bit for bit.

Per path (after zero-padding short paths to F + 1 steps, :62-68) with L steps there are n = L - 1 rows; row s, future i:
    concat_obs[s, i]      = observations[s + i]        if s + i     <= L - 1 else 0     (:73,78-79)
    concat_act[s, i]      = actions[s + i]             if s + i     <= L - 1 else 0     (:74,82-83)
    concat_next_obs[s, i] = observations[s + 1 + i]    if s + 1 + i <= L - 1 else 0     (:75,77,80)
    concat_bool           = 1, then for i = 0..F-1: row (-i) columns max(i - remainder, 0).. = 0   (:70,85-86)
The i = 0 pass of that last loop indexes row -0 = row 0: the FIRST row of every path is masked out entirely (a quirk of the
reference, reproduced).  cp_observations / cp_actions are the history windows of the n rows (:99-100).
"""
import numpy as np


def discount_cumsum(x, discount):
    """cadm/utils/tensor_utils.py:217-221 (scipy.signal.lfilter recursion y[t] = x[t] + discount * y[t+1])."""
    y = np.zeros_like(np.asarray(x, np.float64))
    acc = 0.0
    for t in range(len(x) - 1, -1, -1):
        acc = x[t] + discount * acc
        y[t] = acc
    return y


def path_windows(observations, actions, F):
    """One path -> (concat_obs [n, F*D], concat_act [n, F*A], concat_next_obs [n, F*D], concat_bool [n, F], remainder)."""
    L0, D = observations.shape
    A = actions.shape[1]
    remainder = max(F + 1 - L0, 0)                                               # :60-63
    obs = np.concatenate([observations, np.zeros((remainder, D))], axis=0)
    act = np.concatenate([actions, np.zeros((remainder, A))], axis=0)
    L = obs.shape[0]
    n = L - 1
    co, ca, cn = np.zeros((n, F, D)), np.zeros((n, F, A)), np.zeros((n, F, D))
    for i in range(F):
        for s in range(n):
            if s + i <= L - 1:
                co[s, i] = obs[s + i]
                ca[s, i] = act[s + i]
            if s + 1 + i <= L - 1:
                cn[s, i] = obs[s + 1 + i]
    cb = np.ones((n, F))                                                          # :70
    for i in range(F):
        cb[-i][max(i - remainder, 0):] = 0                                        # :85-86 (i = 0 hits row 0)
    return co.reshape(n, F * D), ca.reshape(n, F * A), cn.reshape(n, F * D), cb, remainder


def process_samples(paths, F, discount=0.99):
    """paths: list of dicts(observations [L,D], actions [L,A], rewards [L], cp_obs [L,D*Hh], cp_act [L,A*Hh]) ->
    the reference's samples_data dict (context=True branch)."""
    out = {k: [] for k in ("concat_obs", "concat_act", "concat_next_obs", "concat_bool", "cp_observations", "cp_actions",
                           "observations", "next_observations", "actions", "timesteps", "rewards", "returns")}
    for p in paths:
        o, a = np.asarray(p["observations"]), np.asarray(p["actions"])
        out["observations"].append(o[:-1]); out["next_observations"].append(o[1:]); out["actions"].append(a[:-1])   # :45-47
        out["timesteps"].append(np.arange(len(o) - 1))                                                             # :48
        out["rewards"].append(np.asarray(p["rewards"])[:-1])                                                       # :50
        out["returns"].append(discount_cumsum(np.asarray(p["rewards"]), discount))                                 # :39-41,51
        co, ca, cn, cb, rem = path_windows(o, a, F)
        out["concat_obs"].append(co); out["concat_act"].append(ca); out["concat_next_obs"].append(cn); out["concat_bool"].append(cb)
        cpo = np.concatenate([p["cp_obs"], np.zeros((rem, p["cp_obs"].shape[1]))], axis=0)                         # :67-68
        cpa = np.concatenate([p["cp_act"], np.zeros((rem, p["cp_act"].shape[1]))], axis=0)
        out["cp_observations"].append(cpo[:-1]); out["cp_actions"].append(cpa[:-1])                                # :99-100
    return {k: np.concatenate(v, axis=0) for k, v in out.items()}
