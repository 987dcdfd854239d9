"""Reference checkpoint layout <-> named tensors.

A reference checkpoint `params_epoch_k` is a joblib list of arrays in `tf.trainable_variables()` order
(/root/reference/cadm/dynamics/mlp_cadm_ensemble_cem_dynamics.py:571-588, creation order :140-266 and
core/utils.py:313-339,595-612): [context_model: cp_hidden_i_{weight,bias}.., cp_output_{weight,bias}]
[ff_model: hidden_i_{weight,bias}.., output_mu_*, output_logvar_*, max_logvar, min_logvar] [backward_model: same as
ff_model, only when back_coeff > 0]; `<path>_norm_stats` holds the OrderedDict of (mean, std) pairs.  The list carries no
names: this module infers the architecture from the shapes, names every entry, and converts both ways, so a model trained
with the TF1.15 reference anywhere can be replayed through the HIP planner (`model.load` takes the list as it is) and
inspected / edited as named arrays.  numpy + joblib only.
"""
from collections import OrderedDict

import numpy as np


def _is_w(a):
    return a.ndim == 3 and a.shape[1] > 1


def _is_b(a, w):
    return a.ndim == 3 and a.shape[0] == w.shape[0] and a.shape[1] == 1 and a.shape[2] == w.shape[2]


def _take_dense_chain(arrays, pos):
    """Consume (weight, bias) pairs whose widths chain; returns (pairs, next position)."""
    pairs = []
    while pos + 1 < len(arrays) and _is_w(arrays[pos]) and _is_b(arrays[pos + 1], arrays[pos]):
        if pairs and arrays[pos].shape[1] != pairs[-1][0].shape[2]:
            break
        pairs.append((arrays[pos], arrays[pos + 1]))
        pos += 2
    return pairs, pos


def describe(arrays):
    """Infer the architecture of a reference parameter list.  Returns dict(E, nets=[(net, [names])], context_dim,
    cp_hidden_sizes, hidden_sizes, obs_dim, input_dim, back_model); raises ValueError on a layout that is not the
    reference's."""
    arrays = [np.asarray(a) for a in arrays]
    n = len(arrays)
    pos, nets = 0, []
    info = dict(E=int(arrays[0].shape[0]) if n and arrays[0].ndim == 3 else None, context_dim=0, cp_hidden_sizes=(), back_model=False)

    def dyn_net(pos, label):
        # hidden chain, then output_mu, output_logvar (both HID -> D), then max_logvar, min_logvar [1, D]
        pairs, p2 = _take_dense_chain(arrays, pos)
        # the chain greedily swallowed output_mu (HID -> D); output_logvar has the same input width and is not chained
        if len(pairs) < 2:
            raise ValueError("%s: expected hidden layers + output_mu at entry %d" % (label, pos))
        mu, hidden = pairs[-1], pairs[:-1]
        Dd = mu[0].shape[2]
        if p2 + 4 > n or not (_is_w(arrays[p2]) and arrays[p2].shape == mu[0].shape and _is_b(arrays[p2 + 1], arrays[p2])):
            raise ValueError("%s: output_logvar (same shape as output_mu) missing at entry %d" % (label, p2))
        if arrays[p2 + 2].shape != (1, Dd) or arrays[p2 + 3].shape != (1, Dd):
            raise ValueError("%s: max_logvar / min_logvar [1,%d] missing at entry %d" % (label, Dd, p2 + 2))
        names = []
        for i in range(len(hidden)):
            names += ["hidden_%d_weight" % i, "hidden_%d_bias" % i]
        names += ["output_mu_weight", "output_mu_bias", "output_logvar_weight", "output_logvar_bias", "max_logvar", "min_logvar"]
        return names, p2 + 4, dict(hidden_sizes=tuple(int(w.shape[2]) for w, _ in hidden), obs_dim=int(Dd),
                                   input_dim=int(hidden[0][0].shape[1]))

    # a context encoder comes first iff the first chain is NOT followed by an equally shaped logvar head
    pairs, p2 = _take_dense_chain(arrays, 0)
    has_ctx = not (p2 + 1 < n and _is_w(arrays[p2]) and len(pairs) >= 1 and arrays[p2].shape == pairs[-1][0].shape)
    if has_ctx:
        if len(pairs) < 1:
            raise ValueError("no dense layers found at the head of the list")
        names = []
        for i in range(len(pairs) - 1):
            names += ["cp_hidden_%d_weight" % i, "cp_hidden_%d_bias" % i]
        names += ["cp_output_weight", "cp_output_bias"]
        nets.append(("context_model", names))
        info["context_dim"] = int(pairs[-1][0].shape[2])
        info["cp_hidden_sizes"] = tuple(int(w.shape[2]) for w, _ in pairs[:-1])
        info["cp_input_dim"] = int(pairs[0][0].shape[1])
        pos = p2
    names, pos, d = dyn_net(pos, "ff_model")
    nets.append(("ff_model", names))
    info.update(d)
    if pos < n:
        names, pos, d2 = dyn_net(pos, "backward_model")
        if d2 != d:
            raise ValueError("backward_model architecture %r differs from ff_model %r" % (d2, d))
        nets.append(("backward_model", names))
        info["back_model"] = True
    if pos != n:
        raise ValueError("%d trailing arrays after the last recognised network" % (n - pos))
    info["nets"] = nets
    return info


def to_named(arrays):
    """list (tf.trainable_variables() order) -> OrderedDict('net/param' -> array)."""
    info = describe(arrays)
    out, it = OrderedDict(), iter(arrays)
    for net, names in info["nets"]:
        for name in names:
            out["%s/%s" % (net, name)] = np.asarray(next(it))
    return out


def from_named(named):
    """OrderedDict / npz mapping 'net/param' -> array  ->  list in tf.trainable_variables() order."""
    keys = list(named.keys())
    def chain(net, prefix, tail):
        out, i = [], 0
        while "%s/%s_%d_weight" % (net, prefix, i) in named:
            out += ["%s/%s_%d_weight" % (net, prefix, i), "%s/%s_%d_bias" % (net, prefix, i)]
            i += 1
        return out + ["%s/%s" % (net, t) for t in tail]
    order = []
    if any(k.startswith("context_model/") for k in keys):
        order += chain("context_model", "cp_hidden", ["cp_output_weight", "cp_output_bias"])
    dyn_tail = ["output_mu_weight", "output_mu_bias", "output_logvar_weight", "output_logvar_bias", "max_logvar", "min_logvar"]
    order += chain("ff_model", "hidden", dyn_tail)
    if any(k.startswith("backward_model/") for k in keys):
        order += chain("backward_model", "hidden", dyn_tail)
    missing = [k for k in order if k not in named]
    extra = [k for k in keys if k not in order and not k.startswith("norm_stats/")]
    if missing or extra:
        raise ValueError("named checkpoint does not match the reference layout (missing %r, unexpected %r)" % (missing, extra))
    arrays = [np.asarray(named[k]) for k in order]
    describe(arrays)
    return arrays
