"""Shared helpers for the test-suite (tests may import both the product and the oracle)."""
import numpy as np

from cadm_amd import synth
from oracle import envs as oenvs
from oracle import nets as onets


def oracle_problem(prob, dtype):
    """Cast a synth problem to oracle inputs of the given dtype."""
    o = dict(env=oenvs.make_env(prob["env"]),
             ff=onets.cast_params(prob["ff"], dtype),
             cp=None if prob["cp"] is None else onets.cast_params(prob["cp"], dtype),
             back=None if prob.get("back") is None else onets.cast_params(prob["back"], dtype),
             st=onets.cast_stats(prob["stats"], dtype))
    for k in ("obs", "cp_obs", "cp_act", "init_mean", "init_var"):
        o[k] = prob[k].astype(dtype)
    return o


make_engine = synth.make_engine   # product code (cadm_amd/synth.py); re-exported for the tests


def trunc_z(rng, shape):
    z = rng.standard_normal(shape)
    bad = np.abs(z) >= 2.0
    while bad.any():
        z[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(z) >= 2.0
    return z


def rel_err(a, b):
    """norm-wise relative error max|a-b| / max|b|."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def assert_close(a, b, rtol=1e-5, what=""):
    """|a-b| <= rtol * max(|b|, scale) elementwise with scale = rms(b): the 1e-5 relative bar of
    BASELINE.json with an absolute floor that tolerates cancellation in fp32 dot products."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(float(np.sqrt(np.mean(b * b))), 1e-30)
    bound = rtol * np.maximum(np.abs(b), scale)
    bad = np.abs(a - b) > bound
    assert not bad.any(), "%s: %d/%d elements off, worst |diff|=%.3e (bound %.3e), rel_err=%.3e" % (
        what, int(bad.sum()), bad.size, float(np.abs(a - b).max()), float(bound.flat[np.abs(a - b).argmax()]), rel_err(a, b))


def floor_reliance(a, b, rtol=1e-5):
    """How much of `assert_close`'s verdict leans on its rms floor: (elements that violate the PURE relative bound
    |a-b| <= rtol*|b|, total elements, largest |b|/rms among those).  Elements far below the tensor's rms are sums that
    cancelled: no fp32 evaluation order reproduces them to 1e-5 of themselves."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(float(np.sqrt(np.mean(b * b))), 1e-30)
    bad = np.abs(a - b) > rtol * np.abs(b)
    worst = float((np.abs(b)[bad] / scale).max()) if bad.any() else 0.0
    return int(bad.sum()), int(bad.size), worst
