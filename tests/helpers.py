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


def f16_split_saturating(x):
    """What the rollout kernels feed the matrix pipe for a value x: hi = sat(f16(x)), lo = sat(f16(x - hi)) with f16 overflow saturating
    at +-65504 (MODE.FP16_OVFL, csrc/rollout_env.h: fp16_saturate_on), returned as hi + lo in fp32: x to 2^-22 inside the f16 range,
    at most +-131008 beyond it."""
    x = np.asarray(x, np.float32)
    big = np.float32(65504.0)

    def sat(v):
        with np.errstate(over="ignore", invalid="ignore"):
            h = v.astype(np.float16).astype(np.float32)
        return np.where(np.isinf(h) & np.isfinite(v), np.sign(v) * big, h).astype(np.float32)
    hi = sat(x)
    with np.errstate(invalid="ignore"):
        lo = sat((x - hi).astype(np.float32))
    return (hi + lo).astype(np.float32)
