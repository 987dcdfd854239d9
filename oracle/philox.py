"""Oracle restatement of the counter-based RNG the HIP planner uses.

Test infrastructure only (see oracle/__init__.py).

The reference draws from TF's unseeded Philox streams (utils.py:365,429,499-502;
``set_seed`` is never called), which are not reproducible, so the
build defines its own stream layout on the *published*
Philox4x32-10 generator (Salmon et al., "Parallel random numbers: as easy as
1, 2, 3", SC'11; Random123 v1.x).  Pinned here against Random123's known-answer
vectors (tests/test_oracle_philox.py).
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint32(0x9E3779B9)
W1 = np.uint32(0xBB67AE85)

STREAM_EPS = 1   # Gaussian-head noise
STREAM_ACT = 2   # truncated-normal action draws
STREAM_UNI = 3   # random-shooting uniforms


def philox4x32_10(ctr, key):
    """ctr [...,4] uint32, key [...,2] uint32 -> [...,4] uint32."""
    c = [np.asarray(ctr[..., i], dtype=np.uint32) for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint32)
    k1 = np.asarray(key[..., 1], dtype=np.uint32)
    mask = np.uint64(0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for r in range(10):
            p0 = M0 * c[0].astype(np.uint64)
            p1 = M1 * c[2].astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & mask).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & mask).astype(np.uint32)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            if r < 9:
                k0 = (k0 + W0).astype(np.uint32)
                k1 = (k1 + W1).astype(np.uint32)
    return np.stack(c, axis=-1)


def u01(x):
    """uint32 -> float32 in (0,1): (x >> 8) * 2^-24 + 2^-25."""
    return ((x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
            + np.float32(2.0 ** -25))


def box_muller(u1, u2):
    r = np.sqrt(np.float32(-2.0) * np.log(u1))
    th = np.float32(2.0 * np.pi) * u2
    return r * np.cos(th), r * np.sin(th)


def _ctr(a, b, c, d):
    a, b, c, d = np.broadcast_arrays(a, b, c, d)
    return np.stack([a, b, c, d], axis=-1).astype(np.uint32)


def _key(seed, call, shape):
    k = np.empty(shape + (2,), np.uint32)
    k[..., 0] = np.uint32(seed & 0xFFFFFFFF)
    k[..., 1] = np.uint32(call & 0xFFFFFFFF)
    return k


def eps_normals(seed, call, it, m, n_global, p, H, D, cand_lo=0, cand_hi=None):
    """Gaussian-head noise [H, m, n, p, D] for candidates [cand_lo, cand_hi).
    One call yields four words = two Box-Muller pairs = the noise of TWO dim pairs: pairs dp and dp + 4 share the call with
    counter = (global_row, t, (dp & 3) | (dp >> 3) << 2, STREAM_EPS | it << 8); pair dp takes words (2 s, 2 s + 1), s = (dp >> 2) & 1,
    whose two Box-Muller outputs feed dims 2 dp and 2 dp + 1 (csrc/rollout_env.h: eps_group / eps_sub)."""
    cand_hi = n_global if cand_hi is None else cand_hi
    n = cand_hi - cand_lo
    mi = np.arange(m)[:, None, None]
    ni = np.arange(cand_lo, cand_hi)[None, :, None]
    j = np.arange(p)[None, None, :]
    row = ((mi * n_global + ni) * p + j).astype(np.uint32)      # [m,n,p]
    ndp = (D + 1) // 2
    out = np.empty((H, m, n, p, 2 * ndp), np.float32)
    for t in range(H):
        calls = {}
        for dp in range(ndp):
            group, sub = (dp & 3) | ((dp >> 3) << 2), (dp >> 2) & 1
            if group not in calls:
                ctr = _ctr(row, np.uint32(t), np.uint32(group), np.uint32(STREAM_EPS | (it << 8)))
                calls[group] = philox4x32_10(ctr, _key(seed, call, row.shape))
            r = calls[group]
            z0, z1 = box_muller(u01(r[..., 2 * sub]), u01(r[..., 2 * sub + 1]))
            out[t, ..., 2 * dp] = z0
            out[t, ..., 2 * dp + 1] = z1
    return out[..., :D]


def truncated_normals(seed, call, it, m, n_global, H, A, max_attempts=64):
    """Standard normals truncated to |z| < 2 by rejection (TF's
    TruncatedNormalDistribution, kTruncateValue = 2), [m, n_global, H, A].
    counter = (elem_lo, attempt, elem_hi, STREAM_ACT | it << 8); each attempt
    yields 4 candidates (two Box-Muller pairs), the first accepted one is used."""
    L = np.arange(m * n_global * H * A, dtype=np.uint64)
    lo = (L & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (L >> np.uint64(32)).astype(np.uint32)
    out = np.full(L.shape, np.nan, np.float32)
    todo = np.ones(L.shape, bool)
    for attempt in range(max_attempts):
        if not todo.any():
            break
        idx = np.nonzero(todo)[0]
        ctr = _ctr(lo[idx], np.uint32(attempt), hi[idx], np.uint32(STREAM_ACT | (it << 8)))
        r = philox4x32_10(ctr, _key(seed, call, idx.shape))
        za, zb = box_muller(u01(r[..., 0]), u01(r[..., 1]))
        zc, zd = box_muller(u01(r[..., 2]), u01(r[..., 3]))
        for zc_ in (za, zb, zc, zd):
            sel = np.isnan(out[idx]) & (np.abs(zc_) < np.float32(2.0))
            out[idx[sel]] = zc_[sel]
        todo[idx] = np.isnan(out[idx])
    return out.reshape(m, n_global, H, A)


def rs_uniforms(seed, call, m, n_global, H, A):
    """U[-1,1) draws [m, n_global, H, A]: a = 2 u - 1, counter = (elem_lo, 0, elem_hi, STREAM_UNI)."""
    L = np.arange(m * n_global * H * A, dtype=np.uint64)
    lo = (L & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (L >> np.uint64(32)).astype(np.uint32)
    ctr = _ctr(lo, np.uint32(0), hi, np.uint32(STREAM_UNI))
    r = philox4x32_10(ctr, _key(seed, call, L.shape))
    u = u01(r[..., 0])
    return (np.float32(2.0) * u - np.float32(1.0)).reshape(m, n_global, H, A)
