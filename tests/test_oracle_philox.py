"""CPU: Philox4x32-10 against Random123's published known-answer vectors, and the stream layout."""
import numpy as np

from oracle import philox as ph


def _h(c, k):
    return [int(x) for x in ph.philox4x32_10(np.array(c, dtype=np.uint32), np.array(k, dtype=np.uint32))]


def test_random123_kat():
    # Random123 kat_vectors: philox4x32 10
    assert _h([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _h([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _h([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_truncated_normals_statistics_and_bounds():
    z = ph.truncated_normals(7, 1, 0, 1, 2000, 10, 6)
    assert np.abs(z).max() < 2.0 and not np.isnan(z).any()
    assert abs(z.mean()) < 0.01 and abs(z.std() - 0.8796) < 0.01      # std of N(0,1) truncated to |z| < 2
    z2 = ph.truncated_normals(7, 1, 1, 1, 2000, 10, 6)
    assert not np.array_equal(z, z2)                                   # iteration changes the stream


def test_eps_shard_independence():
    """The noise of a candidate does not depend on how candidates are sharded over GPUs."""
    full = ph.eps_normals(3, 9, 2, 2, 12, 4, 3, 18)
    part = ph.eps_normals(3, 9, 2, 2, 12, 4, 3, 18, cand_lo=4, cand_hi=8)
    np.testing.assert_array_equal(full[:, :, 4:8], part)
    assert abs(full.mean()) < 0.05 and abs(full.std() - 1.0) < 0.05


def test_rs_uniform_range():
    u = ph.rs_uniforms(1, 2, 2, 50, 5, 6)
    assert u.min() >= -1.0 and u.max() < 1.0 and abs(u.mean()) < 0.02
