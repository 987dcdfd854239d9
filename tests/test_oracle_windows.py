"""SURVEY.md 8f-2 on the CPU: the oracle restatement of the reference's future-window builder against golden vectors that
the reference's OWN code produced (tests/golden/make_f2_golden.py runs /root/reference/cadm/samplers/
model_sample_processor.py:23-127 unchanged) -- a pinned oracle.  Bit-exact: it is index / byte work."""
import os

import numpy as np
import pytest

from oracle import windows as ow

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "f2_windows.npz"))
CASES = sorted({k.split("/")[0] for k in GOLD.files})


def paths_of(case):
    D, A, Hh, F = (int(v) for v in GOLD[case + "/meta"])
    lengths = GOLD[case + "/lengths"]
    offs = np.concatenate([[0], np.cumsum(lengths)])
    paths = []
    for i, L in enumerate(lengths):
        sl = slice(offs[i], offs[i + 1])
        paths.append({k: GOLD[case + "/in/" + k][sl] for k in ("observations", "actions", "cp_obs", "cp_act", "rewards")})
    return paths, (D, A, Hh, F)


@pytest.mark.parametrize("case", CASES)
def test_oracle_window_builder_equals_the_reference(case):
    paths, (D, A, Hh, F) = paths_of(case)
    got = ow.process_samples(paths, F)
    for k in ("concat_obs", "concat_act", "concat_next_obs", "concat_bool", "cp_observations", "cp_actions", "observations",
              "next_observations", "actions", "timesteps", "rewards"):
        np.testing.assert_array_equal(got[k], GOLD[case + "/out/" + k], err_msg="%s/%s" % (case, k))
    np.testing.assert_allclose(got["returns"], GOLD[case + "/out/returns"], rtol=1e-15, atol=0)


@pytest.mark.parametrize("case", CASES)
def test_reference_quirks_are_in_the_golden(case):
    """What the restatement guide says about the reference holds in the reference's own output: row 0
    of every path is masked out completely; a short path keeps max(i - remainder, 0) futures on its i-th last row."""
    paths, (D, A, Hh, F) = paths_of(case)
    cb = GOLD[case + "/out/concat_bool"]
    row = 0
    for p in paths:
        L = max(len(p["observations"]), F + 1)
        assert cb[row].sum() == 0
        rem = max(F + 1 - len(p["observations"]), 0)
        for i in range(1, F):
            assert cb[row + (L - 1) - i].sum() == max(i - rem, 0) or (L - 1) - i == 0
        row += L - 1
    assert row == cb.shape[0]
