"""SURVEY.md 8f-2 on the device: `cadm_build_windows` / the drop-in ModelSampleProcessor against golden vectors produced by
the reference's own `process_samples` (tests/golden/make_f2_golden.py) and against the (pinned) oracle restatement on a
larger ragged batch.  Index / byte work: bit-exact, for float64 (the reference's dtype) and float32."""
import os

import numpy as np
import pytest
import torch

from cadm_amd.samplers import ModelSampleProcessor
from oracle import windows as ow

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "f2_windows.npz"))
CASES = sorted({k.split("/")[0] for k in GOLD.files})
KEYS = ("concat_obs", "concat_act", "concat_next_obs", "concat_bool", "cp_observations", "cp_actions", "observations",
        "next_observations", "actions", "timesteps", "rewards")


def paths_of(case, dtype=np.float64):
    D, A, Hh, F = (int(v) for v in GOLD[case + "/meta"])
    lengths = GOLD[case + "/lengths"]
    offs = np.concatenate([[0], np.cumsum(lengths)])
    return [{k: GOLD[case + "/in/" + k][offs[i]:offs[i + 1]].astype(dtype) for k in ("observations", "actions", "cp_obs", "cp_act", "rewards")}
            for i in range(len(lengths))], F


@pytest.mark.parametrize("case", CASES)
def test_device_windows_equal_the_reference_golden(gpu, case):
    paths, F = paths_of(case)
    got = ModelSampleProcessor(context=True, future_length=F).process_samples(paths)
    for k in KEYS:
        np.testing.assert_array_equal(got[k], GOLD[case + "/out/" + k], err_msg="%s/%s" % (case, k))
    np.testing.assert_allclose(got["returns"], GOLD[case + "/out/returns"], rtol=1e-15, atol=0)
    # the reference leaves its zero padding in the caller's path dicts (:64-68)
    for p, L in zip(paths, GOLD[case + "/lengths"]):
        assert len(p["observations"]) == max(int(L), F + 1)


def test_device_windows_large_ragged_batch_float32_and_device_output(gpu):
    rng = np.random.default_rng(5)
    D, A, Hh, F = 18, 6, 10, 10
    lengths = list(rng.integers(1, 201, size=64)) + [1, 2, 11, 200]
    paths = [dict(observations=rng.standard_normal((L, D)).astype(np.float32), actions=rng.uniform(-1, 1, (L, A)).astype(np.float32),
                  rewards=rng.standard_normal(L), cp_obs=rng.standard_normal((L, D * Hh)).astype(np.float32),
                  cp_act=rng.uniform(-1, 1, (L, A * Hh)).astype(np.float32)) for L in lengths]
    ref = ow.process_samples([{k: np.asarray(v, np.float64) for k, v in p.items()} for p in paths], F)
    dev = ModelSampleProcessor(context=True, future_length=F).process_samples([dict(p) for p in paths], as_device=True)
    for k in ("concat_obs", "concat_act", "concat_next_obs", "concat_bool", "cp_observations", "cp_actions"):
        assert isinstance(dev[k], torch.Tensor) and dev[k].is_cuda and dev[k].dtype == torch.float32
        np.testing.assert_array_equal(dev[k].cpu().numpy().astype(np.float64), ref[k], err_msg=k)
    assert dev["concat_obs"].shape[0] == int(np.sum(np.maximum(lengths, F + 1) - 1))
    # empty-history model (vanilla shape: Hh = 0 columns) goes through the same kernel
    for p in paths:
        p["cp_obs"], p["cp_act"] = p["cp_obs"][:, :0], p["cp_act"][:, :0]
    out = ModelSampleProcessor(context=True, future_length=F).process_samples(paths)
    np.testing.assert_array_equal(out["concat_obs"].astype(np.float64), ref["concat_obs"])
    assert out["cp_observations"].shape == (ref["concat_obs"].shape[0], 0)
