#!/usr/bin/env python3
"""Golden vectors for SURVEY.md 8f-2 (the future-window builder), produced by the REFERENCE ITSELF.

`ModelSampleProcessor.process_samples` (/root/reference/cadm/samplers/model_sample_processor.py:23-127) is pure numpy /
scipy; its module chain merely IMPORTS `tensorflow` (cadm/utils/tensor_utils.py:1), `pyprind` (cadm/samplers/base.py:6)
and, through cadm/utils/utils.py and cadm/logger, a few more packages this image lacks.  None of them is touched on this
path, so they are satisfied with EMPTY placeholder modules (types.ModuleType with no attributes that do anything): every
arithmetic instruction that produces the vectors below is the reference's own code, run unchanged from /root/reference.

Run in the build container only (needs /root/reference):   python tests/golden/make_f2_golden.py
Writes tests/golden/f2_windows.npz = inputs (ragged paths, flattened) + the reference's outputs.  Only that data file
travels; nothing of the reference's source does.
"""
import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


class _Anything(types.ModuleType):
    """Import-only placeholder: attribute access yields another placeholder, calling anything raises."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        child = _Anything(self.__name__ + "." + name)
        setattr(self, name, child)
        return child

    def __call__(self, *a, **k):
        raise RuntimeError("placeholder module %s was CALLED: the golden would not be the reference's arithmetic" % self.__name__)


def import_reference_processor():
    for name in ("tensorflow", "pyprind", "gym", "gym.spaces", "mujoco_py", "baselines", "tensorboardX", "mpi4py"):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = _Anything(name)
    sys.path.insert(0, REF)
    from cadm.samplers.model_sample_processor import ModelSampleProcessor   # noqa: E402  (the reference, unchanged)
    return ModelSampleProcessor


def make_paths(rng, lengths, D, A, Hh):
    paths = []
    for L in lengths:
        paths.append(dict(observations=rng.standard_normal((L, D)), actions=rng.uniform(-1, 1, (L, A)),
                          rewards=rng.standard_normal(L), cp_obs=rng.standard_normal((L, D * Hh)),
                          cp_act=rng.uniform(-1, 1, (L, A * Hh))))
    return paths


CASES = {   # name: (path lengths, D, A, Hh, F)   -- short paths (< F + 1) exercise the zero padding (:62-68)
    "mixed": ([23, 11, 4, 30, 2, 12], 5, 2, 3, 10),
    "all_short": ([3, 5, 1, 10], 4, 3, 2, 10),
    "f4": ([9, 5, 6, 2, 17], 3, 1, 2, 4),
    "exact": ([11, 12], 18, 6, 10, 10),
}


def main():
    MSP = import_reference_processor()
    out = {}
    for name, (lengths, D, A, Hh, F) in CASES.items():
        rng = np.random.default_rng(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
        paths = make_paths(rng, lengths, D, A, Hh)
        flat = {k: np.concatenate([p[k] for p in paths], axis=0) for k in ("observations", "actions", "cp_obs", "cp_act")}
        proc = MSP(context=True, future_length=F, max_path_length=max(lengths))
        res = proc.process_samples([{k: v.copy() for k, v in p.items()} for p in paths], log=False)
        out[name + "/meta"] = np.array([D, A, Hh, F], np.int64)
        out[name + "/lengths"] = np.array(lengths, np.int64)
        for k, v in flat.items():
            out[name + "/in/" + k] = v
        out[name + "/in/rewards"] = np.concatenate([p["rewards"] for p in paths], axis=0)
        for k in ("concat_obs", "concat_act", "concat_next_obs", "concat_bool", "cp_observations", "cp_actions", "observations",
                  "next_observations", "actions", "timesteps", "rewards", "returns"):
            out[name + "/out/" + k] = np.asarray(res[k])
    np.savez_compressed(os.path.join(HERE, "f2_windows.npz"), **out)
    print("wrote f2_windows.npz:", {k: v.shape for k, v in out.items() if "/out/" in k and k.startswith("mixed")})


if __name__ == "__main__":
    main()
