"""ModelSampleProcessor -- MI355X drop-in for the reference's training-set builder
(/root/reference/cadm/samplers/model_sample_processor.py:6-127).

Same constructor kwargs and the same `process_samples(paths, log, log_prefix, itr)` -> samples_data dict.  The heavy part
of the `context` branch -- exploding every path into future windows (concat_obs / concat_act / concat_next_obs /
concat_bool, :58-97) -- runs on the device (`cadm_build_windows`, cadm_amd/csrc/windows.hip): one H2D of the concatenated
paths, one kernel, and either one D2H (numpy out, the reference's contract) or none (`as_device=True`: torch tensors that
`fit` can take as they are).  The small per-path bookkeeping (discounted returns, the plain transition arrays) stays numpy.
Logging (`log != False`) goes to the injected logger (cadm_amd.utils.log)."""
import ctypes as ct

import numpy as np
import torch

from .. import _lib
from ..utils import log as logger


def _discount_cumsum(x, discount):
    """cadm/utils/tensor_utils.py:217-221: y[t] = x[t] + discount * y[t+1] (what scipy.signal.lfilter evaluates)."""
    y = np.zeros(len(x), np.float64)
    acc = 0.0
    for t in range(len(x) - 1, -1, -1):
        acc = x[t] + discount * acc
        y[t] = acc
    return y


class ModelSampleProcessor(object):
    def __init__(self, discount=0.99, max_path_length=200, recurrent=False, context=False, writer=None, future_length=10,
                 device=None):
        self.discount = discount
        self.max_path_length = max_path_length
        self.recurrent = recurrent
        self.context = context
        self.writer = writer
        self.future_length = future_length
        self.device = device

    def process_samples(self, paths, log=False, log_prefix="", itr=None, as_device=False):
        assert len(paths) > 0
        for path in paths:                                                        # :37-41
            path["returns"] = _discount_cumsum(path["rewards"], self.discount)
        self._log_path_stats(paths, log, log_prefix, itr)
        # tensor_utils.concat_tensor_list (cadm/utils/tensor_utils.py:119-123): recurrent -> np.array(list), i.e. one leading
        # axis per path (paths of equal length; numpy refuses ragged ones), else concatenation along the steps
        cat = (lambda xs: np.array(xs)) if self.recurrent else (lambda xs: np.concatenate(xs, axis=0))
        data = dict(observations=cat([p["observations"][:-1] for p in paths]),                              # :45-51
                    next_observations=cat([p["observations"][1:] for p in paths]),
                    actions=cat([p["actions"][:-1] for p in paths]),
                    timesteps=np.concatenate([np.arange(len(p["observations"]) - 1) for p in paths], axis=0),
                    rewards=cat([p["rewards"][:-1] for p in paths]),
                    returns=cat([p["returns"] for p in paths]))
        if self.context:
            win, rows = self._windows(paths, as_device)
            if self.recurrent:          # the flat [N, .] device result, cut per path
                if len(set(int(r) for r in rows)) != 1:
                    raise ValueError("recurrent=True needs paths of equal length (np.array of ragged per-path arrays)")
                win = {k: v.reshape((len(paths), int(rows[0])) + tuple(v.shape[1:])) for k, v in win.items()}
            data.update(win)
        return data

    # ------------------------------------------------------------------ device part
    def _windows(self, paths, as_device):
        if not torch.cuda.is_available():
            raise _lib.CadmError("no HIP device visible: the window builder runs only on the GPU (libcadm_hip.so); "
                                 "there is no CPU fallback")
        lib = _lib.load()
        F = self.future_length
        dev = torch.device(self.device if self.device is not None else "cuda:%d" % torch.cuda.current_device())
        obs0 = np.asarray(paths[0]["observations"])
        dtype = np.float32 if obs0.dtype == np.float32 else np.float64          # the reference works in float64
        D, A = obs0.shape[1], np.asarray(paths[0]["actions"]).shape[1]
        Dh, Ah = np.asarray(paths[0]["cp_obs"]).shape[1], np.asarray(paths[0]["cp_act"]).shape[1]
        lens = np.array([len(p["observations"]) for p in paths], np.int64)
        rows = np.maximum(lens, F + 1) - 1                                        # :60-63,70
        path_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        row_path = np.repeat(np.arange(len(paths), dtype=np.int32), rows)
        row_step = (np.arange(int(rows.sum()), dtype=np.int64) - np.repeat(np.cumsum(rows) - rows, rows)).astype(np.int32)
        N = int(rows.sum())
        up = lambda key: torch.as_tensor(np.ascontiguousarray(np.concatenate([np.asarray(p[key], dtype) for p in paths], axis=0))).to(dev)
        t_obs, t_act, t_cpo, t_cpa = up("observations"), up("actions"), up("cp_obs"), up("cp_act")
        ti = lambda x: torch.as_tensor(x).to(dev)
        d_off, d_rp, d_rs = ti(path_off), ti(row_path), ti(row_step)
        tdt = torch.float32 if dtype == np.float32 else torch.float64
        out = {k: torch.empty((N, w), dtype=tdt, device=dev) for k, w in (("concat_obs", F * D), ("concat_act", F * A),
                                                                         ("concat_next_obs", F * D), ("concat_bool", F),
                                                                         ("cp_observations", Dh), ("cp_actions", Ah))}
        P = _lib.ptr
        stream = ct.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev):
            _lib.check(lib.cadm_build_windows(P(t_obs), P(t_act), P(t_cpo), P(t_cpa), t_obs.element_size(), D, A, Dh, Ah, P(d_off),
                                              P(d_rp), P(d_rs), N, F, P(out["concat_obs"]), P(out["concat_act"]),
                                              P(out["concat_next_obs"]), P(out["concat_bool"]), P(out["cp_observations"]),
                                              P(out["cp_actions"]), stream), "cadm_build_windows")
        # the reference also leaves the zero padding in the caller's path dicts (:64-68); keep that visible side effect
        for p, L in zip(paths, lens):
            rem = max(F + 1 - int(L), 0)
            if rem:
                for key in ("observations", "actions", "cp_obs", "cp_act"):
                    p[key] = np.concatenate([p[key], np.zeros((rem, np.asarray(p[key]).shape[1]))], axis=0)
        if not as_device:
            # the reference builds the windows in float64 whenever zero padding is concatenated (np.zeros): numpy out is
            # float64 like its output; device tensors keep the paths' dtype
            out = {k: v.cpu().numpy().astype(np.float64, copy=False) for k, v in out.items()}
        return out, rows

    def _log_path_stats(self, paths, log, log_prefix, itr):                      # cadm/samplers/base.py:222-250
        undiscounted = [float(np.sum(p["rewards"])) for p in paths]
        w = self.writer
        if log == "reward":
            logger.logkv(log_prefix + "AverageReturn", np.mean(undiscounted))
            if w is not None:
                w.add_scalar("log/AverageReturn", np.mean(undiscounted))
        elif log == "all" or log is True:
            stats = (("AverageDiscountedReturn", np.mean([p["returns"][0] for p in paths])), ("AverageReturn", np.mean(undiscounted)),
                     ("NumTrajs", len(paths)), ("StdReturn", np.std(undiscounted)), ("MaxReturn", np.max(undiscounted)),
                     ("MinReturn", np.min(undiscounted)))
            for k, v in stats:
                logger.logkv(log_prefix + k, v)
            if w is not None:                                                     # TensorBoard scalars, base.py:243-249
                for k, v in stats:
                    w.add_scalar("log/" + k, v, itr)
