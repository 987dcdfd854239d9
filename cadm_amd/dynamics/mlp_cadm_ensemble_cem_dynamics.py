"""MLPEnsembleCEMDynamicsModel (PE-TS + CaDM) -- MI355X drop-in for the reference class
/root/reference/cadm/dynamics/mlp_cadm_ensemble_cem_dynamics.py:12.

Same constructor kwargs (:26-54) and public methods -- get_action (:344), get_context_pred
(:369), fit (:382), save (:571), load (:579), compute_normalization (:590),
get_normalization_stats (:604) -- so `run_cadm_pets.py` can build it unchanged.  What was one
TensorFlow graph behind `sess.run` is here libcadm_hip.so (hand-written HIP for gfx950)
reached through cadm_amd.engine.HipEngine; host code keeps the reference's numpy-level data
handling (dataset growth, float64 statistics, bootstrap indices, early stopping).

Differences a caller can see (all opt-in except the first):
  * `env` may be a reference env object, a NormalizedEnv wrapper or a cadm_amd.envs.EnvSpec;
    its class selects the compiled-in closures (cadm_amd/envs.py);
  * extra kwargs: `reference_quirks` (default True: reproduce the context-layout quirks Q1/Q2
    of core/utils.py:434-435), `seed` (device Philox key; the reference never seeds TF),
    `device`, `process_group` (OPT-IN: shard candidates over the ranks of that torch.distributed group; every
    rank must then call get_action with the same observations -- None, the default, never shards.  The GUARD against ranks
    that drift apart (different obs / history / warm start -> different elite sets -> silently different plans): the first
    `check_replicated_calls` (2) sharded calls AND every `check_replicated_every`-th (256th) one after them compare a checksum
    of the inputs with one extra all-reduce and raise on a mismatch; all other calls run only the path's own collectives.
    `check_replicated_every=0` switches the periodic check OFF -- then nothing detects divergence after the first calls);
  * `predict(obs, act, cp_obs, cp_act)` -- thin alias the north-star asks for: one-step mean
    prediction of every ensemble member (the reference has no public predict).
"""
import time
from collections import OrderedDict

import joblib
import numpy as np
import torch

from .. import planner as _planner
from ..engine import HipEngine
from ..envs import resolve_env_kind
from ..utils import log as logger

_ACTIVATIONS = (None, "relu", "tanh", "sigmoid", "softmax", "swish")   # reference :17-24


class FitIndexStream(object):
    """Every random index `fit` consumes, in the order the reference draws them from the global `np.random`
    (/root/reference/cadm/dynamics/mlp_cadm_ensemble_cem_dynamics.py:444, :465, :472-474 + :483):

        permutation(N)             -> [N]        train / validation split of the windows
        bootstrap(E, n_train)      -> [E, n]     each member's bootstrap sample of training rows (E > 1 only)
        epoch_order(E, n_train)    -> [E, n]     `shuffle_rows`: per-member argsort of uniforms, once per epoch

    The default draws them on the host from a numpy Generator.  A caller (or a parity test) may pass any object with
    these three methods as `fit(..., index_stream=...)`, e.g. `ReplayIndexStream` with arrays captured elsewhere."""

    def __init__(self, rng):
        self.rng = rng

    def permutation(self, n):
        return self.rng.permutation(n)

    def bootstrap(self, E, n_train):
        return self.rng.integers(0, n_train, size=(E, n_train))

    def epoch_order(self, E, n_train):
        return np.argsort(self.rng.uniform(size=(E, n_train)), axis=-1)


class RecordingIndexStream(object):
    """Wraps a stream and keeps what it handed out (`.log`: list of (kind, array))."""

    def __init__(self, inner):
        self.inner, self.log = inner, []

    def _rec(self, kind, arr):
        self.log.append((kind, np.array(arr)))
        return arr

    def permutation(self, n):
        return self._rec("permutation", self.inner.permutation(n))

    def bootstrap(self, E, n_train):
        return self._rec("bootstrap", self.inner.bootstrap(E, n_train))

    def epoch_order(self, E, n_train):
        return self._rec("epoch_order", self.inner.epoch_order(E, n_train))


class ReplayIndexStream(object):
    """Replays a recorded log; raises if `fit` asks for something else than what was recorded."""

    def __init__(self, log):
        self.log, self.pos = list(log), 0

    def _next(self, kind, shape):
        if self.pos >= len(self.log) or self.log[self.pos][0] != kind or tuple(self.log[self.pos][1].shape) != tuple(shape):
            raise ValueError("index stream exhausted or out of order at draw %d (%s %r)" % (self.pos, kind, shape))
        arr = self.log[self.pos][1]
        self.pos += 1
        return arr

    def permutation(self, n):
        return self._next("permutation", (n,))

    def bootstrap(self, E, n_train):
        return self._next("bootstrap", (E, n_train))

    def epoch_order(self, E, n_train):
        return self._next("epoch_order", (E, n_train))


class MLPEnsembleCEMDynamicsModel(object):
    """Probabilistic-ensemble MLP dynamics model with a context encoder and a CEM / RS planner."""

    def __init__(self,
                 name,
                 env,
                 hidden_sizes=(200, 200, 200, 200),
                 hidden_nonlinearity="swish",       # (the reference's default, tf.nn.relu, is not a key of its own `_activations` table:
                 output_nonlinearity=None,          #  dynamics.py:30 vs :104 raises KeyError on a defaulted call; every script passes 'swish')
                 batch_size=128,
                 learning_rate=0.001,
                 normalize_input=True,
                 optimizer=None,
                 valid_split_ratio=0.2,
                 rolling_average_persitency=0.99,
                 n_forwards=30,
                 n_candidates=2500,
                 ensemble_size=5,
                 n_particles=20,
                 use_cem=False,
                 deterministic=False,
                 weight_decays=(0., 0., 0., 0., 0.),
                 weight_decay_coeff=0.0,
                 cp_hidden_sizes=(256, 128, 64),
                 context_weight_decays=(0., 0., 0., 0.),
                 context_out_dim=10,
                 context_hidden_nonlinearity="relu",
                 history_length=10,
                 future_length=10,
                 state_diff=False,
                 back_coeff=0.0,
                 # ---- extensions (not in the reference) ----
                 reference_quirks=True,
                 seed=0,
                 device=None,
                 process_group=None,
                 check_replicated_calls=2,
                 check_replicated_every=256,
                 engine_lib=None,
                 ):
        self.env = env
        self.name = name
        self._dataset = None

        if hidden_nonlinearity not in _ACTIVATIONS or output_nonlinearity not in _ACTIVATIONS:
            raise KeyError("unknown nonlinearity %r / %r" % (hidden_nonlinearity, output_nonlinearity))
        if hidden_nonlinearity == "softmax" or output_nonlinearity is not None:
            # reference: `_activations` (dynamics.py:17-24) offers both; `output_nonlinearity` would be applied to BOTH heads and to the
            # context encoder's output (core/utils.py:327,333,609), 'softmax' row-wise over a hidden layer's units.  No script of the
            # reference passes either (run_cadm_pets.py:221, run_pets.py: 'swish' / None).
            raise NotImplementedError(
                "the HIP kernels implement hidden_nonlinearity in (swish, relu, tanh, sigmoid, None) and output_nonlinearity=None "
                "(reference: _activations, cadm/dynamics/mlp_cadm_ensemble_cem_dynamics.py:17-24; its scripts pass 'swish' / None: "
                "run_cadm_pets.py:221); got hidden_nonlinearity=%r, output_nonlinearity=%r" % (hidden_nonlinearity, output_nonlinearity))
        # context_hidden_nonlinearity is accepted and ignored exactly like the reference
        # (always ReLU: dynamics.py:49 vs :141-156, layers.py:34).
        if optimizer is not None and getattr(optimizer, "__name__", "") not in ("AdamOptimizer",):
            raise NotImplementedError("only Adam (TF1 semantics) is implemented")

        self.deterministic = deterministic
        self.n_forwards = n_forwards
        self.n_candidates = n_candidates
        self.use_cem = use_cem
        self.weight_decays = weight_decays
        self.weight_decay_coeff = weight_decay_coeff
        self.normalization = None
        self.normalize_input = normalize_input
        self.batch_size = batch_size
        self.learning_rate = learning_rate
        self.valid_split_ratio = valid_split_ratio
        self.rolling_average_persitency = rolling_average_persitency
        self.ensemble_size = ensemble_size
        self.n_particles = n_particles
        self.cp_hidden_sizes = cp_hidden_sizes
        self.context_out_dim = context_out_dim
        self.history_length = history_length
        self.future_length = future_length
        self.context_weight_decays = context_weight_decays
        self.state_diff = state_diff
        self.back_coeff = back_coeff

        self.obs_space_dims = obs_space_dims = env.observation_space.shape[0]
        self.proc_obs_space_dims = env.proc_observation_space_dims
        if len(env.action_space.shape) == 0:
            self.action_space_dims = env.action_space.n
            self.discrete = True
        else:
            self.action_space_dims = env.action_space.shape[0]
            self.discrete = False

        if n_particles % ensemble_size != 0:
            raise ValueError("n_particles must be a multiple of ensemble_size (core/utils.py:447 int(p/E))")

        self.env_kind = resolve_env_kind(env)
        self.seed = int(seed)
        self._call = 0
        self._checked_sig = None        # shape set of the last get_action whose inputs were checked
        self._group = process_group
        self._shard1 = None
        self._check_replicated_left = int(check_replicated_calls)   # sharded get_action calls that still verify replicated inputs
        self._check_replicated_every = int(check_replicated_every)  # ... and every N-th sharded call after those (0: never again)
        self._sharded_calls = 0
        self.engine = HipEngine(self.env_kind, ensemble_size, n_particles, obs_space_dims, self.action_space_dims,
                                self.proc_obs_space_dims, context_out_dim, hidden_sizes, n_forwards,
                                deterministic=deterministic, discrete=self.discrete,
                                reference_quirks=reference_quirks, history_length=history_length,
                                cp_hidden_sizes=cp_hidden_sizes, back_model=back_coeff > 0.0, device=device,
                                hidden_nonlinearity=hidden_nonlinearity,
                                lib=engine_lib)      # engine_lib: developer builds only (tools/ab.sh); None = the product library
        # tf.global_variables_initializer() equivalent (mb_trainer.py:164)
        self.engine.init_weights(np.random.default_rng(self.seed))
        self._train_ready = False
        self._stats_dirty = True
        self._dist_failed = False
        # The rollout kernel is compiled per geometry: the reference defaults are in the library, anything else
        # (`--hidden_size`, `--context_out_dim`, depth, nonlinearity) is built now (cadm_amd.jit, ~30 s once, then cached),
        # and a launch that cannot fit the hardware (LDS for this horizon) raises here -- not at the first get_action.
        self.engine.ensure_rollout(None, 1, max(1, n_candidates))

    # ------------------------------------------------------------------ planning
    def _push_stats(self):
        if self._stats_dirty:
            keys = ("obs_mean", "obs_std", "act_mean", "act_std", "delta_mean", "delta_std", "cp_obs_mean",
                    "cp_obs_std", "cp_act_mean", "cp_act_std", "back_delta_mean", "back_delta_std")
            self.engine.set_stats(dict(zip(keys, self._stats12())))
            self._stats_dirty = False

    def _next_call(self):
        self._call += 1
        return self._call & 0xFFFFFFFF

    def _check_planner_inputs(self, obs, cp_obs, cp_act, cem_init_mean, cem_init_var):
        """Host-side shape checks (the reference would raise a TF shape error; raw device pointers would not)."""
        D, A, Hh, H = self.obs_space_dims, self.action_space_dims, self.history_length, self.n_forwards
        shp = lambda x: tuple(int(v) for v in x.shape)
        m = shp(obs)[0]
        want = {"obs": (obs, (m, D)), "cem_init_mean": (cem_init_mean, (m, H, A)), "cem_init_var": (cem_init_var, (m, H, A))}
        if self.context_out_dim > 0:
            want["cp_obs"] = (cp_obs, (m, D * Hh))
            want["cp_act"] = (cp_act, (m, A * Hh))
            if cp_obs is None or cp_act is None:
                raise ValueError("get_action: cp_obs and cp_act are required for a context model")
        for name, (x, expect) in want.items():
            if x is not None and shp(x) != expect:
                raise ValueError("get_action: %s has shape %r, expected %r" % (name, shp(x), expect))
        if (cem_init_mean is None) != (cem_init_var is None):
            raise ValueError("get_action: cem_init_mean and cem_init_var must be given together")

    def _sharding(self):
        """Candidate shard of this rank.  The sharded planner is the library's loop on every rank; what is negotiated on first use is
        only its collective: the ctx's own RCCL communicator when EVERY rank of the group can build one (backend nccl, one GPU per
        rank), else -- all ranks together -- torch.distributed's all-gather through `cadm_dist_init_external`.  (The second element,
        always True, is kept for callers that asked "is the plan one library call?")"""
        if self._group is None:         # never sharded (the default): nothing to negotiate
            if self._shard1 is None or self._shard1.n != self.n_candidates:
                self._shard1 = _planner.Shard(self.n_candidates)
            return self._shard1, True
        shard = _planner.Shard.from_group(self.n_candidates, self._group)
        if shard.world > 1 and self.engine.dist_world == 1:
            import torch.distributed as dist
            backend = dist.get_backend(self._group)
            ok = 0.0
            if backend == "nccl" and not self._dist_failed:
                ok = 1.0
                try:       # in-library RCCL communicator: the whole sharded planner stays on the stream
                    self.engine.dist_init(self._group)
                except Exception as exc:
                    ok = 0.0
                    logger.log("cadm_amd: in-library RCCL init failed on this rank (%s)" % exc)
            # every rank must take the SAME collective, or some would wait in ncclAllGather and others in torch's
            flag = torch.tensor([ok], device=self.engine.device if backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self._group)
            if float(flag.item()) < 1.0:
                if self.engine.dist_world > 1:
                    self.engine.dist_destroy()
                if backend == "nccl":
                    self._dist_failed = True
                    logger.log("cadm_amd: a rank failed in-library RCCL init; all ranks plug torch.distributed's all_gather into the same planner")
                self.engine.dist_init_external(self._group)
        return shard, True

    def _replication_check_due(self, peek=False):
        """Sharded calls only.  Counts the call (unless `peek`: the same call asking again further down) and says whether this one
        verifies that every rank was fed the same inputs: the first `check_replicated_calls` calls, then every
        `check_replicated_every`-th.  Every rank makes the same sequence of calls, so every rank takes the same decision."""
        if not peek:
            self._sharded_calls += 1
        if self._check_replicated_left > 0:
            return True
        return self._check_replicated_every > 0 and self._sharded_calls % self._check_replicated_every == 0

    def get_action(self, obs, cp_obs, cp_act, cem_init_mean=None, cem_init_var=None):
        """reference :344-367.  CEM: returns the whole plan [m,H,A]; RS: the first action [m,A]
        (ints [m] for discrete envs).  Continuous outputs are clipped to [-1,1]."""
        if self._stats_dirty:
            self._push_stats()
        nd = np.ndarray
        counted = False
        if (type(obs) is nd and type(cem_init_mean) is nd and type(cem_init_var) is nd and (cp_obs is None or type(cp_obs) is nd)
                and (cp_act is None or type(cp_act) is nd) and obs.shape[0] > 0):
            # the samplers' call, shapes read off the arrays (np.shape / isinstance over five arguments cost ~4 us of a 0.9 ms call)
            sig = (obs.shape, None if cp_obs is None else cp_obs.shape, None if cp_act is None else cp_act.shape, cem_init_mean.shape,
                   cem_init_var.shape)
            if sig == self._checked_sig:
                shard, fused = self._sharding()
                due = False
                if shard.world > 1:      # the call is counted HERE, whatever path it then takes (the torch.distributed
                    due = self._replication_check_due()      # fallback used to re-ask with peek=True without ever having counted)
                    counted = True
                if not due:
                    self._call += 1
                    return self.engine.cem_plan_host((obs, cp_obs, cp_act, cem_init_mean, cem_init_var), self.n_candidates, seed=self.seed,
                                                     call=self._call & 0xFFFFFFFF, shapes=sig)
        m = int(np.shape(obs)[0])
        if m == 0:      # an empty batch of environments: the reference's graph returns empty arrays
            if cem_init_mean is not None:
                return np.zeros((0, self.n_forwards, self.action_space_dims), np.float32)
            return np.zeros((0,), np.int32) if self.discrete else np.zeros((0, self.action_space_dims), np.float32)
        host_in = not any(isinstance(x, torch.Tensor) for x in (obs, cp_obs, cp_act, cem_init_mean, cem_init_var))
        # shape checks once per shape set (the reference would raise a TF shape error; raw device pointers would not)
        sig = tuple(None if x is None else tuple(np.shape(x)) for x in (obs, cp_obs, cp_act, cem_init_mean, cem_init_var))
        if sig != self._checked_sig:
            self._check_planner_inputs(obs, cp_obs, cp_act, cem_init_mean, cem_init_var)
            self._checked_sig = sig
        call = self._next_call()
        shard, fused = self._sharding()
        check_due = shard.world > 1 and self._replication_check_due(peek=counted)
        if host_in and cem_init_mean is not None and not check_due:
            # the hot path of the samplers' loop: one library call from host arrays to the host plan (clipped in the last kernel)
            return self.engine.cem_plan_host((obs, cp_obs, cp_act, cem_init_mean, cem_init_var), self.n_candidates, seed=self.seed, call=call,
                                             shapes=sig)
        if host_in:
            obs, cp_obs, cp_act, cem_init_mean, cem_init_var = self.engine.stage((obs, cp_obs, cp_act, cem_init_mean, cem_init_var))
        if check_due:      # first calls + every N-th: one blocking all-reduce + a host sync
            self._check_replicated_left = max(0, self._check_replicated_left - 1)
            _planner.check_replicated([self.engine._t(x) for x in (obs, cp_obs, cp_act, cem_init_mean) if x is not None], shard)
        if cem_init_mean is not None:
            # the plan lands in a persistent pinned host buffer (written by the last refit kernel): one stream
            # synchronisation instead of an allocation + D2H copy per call
            eng = self.engine
            host = eng.host_out((m, self.n_forwards, self.action_space_dims))
            eng.cem_plan(obs, cp_obs, cp_act, cem_init_mean, cem_init_var, self.n_candidates, seed=self.seed, call=call, out=host)
            torch.cuda.current_stream(eng.device).synchronize()
            action = host.numpy().copy()
            if shard.world > 1 and eng.dist_mismatch():      # (the sharded refit checks every rank's input checksum on every call)
                raise RuntimeError(eng.MISMATCH_MSG)
            return action if self.discrete else np.minimum(np.maximum(action, -1.0), 1.0)
        action = self.engine.rs_plan(obs, cp_obs, cp_act, self.n_candidates, seed=self.seed, call=call)
        action = action.cpu().numpy()
        if not self.discrete:
            action = np.minimum(np.maximum(action, -1.0), 1.0)
        return action

    def get_context_pred(self, cp_obs, cp_act):
        """reference :369-380 -> [E,m,C]."""
        self._push_stats()
        return self.engine.context_forward(cp_obs, cp_act).cpu().numpy()

    def predict(self, obs, act, cp_obs=None, cp_act=None, return_std=False):
        """One-step prediction of every ensemble member (no reference twin; the north-star's `predict()`):
        obs [m,D], act [m,A] (+ history for the CaDM model) -> mean next observation [E,m,D]
        (`obs_postproc(obs, denormalize(mu))`), optionally with the predictive std of the delta [E,m,D]."""
        self._push_stats()
        E = self.ensemble_size
        tile = lambda x: None if x is None else np.tile(np.asarray(x, dtype=np.float32)[None], (E, 1, 1))
        mu, lv = self.engine.predict_heads(tile(obs), tile(act), tile(cp_obs), tile(cp_act))
        stats = self._stats12()
        dmean, dstd = np.asarray(stats[4], np.float32), np.asarray(stats[5], np.float32)
        delta = mu.cpu().numpy() * (dstd + 1e-10) + dmean
        nxt = self.env.obs_postproc(np.broadcast_to(np.asarray(obs, np.float32)[None], delta.shape), delta)
        if return_std:
            if lv is None:
                raise ValueError("a deterministic model has no predictive variance")
            return nxt, np.exp((lv.cpu().numpy() + 2.0 * np.log(dstd)) / 2.0)
        return nxt

    # ------------------------------------------------------------------ training
    def _ensure_train(self):
        if not self._train_ready:
            self.engine.train_configure(self.learning_rate, self.weight_decays, self.context_weight_decays,
                                        self.weight_decay_coeff, self.back_coeff, max_batch=0)
            self._train_ready = True

    def fit(self, obs, act, obs_next, cp_obs, cp_act, future_bool, epochs=1000, compute_normalization=True,
            valid_split_ratio=None, rolling_average_persitency=None, verbose=False, log_tabular=False,
            max_logging=5000, rng=None, index_stream=None, keep_trace=None):
        """reference :382-569.  `rng` (numpy Generator) replaces the reference's global np.random; `index_stream`
        (see FitIndexStream) replaces the draws themselves -- without one, the per-epoch shuffles are drawn ON THE DEVICE
        (an [E, n_train] argsort per epoch is ~1 s of host time and an 80 MB copy at 2 M rows).  `self.last_fit_trace` keeps
        the per-epoch validation losses and, with `keep_trace` (default: only when an index stream is injected), one
        [steps, 3] array of training losses [mse, back_mse, recon] per epoch."""
        D, A, F, Hh = self.obs_space_dims, self.action_space_dims, self.future_length, self.history_length
        assert obs.ndim == 2 and obs.shape[1] == D * F
        assert obs_next.ndim == 2 and obs_next.shape[1] == D * F
        assert act.ndim == 2 and act.shape[1] == A * F
        assert cp_obs.ndim == 2 and cp_obs.shape[1] == D * Hh
        assert cp_act.ndim == 2 and cp_act.shape[1] == A * Hh
        assert future_bool.ndim == 2 and future_bool.shape[1] == F
        if valid_split_ratio is None:
            valid_split_ratio = self.valid_split_ratio
        if rolling_average_persitency is None:
            rolling_average_persitency = self.rolling_average_persitency
        assert 1 > valid_split_ratio >= 0
        injected = index_stream is not None
        if index_stream is None:
            index_stream = FitIndexStream(rng if rng is not None else np.random.default_rng(self.seed + 7919 * (self._call + 1)))
        if keep_trace is None:
            keep_trace = injected

        obs = obs.reshape(-1, D)
        obs_next = obs_next.reshape(-1, D)
        delta = self.env.targ_proc(obs, obs_next)
        back_delta = self.env.targ_proc(obs_next, obs)
        obs = obs.reshape(-1, F * D)
        obs_next = obs_next.reshape(-1, F * D)
        delta = delta.reshape(-1, F * D)
        back_delta = back_delta.reshape(-1, F * D)
        new = dict(obs=obs, act=act, delta=delta, cp_obs=cp_obs, cp_act=cp_act, future_bool=future_bool,
                   obs_next=obs_next, back_delta=back_delta, single_obs=obs[:, :D], single_act=act[:, :A],
                   single_delta=delta[:, :D], single_back_delta=back_delta[:, :D])
        if self._dataset is None:
            self._dataset = new
        else:
            for k, v in new.items():
                self._dataset[k] = np.concatenate([self._dataset[k], v])
        ds = self._dataset
        self.compute_normalization(ds["single_obs"], ds["single_act"], ds["single_delta"], ds["cp_obs"],
                                   ds["cp_act"], ds["single_back_delta"])
        self._push_stats()

        N = ds["obs"].shape[0]
        n_valid = min(int(N * valid_split_ratio), max_logging)
        perm = np.asarray(index_stream.permutation(N))
        # The reference explodes every window into F rows on the host (`_preprocess_inputs`, :676-696: reshape, tile the
        # history F times, mask by future_bool) and feeds numpy batches.  Here the windowed dataset goes to HBM once and
        # a row is the pair (window, future offset): batches gather straight from the windows, the history is never tiled.
        dev = self._upload_dataset(ds)
        train_rows = self._row_index(ds["future_bool"], perm[n_valid:])
        valid_rows = self._row_index(ds["future_bool"], perm[:n_valid]) if n_valid > 0 else None
        return self._fit_loop(dev, train_rows, valid_rows, epochs, rolling_average_persitency, verbose, log_tabular, index_stream,
                              device_shuffle=not injected, keep_trace=keep_trace)

    _BATCH_KEYS = ("obs", "act", "delta", "obs_next", "back_delta", "cp_obs", "cp_act")

    def _upload_dataset(self, ds):
        eng = self.engine
        return {k: eng._t(ds[k]) for k in self._BATCH_KEYS}

    @staticmethod
    def _row_index(future_bool, windows):
        """Rows of `_preprocess_inputs(ds[windows])` as (window id, future offset), in the same order
        (window-major, masked by future_bool)."""
        wl, f = np.nonzero(future_bool[windows] > 0)
        return np.asarray(windows)[wl].astype(np.int64), f.astype(np.int64)

    def _gather_rows(self, dev, w, f):
        """[..] index tensors -> the batch dict of `cadm_train_step` ([.., dim] each)."""
        D, A, F = self.obs_space_dims, self.action_space_dims, self.future_length
        N = dev["obs"].shape[0]
        out = {}
        for k in ("obs", "delta", "obs_next", "back_delta"):
            out[k] = dev[k].view(N, F, D)[w, f]
        out["act"] = dev["act"].view(N, F, A)[w, f]
        out["cp_obs"] = dev["cp_obs"][w]
        out["cp_act"] = dev["cp_act"][w]
        return out

    def _fit_loop(self, dev, train_rows, valid_rows, epochs, persistency, verbose, log_tabular, stream,
                  device_shuffle=False, keep_trace=True):
        """Epoch / batch loop (reference :460-569): bootstrap indices, shuffle_rows, one fused
        fwd/bwd/Adam step per batch, validation, rolling-average early stop."""
        self._ensure_train()
        eng, E = self.engine, self.ensemble_size
        row_w = torch.as_tensor(train_rows[0], device=eng.device)
        row_f = torch.as_tensor(train_rows[1], device=eng.device)
        n_train = int(row_w.shape[0])
        if E > 1:
            bootstrap_idx = np.asarray(stream.bootstrap(E, n_train))            # :465
        else:
            bootstrap_idx = np.tile(np.arange(n_train, dtype="int64"), (E, 1))  # :467
        didx = torch.as_tensor(bootstrap_idx.astype(np.int64), device=eng.device)
        trace = self.last_fit_trace = dict(train=[], valid=[])
        gen = None
        if device_shuffle:      # keyed by the host stream, so a seeded `rng` still fixes the whole fit
            gen = torch.Generator(device=eng.device)
            gen.manual_seed(int(stream.rng.integers(0, 2 ** 62)))
        dev_valid = None
        if valid_rows is not None and valid_rows[0].shape[0] > 0:
            vw = torch.as_tensor(valid_rows[0], device=eng.device)
            vf = torch.as_tensor(valid_rows[1], device=eng.device)
            dev_valid = {k: v[None].expand(E, -1, -1).contiguous() for k, v in self._gather_rows(dev, vw, vf).items()}  # :470,515-521
        rolling, rolling_prev = None, None
        epoch = -1
        for epoch in range(epochs):
            t0 = time.time()
            # shuffle_rows (:472-474,483): independent permutation of every member's index row (argsort of uniforms) --
            # on the device by default; from the host stream when one is injected (parity tests replay the reference's draws)
            if gen is not None:
                order = torch.argsort(torch.rand((E, n_train), device=eng.device, generator=gen), dim=-1)
            else:
                order = torch.as_tensor(np.asarray(stream.epoch_order(E, n_train)).astype(np.int64), device=eng.device)
            didx = torch.gather(didx, 1, order)
            losses = []
            for b in range(int(np.ceil(n_train / self.batch_size))):
                bi = didx[:, b * self.batch_size:(b + 1) * self.batch_size]    # [E,B] row ids
                # the step reads its rows through (row id -> window, offset) inside the kernels: no gathered batch
                losses.append(eng.train_step_rows(dev, self.future_length, row_w, row_f, bi, train=True))
            step_losses = torch.stack(losses).cpu().numpy() if losses else np.zeros((0, 3))
            if keep_trace:
                trace["train"].append(step_losses)          # one [steps, 3] array per epoch
            tl = step_losses.mean(0) if losses else np.zeros(3)
            if dev_valid is not None:
                v_mse, v_back, v_recon = eng.train_step(dev_valid, train=False).cpu().numpy()
                trace["valid"].append((v_mse, v_back, v_recon))
                if verbose:
                    logger.log("Training DynamicsModel - finished epoch %i --"
                               "[Training] mse loss: %.4f  back mse loss: %.4f  recon loss:  %.4f "
                               "[Validation] mse loss: %.4f  back mse loss: %.4f  recon loss:  %.4f  epoch time: %.2f"
                               % (epoch, tl[0], tl[1], tl[2], v_mse, v_back, v_recon, time.time() - t0))
                if rolling is None:                                             # :544-549
                    rolling, rolling_prev = 1.5 * v_recon, 2 * v_recon
                    if v_recon < 0:
                        rolling, rolling_prev = v_recon / 1.5, v_recon / 2
                rolling = persistency * rolling + (1.0 - persistency) * v_recon  # :551-552
                if rolling_prev < rolling:                                      # :554-556
                    logger.log("Stopping Training of Model since its valid_loss_rolling_average decreased")
                    break
            elif verbose:
                logger.log("Training DynamicsModel - finished epoch %i --[Training] mse loss: %.4f  back mse loss: "
                           "%.4f  recon loss: %.4f  epoch time: %.2f" % (epoch, tl[0], tl[1], tl[2], time.time() - t0))
            rolling_prev = rolling                                              # :564
        eng.repack()   # planner weight streams follow the updated master weights
        if log_tabular:
            logger.logkv("AvgModelEpochTime", float("nan"))   # reference logs mean([]) (Appendix C)
            logger.logkv("Epochs", epoch)

    def _preprocess_inputs(self, obs, act, delta, cp_obs, cp_act, future_bool, obs_next, back_delta):
        """reference :676-696: explode the future window into rows, tile the history, mask.  Kept as the host-side
        definition of a "row" (tests compare the device gather of `fit` against it); `fit` itself no longer calls it."""
        D, A, Hh, F = self.obs_space_dims, self.action_space_dims, self.history_length, self.future_length
        fb = future_bool.reshape(-1) > 0
        rows = lambda x, w: x.reshape((-1, w))[fb]
        if Hh > 0:
            _cp_obs = np.tile(cp_obs, (1, F)).reshape((-1, D * Hh))[fb]
            _cp_act = np.tile(cp_act, (1, F)).reshape((-1, A * Hh))[fb]
        else:       # vanilla model: no history window
            _cp_obs = np.zeros((int(fb.sum()), 0))
            _cp_act = np.zeros((int(fb.sum()), 0))
        return (rows(obs, D), rows(act, A), rows(delta, D), rows(obs_next, D), rows(back_delta, D), _cp_obs, _cp_act)

    # ------------------------------------------------------------------ checkpoint
    def save(self, save_path):
        """reference :571-577: joblib list of arrays in tf.trainable_variables() order + '_norm_stats'."""
        joblib.dump(self.engine.params_list(), save_path)
        if self.normalization is not None:
            joblib.dump(self.normalization, save_path + "_norm_stats")

    def load(self, load_path):
        """reference :579-588 (positional assignment)."""
        self.engine.load_params_list(joblib.load(load_path))
        if self.normalize_input:
            self.normalization = joblib.load(load_path + "_norm_stats")
            self._stats_dirty = True

    # ------------------------------------------------------------------ statistics
    def compute_normalization(self, obs, act, delta, cp_obs, cp_act, back_delta):
        """reference :590-602 (population statistics in float64 on the host)."""
        assert obs.shape[0] == delta.shape[0] == act.shape[0]
        proc_obs = self.env.obs_preproc(obs)
        n = OrderedDict()
        n["obs"] = (np.mean(proc_obs, axis=0), np.std(proc_obs, axis=0))
        n["delta"] = (np.mean(delta, axis=0), np.std(delta, axis=0))
        n["act"] = (np.mean(act, axis=0), np.std(act, axis=0))
        n["cp_obs"] = (np.mean(cp_obs, axis=0), np.std(cp_obs, axis=0))
        n["cp_act"] = (np.mean(cp_act, axis=0), np.std(cp_act, axis=0))
        n["back_delta"] = (np.mean(back_delta, axis=0), np.std(back_delta, axis=0))
        self.normalization = n
        self._stats_dirty = True

    def set_normalization(self, normalization):
        """Install precomputed statistics (same OrderedDict layout as `self.normalization`)."""
        self.normalization = normalization
        self._stats_dirty = True

    def get_normalization_stats(self):
        """reference :604-645."""
        return self._stats12()

    def _stats12(self):
        """The 12 statistic vectors in STAT_KEYS order (what the engine consumes)."""
        D, A, P, Hh = self.obs_space_dims, self.action_space_dims, self.proc_obs_space_dims, self.history_length
        if self.normalize_input:
            if self.normalization is None:
                raise RuntimeError("normalization statistics are not set: call fit(), load() or set_normalization()")
            nz = self.normalization
            om, os_ = nz["obs"]
            dm, ds = nz["delta"]
            am, as_ = (np.zeros((A,)), np.ones((A,))) if self.discrete else nz["act"]
            # the reference's VANILLA model saves only obs / delta / act (mlp_ensemble_cem_dynamics.py:343-351): a model
            # without history window / backward net never reads the other three, so they default to (0, 1)
            com, cos = (np.zeros((D * Hh,)), np.ones((D * Hh,))) if (self.state_diff or "cp_obs" not in nz) else nz["cp_obs"]
            cam, cas = (np.zeros((A * Hh,)), np.ones((A * Hh,))) if (self.discrete or "cp_act" not in nz) else nz["cp_act"]
            bm, bs = nz["back_delta"] if "back_delta" in nz else (np.zeros((D,)), np.ones((D,)))
        else:
            om, os_ = np.zeros((P,)), np.ones((P,))
            am, as_ = np.zeros((A,)), np.ones((A,))
            dm, ds = np.zeros((D,)), np.ones((D,))
            com, cos = np.zeros((D * Hh,)), np.ones((D * Hh,))
            cam, cas = np.zeros((A * Hh,)), np.ones((A * Hh,))
            bm, bs = np.zeros((D,)), np.ones((D,))
        return (om, os_, am, as_, dm, ds, com, cos, cam, cas, bm, bs)
