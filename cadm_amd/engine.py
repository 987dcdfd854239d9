"""HipEngine: device state (weights, stats, scratch) + typed calls into libcadm_hip.so.

PyTorch-ROCm is used for device memory, the current HIP stream and (for multi-GPU)
torch.distributed only; every planner / training FLOP runs in the hand-written HIP
kernels behind the C ABI.  No host fallback exists: without a GPU or without the
built library the constructor raises.
"""
import ctypes as ct
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from ._lib import Config, TrainHParams, check, ptr

STAT_KEYS = ("obs_mean", "obs_std", "act_mean", "act_std", "delta_mean", "delta_std",
             "cp_obs_mean", "cp_obs_std", "cp_act_mean", "cp_act_std",
             "back_delta_mean", "back_delta_std")


def dyn_param_names(n_hidden):
    """tf.trainable_variables() creation order inside one dynamics MLP
    (/root/reference/cadm/dynamics/core/utils.py:313-339)."""
    names = []
    for i in range(n_hidden):
        names += ["hidden_%d_weight" % i, "hidden_%d_bias" % i]
    names += ["output_mu_weight", "output_mu_bias", "output_logvar_weight", "output_logvar_bias",
              "max_logvar", "min_logvar"]
    return names


def ctx_param_names(n_cp_hidden):
    """core/utils.py:595-612."""
    names = []
    for i in range(n_cp_hidden):
        names += ["cp_hidden_%d_weight" % i, "cp_hidden_%d_bias" % i]
    return names + ["cp_output_weight", "cp_output_bias"]


class HipEngine:
    def __init__(self, env_kind, E, p, D, A, P, C, hidden_sizes, H, deterministic=False, discrete=False,
                 reference_quirks=True, history_length=10, cp_hidden_sizes=(256, 128, 64), back_model=False,
                 num_elites=50, num_cem_iters=5, alpha=0.1, lower_bound=-1.0, upper_bound=1.0, device=None, lib=None,
                 hidden_nonlinearity="swish"):
        if not torch.cuda.is_available():
            raise _lib.CadmError("no HIP device visible: the CaDM planner runs only on the GPU "
                                 "(libcadm_hip.so); there is no CPU fallback")
        # lib: a developer build (tests / tools: _lib.load_dev()); the product always binds libcadm_hip.so
        self.lib = lib if lib is not None else _lib.load()
        self._ext = None            # host-supplied all-gather of a sharded planner (dist_init_external)

        def _check(rc, what=""):
            # an exception raised inside the host-supplied collective cannot cross the library's C frames: re-raise it here
            ext = self._ext
            if rc and ext is not None and ext.error is not None:
                exc, ext.error = ext.error, None
                raise exc
            check(rc, what, self.lib)
        self._check = _check
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        hs = tuple(int(h) for h in hidden_sizes)
        if len(hs) < 1 or min(hs) < 1:
            raise ValueError("hidden_sizes must be positive, got %r" % (hs,))
        # Unequal hidden widths (the reference accepts any tuple, dynamics.py:28): the library works on ONE width, the widest
        # layer's, and the narrower layers are ZERO-PADDED to it -- a padded unit has zero weights and bias on both sides, so it
        # contributes nothing to any layer, its pre-activation is 0 and (for every nonlinearity with act(0) = 0) so is its
        # output: all its gradients are exactly 0, weight decay included, and Adam leaves an exact zero where it is.  The
        # padded tensors are what the library sees; `param_shapes_true` / `params_list` / `load_params_list` speak the model's
        # own shapes (checkpoints are the reference's).  sigmoid(0) = 0.5 would put gradient on the padded weights: refused.
        self.hidden_sizes = hs
        self._padded = len(set(hs)) != 1
        if self._padded and hidden_nonlinearity == "sigmoid":
            raise NotImplementedError("unequal hidden_sizes %r with hidden_nonlinearity='sigmoid': zero-padded units would train "
                                      "(sigmoid(0) != 0); use equal widths or swish / relu / tanh / None" % (hs,))
        self.env_kind, self.E, self.p, self.D, self.A, self.P, self.C = env_kind, E, p, D, A, P, C
        self.H, self.NH, self.HID = H, len(hs), max(hs)
        self.Hh = history_length
        self.cp_hidden_sizes = tuple(cp_hidden_sizes)
        self.deterministic, self.discrete = bool(deterministic), bool(discrete)
        self.back_model = bool(back_model)
        self.K0 = P + A + C
        if hidden_nonlinearity not in _lib.ACT_KINDS:
            raise NotImplementedError("hidden_nonlinearity %r: the kernels implement swish / relu / tanh / sigmoid / None" % (hidden_nonlinearity,))
        self.hidden_nonlinearity, self.hidden_act = hidden_nonlinearity, _lib.ACT_KINDS[hidden_nonlinearity]
        self._rollout_ready = set()
        cfg = Config()
        cfg.abi_version = _lib.ABI_VERSION
        cfg.env_kind = _lib.ENV_KINDS[env_kind]
        cfg.ensemble_size, cfg.n_particles = E, p
        cfg.obs_dim, cfg.act_dim, cfg.proc_obs_dim, cfg.context_dim = D, A, P, C
        cfg.n_hidden, cfg.hidden, cfg.horizon = self.NH, self.HID, H
        cfg.deterministic, cfg.discrete = int(self.deterministic), int(self.discrete)
        cfg.reference_quirks = int(bool(reference_quirks))
        cfg.history_length = history_length
        cfg.n_cp_hidden = len(self.cp_hidden_sizes) if C > 0 else 0
        for i, h in enumerate(self.cp_hidden_sizes if C > 0 else ()):
            cfg.cp_hidden[i] = int(h)
        cfg.num_elites, cfg.num_cem_iters, cfg.alpha = num_elites, num_cem_iters, alpha
        cfg.lower_bound, cfg.upper_bound = lower_bound, upper_bound
        cfg.back_model = int(self.back_model)
        cfg.hidden_act = self.hidden_act
        self.cfg = cfg
        self.num_elites, self.num_cem_iters = num_elites, num_cem_iters
        self._ctx = C_void_p()
        with torch.cuda.device(self.device):
            self._check(self.lib.cadm_ctx_create(ct.byref(cfg), ct.byref(self._ctx)), "cadm_ctx_create")
        self.nets = OrderedDict()   # net name -> OrderedDict(param name -> tensor)
        self._ws = None
        self._ws_key = None
        self._host_plan = None
        self._train_B = 0
        self.dist_world, self.dist_rank = 1, 0

    # ------------------------------------------------------------------ lifecycle
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self.lib.cadm_ctx_destroy(self._ctx)
            self._ctx = C_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return ct.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _t(self, x, dtype=torch.float32):
        """numpy / tensor -> contiguous device tensor."""
        if isinstance(x, torch.Tensor):
            return x.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).to(self.device)

    def stage(self, arrays):
        """Host arrays -> device tensors through ONE pinned staging buffer and ONE async H2D copy (the planner's
        per-call inputs are five tiny arrays; five pageable copies cost more than the copy itself).  None stays None."""
        live = [(i, np.asarray(a, dtype=np.float32)) for i, a in enumerate(arrays) if a is not None]
        total = sum(a.size for _, a in live)
        if getattr(self, "_stage_host", None) is None or self._stage_host.numel() < total:
            self._stage_host = torch.empty(max(total, 4096), dtype=torch.float32).pin_memory()
            self._stage_dev = torch.empty_like(self._stage_host, device=self.device)
        hv = self._stage_host.numpy()
        out, off = [None] * len(arrays), 0
        for i, a in live:
            hv[off:off + a.size] = a.reshape(-1)
            out[i] = (off, a.shape)
            off += a.size
        self._stage_dev[:off].copy_(self._stage_host[:off], non_blocking=True)
        return [None if o is None else self._stage_dev[o[0]:o[0] + int(np.prod(o[1]))].view(o[1]) for o in out]

    # ------------------------------------------------------------------ weights
    def net_names(self):
        names = []
        if self.C > 0:
            names.append("context_model")
        names.append("ff_model")
        if self.back_model:
            names.append("backward_model")
        return names   # = variable-scope creation order (dynamics.py:140,159,213)

    def param_shapes_true(self, net):
        """The model's own tensor shapes (the reference's checkpoint layout) -- differ from `param_shapes` (what the library
        holds) only for unequal hidden widths, which are zero-padded to the widest layer."""
        shapes = self.param_shapes(net)
        if net == "context_model" or not self._padded:
            return shapes
        E, hs = self.E, self.hidden_sizes
        sizes = [self.K0] + list(hs)
        for i in range(self.NH):
            shapes["hidden_%d_weight" % i] = (E, sizes[i], sizes[i + 1])
            shapes["hidden_%d_bias" % i] = (E, 1, sizes[i + 1])
        for head in ("output_mu", "output_logvar"):
            shapes[head + "_weight"] = (E, hs[-1], self.D)
        return shapes

    def _pad(self, net, name, arr):
        """Model-shaped array -> library-shaped (zero-padded) array; library-shaped input passes through."""
        want, true = self.param_shapes(net)[name], self.param_shapes_true(net)[name]
        a = np.asarray(arr) if not isinstance(arr, torch.Tensor) else arr
        if tuple(a.shape) == tuple(want) or tuple(a.shape) != tuple(true):
            return arr
        if isinstance(a, torch.Tensor):
            out = torch.zeros(want, dtype=a.dtype, device=a.device)
        else:
            out = np.zeros(want, dtype=a.dtype)
        out[tuple(slice(0, n) for n in true)] = a
        return out

    def param_shapes(self, net):
        E = self.E
        shapes = OrderedDict()
        if net == "context_model":
            sizes = [(self.D + self.A) * self.Hh] + list(self.cp_hidden_sizes)
            for i in range(len(sizes) - 1):
                shapes["cp_hidden_%d_weight" % i] = (E, sizes[i], sizes[i + 1])
                shapes["cp_hidden_%d_bias" % i] = (E, 1, sizes[i + 1])
            shapes["cp_output_weight"] = (E, sizes[-1], self.C)
            shapes["cp_output_bias"] = (E, 1, self.C)
        else:
            sizes = [self.K0] + [self.HID] * self.NH
            for i in range(self.NH):
                shapes["hidden_%d_weight" % i] = (E, sizes[i], sizes[i + 1])
                shapes["hidden_%d_bias" % i] = (E, 1, sizes[i + 1])
            for head in ("output_mu", "output_logvar"):
                shapes[head + "_weight"] = (E, self.HID, self.D)
                shapes[head + "_bias"] = (E, 1, self.D)
            shapes["max_logvar"] = (1, self.D)
            shapes["min_logvar"] = (1, self.D)
        return shapes

    def init_weights(self, rng):
        """Reference initialiser: trunc_normal(std = 1/(2 sqrt(in))), bias 0, logvar bounds 0.5 / -10
        (core/utils.py:338-339,636-641)."""
        for net in self.net_names():
            params = OrderedDict()
            for name, shape in self.param_shapes_true(net).items():      # (true fan-in for the initialiser; set_net pads)
                if name == "max_logvar":
                    v = np.ones(shape) / 2.0
                elif name == "min_logvar":
                    v = -np.ones(shape) * 10.0
                elif name.endswith("_bias"):
                    v = np.zeros(shape)
                else:
                    std = 1.0 / (2.0 * np.sqrt(shape[1]))
                    v = rng.standard_normal(shape)
                    bad = np.abs(v) > 2.0
                    while bad.any():
                        v[bad] = rng.standard_normal(int(bad.sum()))
                        bad = np.abs(v) > 2.0
                    v = v * std
                params[name] = v
            self.set_net(net, params)

    def set_net(self, net, params):
        """Install parameters (numpy or tensors) for one net and register them with the library."""
        shapes = self.param_shapes(net)
        cur = self.nets.get(net)
        out = OrderedDict()
        for name, shape in shapes.items():
            t = self._t(self._pad(net, name, params[name]))
            if tuple(t.shape) != tuple(shape):
                raise ValueError("%s/%s: shape %r != %r" % (net, name, tuple(t.shape), tuple(shape)))
            if cur is not None:     # keep registered device pointers stable
                cur[name].copy_(t)
                t = cur[name]
            out[name] = t
        self.nets[net] = out
        if cur is None:
            self._register(net)
        if cur is not None or net == "ff_model":     # weights rewritten in place: the library's packed copies follow
            if "ff_model" in self.nets:
                self._check(self.lib.cadm_repack(self._ctx, self.stream), "cadm_repack")

    def _register(self, net):
        prm = self.nets[net]
        nid = {"ff_model": _lib.NET_FF, "backward_model": _lib.NET_BACK, "context_model": _lib.NET_CTX}[net]
        if net == "context_model":
            layers = ["cp_hidden_%d" % i for i in range(len(self.cp_hidden_sizes))] + ["cp_output"]
        else:
            layers = ["hidden_%d" % i for i in range(self.NH)] + ["output_mu", "output_logvar"]
        for li, base in enumerate(layers):
            self._check(self.lib.cadm_set_weights(self._ctx, nid, li, ptr(prm[base + "_weight"]), ptr(prm[base + "_bias"])),
                  "cadm_set_weights")
        if net != "context_model":
            self._check(self.lib.cadm_set_logvar_bounds(self._ctx, nid, ptr(prm["max_logvar"]), ptr(prm["min_logvar"])),
                  "cadm_set_logvar_bounds")

    def repack(self):
        self._check(self.lib.cadm_repack(self._ctx, self.stream), "cadm_repack")

    def params_list(self):
        """Flat list of numpy arrays in the reference's tf.trainable_variables() order
        (context_model, ff_model, backward_model; dynamics.py:266)."""
        out = []
        for net in self.net_names():
            true = self.param_shapes_true(net)
            for name, t in self.nets[net].items():
                a = t.detach().cpu().numpy()
                out.append(a[tuple(slice(0, n) for n in true[name])].copy())      # (padding of unequal hidden widths stripped)
        return out

    def load_params_list(self, arrays):
        it = iter(arrays)
        for net in self.net_names():
            params = OrderedDict()
            for name in self.param_shapes(net):
                params[name] = np.asarray(next(it))
            self.set_net(net, params)

    # ------------------------------------------------------------------ stats
    def set_stats(self, stats):
        arrs = [np.ascontiguousarray(np.asarray(stats[k], dtype=np.float32)).reshape(-1) for k in STAT_KEYS]
        lens = [self.P, self.P, self.A, self.A, self.D, self.D, self.D * self.Hh, self.D * self.Hh,
                self.A * self.Hh, self.A * self.Hh, self.D, self.D]
        for k, a, n in zip(STAT_KEYS, arrs, lens):
            if a.size != n:
                raise ValueError("stat %s has %d entries, expected %d" % (k, a.size, n))
        ptrs = (ct.c_void_p * 12)(*[a.ctypes.data_as(ct.c_void_p) for a in arrs])
        self._check(self.lib.cadm_set_norm_stats(self._ctx, ptrs, self.stream), "cadm_set_norm_stats")

    # ------------------------------------------------------------------ rollout kernel for this geometry
    def ensure_rollout(self, noise=None, m=1, n_local=1):
        """The rollout kernel is compile-time specialised per geometry: make sure one exists for this engine (compiled in, or
        built on demand by cadm_amd.jit -- ~30 s the first time, cached) and that a launch fits the hardware (LDS for this
        horizon), so that a problem shows at construction, not at the first `get_action`."""
        from . import jit
        if noise is None:
            noise = _lib.NOISE_NONE if self.deterministic else _lib.NOISE_PHILOX
        if noise not in self._rollout_ready:
            jit.ensure(self, noise)
            self._check(self.lib.cadm_rollout_check(self._ctx, noise, int(m), int(n_local)), "cadm_rollout_check")
            self._rollout_ready.add(noise)

    # ------------------------------------------------------------------ planner primitives
    def context_forward(self, cp_obs, cp_act, bs=False):
        cp_obs, cp_act = self._t(cp_obs), self._t(cp_act)
        m = cp_obs.shape[1] if bs else cp_obs.shape[0]
        out = torch.empty((self.E, m, self.C), dtype=torch.float32, device=self.device)
        self._check(self.lib.cadm_context_forward(self._ctx, ptr(cp_obs), ptr(cp_act), m, int(bs), ptr(out), self.stream),
              "cadm_context_forward")
        return out

    def sample_actions(self, mean, var, n_global, z=None, seed=0, call=0, it=0, cand_offset=0, n_local=None):
        """Candidates [cand_offset, cand_offset + n_local) of the [m, n_global, H, A] buffer (default: all of them).  A rank of a sharded
        planner draws only its own shard -- the draws are keyed by the global element index -- and leaves the rest of the buffer alone."""
        mean, var = self._t(mean), self._t(var)
        m = mean.shape[0]
        z = None if z is None else self._t(z)
        n_local = n_global - cand_offset if n_local is None else n_local
        out = torch.empty((m, n_global, self.H, self.A), dtype=torch.float32, device=self.device)
        self._check(self.lib.cadm_sample_actions_shard(self._ctx, ptr(mean), ptr(var), ptr(z), seed, call, it, m, n_global, cand_offset,
                                                       n_local, ptr(out), self.stream), "cadm_sample_actions")
        return out

    def sample_uniform(self, m, n_global, seed=0, call=0):
        out = torch.empty((m, n_global, self.H, self.A), dtype=torch.float32, device=self.device)
        raw = torch.empty((m, n_global, self.H), dtype=torch.int32, device=self.device) if self.discrete else None
        self._check(self.lib.cadm_sample_uniform(self._ctx, seed, call, m, n_global, ptr(out), ptr(raw), self.stream),
              "cadm_sample_uniform")
        return out, raw

    def rollout_returns(self, obs, ctx_vec, actions, eps=None, obs_rows=None, norm_actions=True, seed=0, call=0,
                        it=0, cand_offset=0, n_local=None, want_traj=False):
        obs, actions = self._t(obs), self._t(actions)
        m, n_global = actions.shape[0], actions.shape[1]
        n_local = n_global - cand_offset if n_local is None else n_local
        ctx_vec = None if ctx_vec is None else self._t(ctx_vec)
        eps = None if eps is None else self._t(eps)
        obs_rows = None if obs_rows is None else self._t(obs_rows)
        self.ensure_rollout(_lib.NOISE_NONE if self.deterministic else _lib.NOISE_INJECT if eps is not None else _lib.NOISE_PHILOX, m, n_local)
        rows = torch.empty((m, n_local, self.p), dtype=torch.float32, device=self.device)
        traj = (torch.empty((self.H, m, n_local, self.p, self.D), dtype=torch.float32, device=self.device)
                if want_traj else None)
        self._check(self.lib.cadm_rollout_returns(self._ctx, ptr(obs), ptr(obs_rows), ptr(ctx_vec), ptr(actions), ptr(eps),
                                            int(norm_actions), seed, call, it, cand_offset, n_global, m, n_local,
                                            ptr(rows), ptr(traj), self.stream), "cadm_rollout_returns")
        return (rows, traj) if want_traj else rows

    def particle_mean(self, rows):
        m, n_local = rows.shape[0], rows.shape[1]
        out = torch.empty((m, n_local), dtype=torch.float32, device=self.device)
        self._check(self.lib.cadm_particle_mean(self._ctx, ptr(rows), m, n_local, ptr(out), self.stream), "cadm_particle_mean")
        return out

    def cem_refit(self, cand, actions, mean, var, G=1, want_elites=False, regen=None):
        """cand [G,m,n_local] (or [m,n] when G == 1); mean/var updated IN PLACE.  regen = (seed, call, it): the elites' action
        sequences are drawn again from the device RNG by global candidate id instead of being read from `actions` (a rank of a
        sharded planner holds only its own shard's draws); bit-identical to the gathered form."""
        m = actions.shape[0]
        n_local = actions.shape[1] // G
        el = torch.empty((m, self.num_elites), dtype=torch.int32, device=self.device) if want_elites else None
        if regen is not None:
            seed, call, it = regen
            self._check(self.lib.cadm_cem_refit_regen(self._ctx, ptr(cand), G, n_local, m, ptr(mean), ptr(var), seed, call, it, ptr(el),
                                                      self.stream), "cadm_cem_refit_regen")
            return el
        self._check(self.lib.cadm_cem_refit(self._ctx, ptr(cand), G, n_local, ptr(actions), m, ptr(mean), ptr(var), ptr(el),
                                      self.stream), "cadm_cem_refit")
        return el

    def rs_select(self, cand, actions, G=1):
        m = actions.shape[0]
        n_local = actions.shape[1] // G
        out = torch.empty((m, self.A), dtype=torch.float32, device=self.device)
        best = torch.empty((m,), dtype=torch.int32, device=self.device)
        self._check(self.lib.cadm_rs_select(self._ctx, ptr(cand), G, n_local, ptr(actions), m, ptr(out), ptr(best),
                                      self.stream), "cadm_rs_select")
        return out, best

    # ------------------------------------------------------------------ fused planners
    def _workspace(self, m, n):
        key = (m, n)
        if self._ws_key != key:
            nbytes = self.lib.cadm_plan_workspace_bytes(self._ctx, m, n)
            self._ws = torch.empty((nbytes,), dtype=torch.uint8, device=self.device)
            self._ws_key = key
            self._host_plan = None          # (holds a pointer into the workspace)
        return self._ws

    def host_out(self, shape):
        """Persistent PINNED host buffer the planners can write their (tiny) result into directly: the last kernel of a plan
        stores over PCIe, the caller only synchronises the stream -- no D2H copy launch, no allocation per call."""
        n = int(np.prod(shape))
        if getattr(self, "_host_out", None) is None or self._host_out.numel() < n:
            self._host_out = torch.empty(max(n, 1024), dtype=torch.float32).pin_memory()
        return self._host_out[:n].view(tuple(shape))

    def cem_plan(self, obs, cp_obs, cp_act, init_mean, init_var, n, seed=0, call=0, out=None):
        """`out`: optional [m,H,A] float32 destination (device tensor, or a pinned host tensor such as `host_out`)."""
        obs, init_mean, init_var = self._t(obs), self._t(init_mean), self._t(init_var)
        cp_obs = None if cp_obs is None else self._t(cp_obs)
        cp_act = None if cp_act is None else self._t(cp_act)
        m = obs.shape[0]
        self.ensure_rollout(None, m, max(1, n // self.dist_world))
        ws = self._workspace(m, n)
        if out is None:
            out = torch.empty((m, self.H, self.A), dtype=torch.float32, device=self.device)
        self._check(self.lib.cadm_cem_plan(self._ctx, ptr(obs), ptr(cp_obs), ptr(cp_act), ptr(init_mean), ptr(init_var), m, n,
                                     seed, call, ptr(ws), ptr(out), self.stream), "cadm_cem_plan")
        return out

    def cem_plan_host(self, arrays, n, seed=0, call=0, shapes=None):
        """The class API's fast path: `arrays` = (obs, cp_obs, cp_act, init_mean, init_var) as host arrays (None = absent) ->
        the plan [m,H,A] as a fresh numpy array.  Everything between is ONE library call (`cadm_cem_plan_staged`): the inputs
        go through a persistent pinned block, the plan comes back through another; the block layout is cached per shape set."""
        if shapes is None:
            shapes = tuple(None if a is None else tuple(np.shape(a)) for a in arrays)
        st = self._host_plan
        if st is None or st["shapes"] != shapes or st["n"] != n:
            offs, views, pos = [], [], 0
            sizes = [0 if sh is None else int(np.prod(sh)) for sh in shapes]
            total = max(sum(sizes), 1)
            host = torch.empty(total, dtype=torch.float32).pin_memory()
            hv = host.numpy()
            for sh, sz in zip(shapes, sizes):
                offs.append(-1 if sh is None else pos)
                views.append(None if sh is None else hv[pos:pos + sz].reshape(sh))
                pos += sz
            m = shapes[0][0]
            out = torch.zeros(m * self.H * self.A + 2 * m, dtype=torch.float32).pin_memory()   # plan + m completion flags + m mismatch words
            flags = out.numpy()[m * self.H * self.A + m:].view(np.uint32)
            st = self._host_plan = dict(shapes=shapes, n=n, ws=ptr(self._workspace(m, n)), host=host, views=views, dev=torch.empty(total, dtype=torch.float32, device=self.device),
                                        off=(ct.c_int32 * 5)(*offs), total=total, out=out, out_np=out.numpy()[:m * self.H * self.A].reshape(m, self.H, self.A), m=m, mismatch=flags,
                                        hp=ct.c_void_p(host.data_ptr()), op=ct.c_void_p(out.data_ptr()))
            st["dp"] = ct.c_void_p(st["dev"].data_ptr())
            self.ensure_rollout(None, m, max(1, n // self.dist_world))
        for v, a in zip(st["views"], arrays):
            if v is not None:
                np.copyto(v, a, casting="unsafe")
        rc = self.lib.cadm_cem_plan_staged(self._ctx, st["hp"], st["dp"], st["off"], st["total"], st["m"], n, seed, call,
                                           st["ws"], st["op"], 1, self.stream)
        if rc:
            self._check(rc, "cadm_cem_plan_staged")
        out = st["out_np"].copy()
        if self.dist_world > 1 and st["mismatch"].any():
            # the sharded refit compares the checksums every rank's all-gather payload carries (csrc/cem.hip: RefitRegen) and raises a
            # flag of its own: a NaN plan alone is also what a single-rank call returns for a non-finite observation
            self._check(self.lib.cadm_dist_mismatch(self._ctx, ct.byref(ct.c_int(0)), self.stream), "cadm_dist_mismatch")      # (resets the device word)
            raise RuntimeError(self.MISMATCH_MSG)
        return out

    MISMATCH_MSG = ("candidate-sharded planning: the ranks of the group were fed different obs / history / warm start on this call "
                    "(input checksums differ; the plan is NaN on every rank, and every rank raises on the same call)")

    def dist_mismatch(self):
        """Did the last sharded cem_plan see different replicated inputs on some rank?  Synchronises the stream; reading resets the flag."""
        v = ct.c_int(0)
        self._check(self.lib.cadm_dist_mismatch(self._ctx, ct.byref(v), self.stream), "cadm_dist_mismatch")
        return bool(v.value)

    def rs_plan(self, obs, cp_obs, cp_act, n, seed=0, call=0):
        obs = self._t(obs)
        cp_obs = None if cp_obs is None else self._t(cp_obs)
        cp_act = None if cp_act is None else self._t(cp_act)
        m = obs.shape[0]
        self.ensure_rollout(None, m, max(1, n // self.dist_world))
        ws = self._workspace(m, n)
        out = torch.empty((m, self.A), dtype=torch.float32, device=self.device)
        raw = torch.empty((m,), dtype=torch.int32, device=self.device) if self.discrete else None
        self._check(self.lib.cadm_rs_plan(self._ctx, ptr(obs), ptr(cp_obs), ptr(cp_act), m, n, seed, call, ptr(ws), ptr(out),
                                    ptr(raw), self.stream), "cadm_rs_plan")
        return raw if self.discrete else out

    # ------------------------------------------------------------------ training
    def train_configure(self, learning_rate, weight_decays, context_weight_decays, weight_decay_coeff, back_coeff,
                        max_batch, beta1=0.9, beta2=0.999, epsilon=1e-8):
        hp = TrainHParams()
        hp.learning_rate, hp.beta1, hp.beta2, hp.epsilon = learning_rate, beta1, beta2, epsilon
        hp.back_coeff, hp.weight_decay_coeff = back_coeff, weight_decay_coeff
        wd = list(weight_decays)
        for i in range(self.NH):
            hp.weight_decays[i] = wd[i]
        hp.weight_decays[self.NH] = wd[-1]            # both heads (core/utils.py:328,334)
        if self.C > 0:
            cwd = list(context_weight_decays)
            ncp = len(self.cp_hidden_sizes)
            for i in range(ncp):
                hp.context_weight_decays[i] = cwd[i]
            hp.context_weight_decays[ncp] = cwd[-1]   # cp_output (core/utils.py:610)
        self._check(self.lib.cadm_train_configure(self._ctx, ct.byref(hp), int(max_batch)), "cadm_train_configure")
        self._train_B = int(max_batch)

    def train_step(self, batch, train=True):
        """batch: dict of [E,B,.] tensors (obs, act, delta, obs_next, back_delta, cp_obs, cp_act; the
        last four may be None for the vanilla model).  Returns a device tensor [mse, back_mse, recon]."""
        B = batch["obs"].shape[1]
        g = lambda k: ptr(batch.get(k))
        losses = torch.empty((3,), dtype=torch.float32, device=self.device)
        self._check(self.lib.cadm_train_step(self._ctx, g("obs"), g("act"), g("delta"), g("obs_next"), g("back_delta"),
                                       g("cp_obs"), g("cp_act"), B, int(train), ptr(losses), self.stream),
              "cadm_train_step")
        return losses

    def train_step_rows(self, dev, F, row_w, row_f, idx, train=True):
        """One step on rows of the windowed device dataset `dev` (dict of [N, F*dim] / [N, dim] tensors): idx [E,B] int64
        row ids, row_w / row_f int64 (window, future offset) of every training row.  No gather is materialised."""
        B = idx.shape[1]
        if idx.stride(1) != 1:
            idx = idx.contiguous()
        g = lambda k: ptr(dev.get(k))
        losses = torch.empty((3,), dtype=torch.float32, device=self.device)
        self._check(self.lib.cadm_train_step_rows(self._ctx, g("obs"), g("act"), g("delta"), g("obs_next"), g("back_delta"),
                                            g("cp_obs"), g("cp_act"), int(F), ptr(row_w), ptr(row_f), ptr(idx), int(idx.stride(0)), B,
                                            int(train),
                                            ptr(losses), self.stream), "cadm_train_step_rows")
        return losses

    # ------------------------------------------------------------------ multi-GPU (RCCL communicator owned by the ctx)
    def dist_init(self, group=None):
        """Create the ctx's RCCL communicator over the ranks of a torch.distributed group (the group is only
        used to broadcast the 128-byte ncclUniqueId).  Afterwards cem_plan / rs_plan take the GLOBAL candidate
        count and run the sharded planner entirely on the stream, one ncclAllGather per CEM iteration."""
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        buf = ct.create_string_buffer(128)
        # rank 0 ALWAYS reaches the broadcast: a failure to draw the id (librccl missing, ...) travels as a status byte, so
        # the other ranks never wait in a broadcast that rank 0 skipped (mismatched collectives would hang)
        status, err = 1, None
        if rank == 0:
            try:
                self._check(self.lib.cadm_dist_unique_id(buf), "cadm_dist_unique_id")
            except _lib.CadmError as exc:
                status, err = 0, exc
        t = torch.tensor(list(buf.raw) + [status], dtype=torch.uint8, device=self.device if dist.get_backend(group) == "nccl" else "cpu")
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        raw = t.cpu().tolist()
        if raw[128] != 1:
            raise err if err is not None else _lib.CadmError("cadm_dist_unique_id failed on rank 0 of the group")
        ident = bytes(raw[:128])
        with torch.cuda.device(self.device):
            self._check(self.lib.cadm_dist_init(self._ctx, ident, world, rank), "cadm_dist_init")
        nr, rk = self.dist_info()
        if nr != world or rk != rank:
            raise _lib.CadmError("RCCL communicator reports nranks=%d rank=%d, expected %d / %d" % (nr, rk, world, rank))
        self.dist_world, self.dist_rank = world, rank

    def dist_init_external(self, group=None):
        """The same sharded planner with the all-gather supplied by torch.distributed over `group`, whatever its backend (gloo, or nccl
        where the in-library communicator cannot be built): `cadm_cem_plan` / `cadm_rs_plan` run the loop of the RCCL path and call back
        into `planner.ExternalAllGather` once per CEM iteration.  Both pointers of a call lie inside this engine's planner workspace."""
        import torch.distributed as dist
        from . import planner as _planner
        world, rank = dist.get_world_size(group), dist.get_rank(group)

        def resolve(ptr_, nbytes):
            ws = self._ws
            off = int(ptr_) - ws.data_ptr()
            if ws is None or off < 0 or off + nbytes > ws.numel() or off % 4:
                raise _lib.CadmError("external all-gather: pointer outside the planner workspace")
            return ws[off:off + nbytes].view(torch.float32)
        ext = _planner.ExternalAllGather(group, resolve)
        self._check(self.lib.cadm_dist_init_external(self._ctx, world, rank, ct.cast(ext.cfunc, ct.c_void_p), None), "cadm_dist_init_external")
        self._ext = ext
        self.dist_world, self.dist_rank = world, rank

    def dist_destroy(self):
        self._check(self.lib.cadm_dist_destroy(self._ctx), "cadm_dist_destroy")
        self.dist_world, self.dist_rank = 1, 0
        self._ext = None

    def dist_info(self):
        """(nranks, rank) as RCCL itself reports them for the ctx's communicator (ncclCommCount / ncclCommUserRank)."""
        n, r = ct.c_int(0), ct.c_int(0)
        self._check(self.lib.cadm_dist_info(self._ctx, ct.byref(n), ct.byref(r)), "cadm_dist_info")
        return int(n.value), int(r.value)

    # ------------------------------------------------------------------ in-library kernel timing
    def profile_enable(self, on=True):
        self._check(self.lib.cadm_profile_enable(self._ctx, int(on)), "cadm_profile_enable")

    def profile_read(self):
        """-> (total milliseconds, launches) of the rollout kernel since the last read."""
        ms, cnt = ct.c_float(0.0), ct.c_int(0)
        self._check(self.lib.cadm_profile_read(self._ctx, ct.byref(ms), ct.byref(cnt)), "cadm_profile_read")
        return float(ms.value), int(cnt.value)

    # ------------------------------------------------------------------ developer hooks (libcadm_hip_dev.so only)
    def dev_set_rollout(self, kind="xdl", row_tiles=0):
        """Select the fp32-MFMA comparison kernel ("f32") / force a flavour of the production kernel on THIS engine (row_tiles: 0 = the
        launcher's plan, 1 / 2 = cooperative kernel with one / two row tiles per workgroup, 3 / 4 = wave-tile kernel with 8 / 4 tiles).
        Exists only when the engine was built on the developer library (HipEngine(..., lib=_lib.load_dev()))."""
        if not hasattr(self.lib, "cadm_dev_set_rollout"):
            raise _lib.CadmError("dev_set_rollout needs the developer library (HipEngine(..., lib=_lib.load_dev()))")
        self._check(self.lib.cadm_dev_set_rollout(self._ctx, {"xdl": _lib.DEV_ROLLOUT_XDL, "f32": _lib.DEV_ROLLOUT_F32}[kind],
                                                  int(row_tiles)), "cadm_dev_set_rollout")

    def dev_read_adam_moment(self, net, name, second=False):
        """Adam's first (second=False) / second moment of one trained tensor, shaped like the tensor.  With beta1 = 0 the first
        moment after ONE training step is that step's gradient exactly (m = 0 m + 1 g): how the tests read dL/dW, dL/db element
        by element.  Developer library only."""
        if not hasattr(self.lib, "cadm_dev_read_adam_moment"):
            raise _lib.CadmError("dev_read_adam_moment needs the developer library (HipEngine(..., lib=_lib.load_dev()))")
        nid = {"ff_model": _lib.NET_FF, "backward_model": _lib.NET_BACK, "context_model": _lib.NET_CTX}[net]
        if name in ("max_logvar", "min_logvar"):
            layer, is_bias = (-1 if name == "max_logvar" else -2), 0
        else:
            base, kind = name.rsplit("_", 1)
            if net == "context_model":
                layers = ["cp_hidden_%d" % i for i in range(len(self.cp_hidden_sizes))] + ["cp_output"]
            else:
                layers = ["hidden_%d" % i for i in range(self.NH)] + ["output_mu", "output_logvar"]
            layer, is_bias = layers.index(base), int(kind == "bias")
        out = torch.empty_like(self.nets[net][name])
        self._check(self.lib.cadm_dev_read_adam_moment(self._ctx, nid, layer, is_bias, int(second), ptr(out), out.numel(), self.stream),
                    "cadm_dev_read_adam_moment")
        return out

    def profile_read_collective(self):
        """-> (total milliseconds, calls) of the in-library ncclAllGather since the last read (sharded planner)."""
        ms, cnt = ct.c_float(0.0), ct.c_int(0)
        self._check(self.lib.cadm_profile_read_collective(self._ctx, ct.byref(ms), ct.byref(cnt)), "cadm_profile_read_collective")
        return float(ms.value), int(cnt.value)

    def predict_heads(self, obs, act, cp_obs=None, cp_act=None):
        """[E,B,.] inputs -> (mu [E,B,D] normalised, logvar [E,B,D] clamped or None)."""
        obs, act = self._t(obs), self._t(act)
        cp_obs = None if cp_obs is None else self._t(cp_obs)
        cp_act = None if cp_act is None else self._t(cp_act)
        B = obs.shape[1]
        mu = torch.empty((self.E, B, self.D), dtype=torch.float32, device=self.device)
        lv = None if self.deterministic else torch.empty_like(mu)
        self._check(self.lib.cadm_predict(self._ctx, ptr(obs), ptr(act), ptr(cp_obs), ptr(cp_act), B, ptr(mu), ptr(lv), self.stream),
              "cadm_predict")
        return mu, lv

    def train_reset(self):
        self._check(self.lib.cadm_train_reset(self._ctx, self.stream), "cadm_train_reset")


def C_void_p():
    return ct.c_void_p()
