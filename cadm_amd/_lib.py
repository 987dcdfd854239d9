"""ctypes binding of libcadm_hip.so (the C ABI declared in include/cadm_hip.h).

The library is the product: there is NO Python / PyTorch fallback for any planner or
training arithmetic.  If the shared object is missing or a call fails, this module
raises -- loudly -- instead of computing anything on the host.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcadm_hip.so")              # the product; nothing in the environment redirects it
DEV_LIB_PATH = os.path.join(_HERE, "libcadm_hip_dev.so")      # product objects + csrc/dev/ (comparison kernel, developer hooks)
CSRC = os.path.join(_HERE, "csrc")

ABI_VERSION = 2
MAX_HIDDEN_LAYERS = 8
MAX_CP_LAYERS = 8

ENV_KINDS = {"halfcheetah": 0, "cripple_halfcheetah": 0, "ant": 1, "slim_humanoid": 2,
             "cartpole": 3, "pendulum": 4}
NET_FF, NET_BACK, NET_CTX = 0, 1, 2
# hidden nonlinearity codes (include/cadm_hip.h CADM_ACT_*; the reference's `_activations`, dynamics.py:17-24)
ACT_KINDS = {"swish": 0, "relu": 1, "tanh": 2, "sigmoid": 3, None: 4}
NOISE_PHILOX, NOISE_INJECT, NOISE_NONE = 0, 1, 2


class CadmError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("env_kind", C.c_int32), ("ensemble_size", C.c_int32),
        ("n_particles", C.c_int32), ("obs_dim", C.c_int32), ("act_dim", C.c_int32),
        ("proc_obs_dim", C.c_int32), ("context_dim", C.c_int32), ("n_hidden", C.c_int32),
        ("hidden", C.c_int32), ("horizon", C.c_int32), ("deterministic", C.c_int32),
        ("discrete", C.c_int32), ("reference_quirks", C.c_int32), ("history_length", C.c_int32),
        ("n_cp_hidden", C.c_int32), ("cp_hidden", C.c_int32 * MAX_CP_LAYERS),
        ("num_elites", C.c_int32), ("num_cem_iters", C.c_int32), ("alpha", C.c_float),
        ("lower_bound", C.c_float), ("upper_bound", C.c_float), ("back_model", C.c_int32),
        ("hidden_act", C.c_int32), ("reserved", C.c_int32 * 6),
    ]


class TrainHParams(C.Structure):
    _fields_ = [
        ("learning_rate", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("epsilon", C.c_float),
        ("back_coeff", C.c_float), ("weight_decay_coeff", C.c_float),
        ("weight_decays", C.c_float * (MAX_HIDDEN_LAYERS + 1)),
        ("context_weight_decays", C.c_float * (MAX_CP_LAYERS + 1)),
    ]


_P = C.c_void_p
_u32, _i = C.c_uint32, C.c_int

# name -> (restype, argtypes); must list every symbol include/cadm_hip.h declares
SIGNATURES = {
    "cadm_last_error": (C.c_char_p, []),
    "cadm_abi_version": (_i, []),
    "cadm_build_id": (C.c_char_p, []),
    "cadm_ctx_create": (_i, [C.POINTER(Config), C.POINTER(_P)]),
    "cadm_ctx_destroy": (_i, [_P]),
    "cadm_set_weights": (_i, [_P, _i, _i, _P, _P]),
    "cadm_set_logvar_bounds": (_i, [_P, _i, _P, _P]),
    "cadm_repack": (_i, [_P, _P]),
    "cadm_set_norm_stats": (_i, [_P, C.POINTER(_P), _P]),
    "cadm_context_forward": (_i, [_P, _P, _P, _i, _i, _P, _P]),
    "cadm_sample_actions": (_i, [_P, _P, _P, _P, _u32, _u32, _i, _i, _i, _P, _P]),
    "cadm_sample_actions_shard": (_i, [_P, _P, _P, _P, _u32, _u32, _i, _i, _i, _i, _i, _P, _P]),
    "cadm_sample_uniform": (_i, [_P, _u32, _u32, _i, _i, _P, _P, _P]),
    "cadm_rollout_returns": (_i, [_P, _P, _P, _P, _P, _P, _i, _u32, _u32, _i, _i, _i, _i, _i, _P, _P, _P]),
    "cadm_rollout_builtin": (_i, [_P]),
    "cadm_register_rollout": (_i, [_P, _i, _P, C.POINTER(_i)]),
    "cadm_rollout_check": (_i, [_P, _i, _i, _i]),
    "cadm_particle_mean": (_i, [_P, _P, _i, _i, _P, _P]),
    "cadm_cem_refit": (_i, [_P, _P, _i, _i, _P, _i, _P, _P, _P, _P]),
    "cadm_cem_refit_regen": (_i, [_P, _P, _i, _i, _i, _P, _P, _u32, _u32, _i, _P, _P]),
    "cadm_rs_select": (_i, [_P, _P, _i, _i, _P, _i, _P, _P, _P]),
    "cadm_plan_workspace_bytes": (C.c_size_t, [_P, _i, _i]),
    "cadm_cem_plan": (_i, [_P, _P, _P, _P, _P, _P, _i, _i, _u32, _u32, _P, _P, _P]),
    "cadm_cem_plan_staged": (_i, [_P, _P, _P, C.POINTER(C.c_int32), _i, _i, _i, _u32, _u32, _P, _P, _i, _P]),
    "cadm_rs_plan": (_i, [_P, _P, _P, _P, _i, _i, _u32, _u32, _P, _P, _P, _P]),
    "cadm_train_configure": (_i, [_P, C.POINTER(TrainHParams), _i]),
    "cadm_train_step": (_i, [_P, _P, _P, _P, _P, _P, _P, _P, _i, _i, _P, _P]),
    "cadm_train_step_rows": (_i, [_P, _P, _P, _P, _P, _P, _P, _P, _i, _P, _P, _P, C.c_longlong, _i, _i, _P, _P]),
    "cadm_train_reset": (_i, [_P, _P]),
    "cadm_predict": (_i, [_P, _P, _P, _P, _P, _i, _P, _P, _P]),
    "cadm_profile_enable": (_i, [_P, _i]),
    "cadm_profile_read": (_i, [_P, C.POINTER(C.c_float), C.POINTER(_i)]),
    "cadm_profile_read_collective": (_i, [_P, C.POINTER(C.c_float), C.POINTER(_i)]),
    "cadm_warm_start_shift": (_i, [_P, _P, _i, _P, _P, _P]),
    "cadm_history_update": (_i, [_P, _P, _P, _P, _P, _i, _i, _P, _P, _P, _P, _P]),
    "cadm_build_windows": (_i, [_P, _P, _P, _P, _i, _i, _i, _i, _i, _P, _P, _P, _i, _i, _P, _P, _P, _P, _P, _P, _P]),
    "cadm_dist_unique_id": (_i, [C.c_char_p]),
    "cadm_dist_init": (_i, [_P, C.c_char_p, _i, _i]),
    "cadm_dist_destroy": (_i, [_P]),
    "cadm_dist_info": (_i, [_P, C.POINTER(_i), C.POINTER(_i)]),
    "cadm_dist_init_external": (_i, [_P, _i, _i, C.c_void_p, _P]),
    "cadm_dist_mismatch": (_i, [_P, C.POINTER(_i), _P]),
}
# int fn(void* user, const void* send, void* recv, size_t count, void* stream)  (include/cadm_hip.h: cadm_allgather_fn)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

# developer entry points (csrc/dev/dev_api.h): only in libcadm_hip_dev.so, typed by load_dev()
DEV_SIGNATURES = {
    "cadm_dev_set_rollout": (_i, [_P, _i, _i]),
    "cadm_dev_set_timing_buffer": (_i, [_P, _P]),
    "cadm_dev_read_adam_moment": (_i, [_P, _i, _i, _i, _i, _P, C.c_long, _P]),
    "cadm_dev_rollout_plan": (_i, [_i, _i, _i, C.POINTER(_i)]),
    "cadm_dev_rollout_plan_for": (_i, [_i, _i, _i, _i, _i, C.POINTER(_i), C.POINTER(C.c_float)]),
    "cadm_dev_set_train_flavour": (_i, [_P, _i]),
    "cadm_dev_refit_sharded": (_i, [_P, _P, _i, _i, _i, _i, _P, _P, _u32, _u32, _i, _P, _P]),
    "cadm_dev_input_checksum": (_i, [_P, _P, _P, _P, _P, _P, _i, _P, _P]),
}
DEV_ROLLOUT_XDL, DEV_ROLLOUT_F32 = 0, 1

_lib = None
_dev_libs = {}


def build(verbose=False):
    """Compile libcadm_hip.so (+ the developer library) for gfx950 with hipcc (cross-compiles without a GPU).  Returns
    {"up_to_date_before": bool, "compiled": [objects hipcc produced in this call], "build_id": str} and prints one line saying
    so: an incremental `make` on a tree that already holds current objects compiles nothing, and the caller should be able
    to tell that from a build that exercised the compiler."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    before = subprocess.run(["make", "-C", CSRC, "-q", "HIPCC=" + hipcc], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode == 0
    r = subprocess.run(["make", "-C", CSRC, "-j", str(os.cpu_count() or 4), "HIPCC=" + hipcc],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
    if r.returncode != 0:
        raise CadmError("building libcadm_hip.so failed (see output above)")
    compiled = [ln.split(" -o ")[-1].strip() for ln in r.stdout.splitlines() if " -c " in ln and " -o " in ln]
    linked = [ln.split(" -o ")[-1].strip() for ln in r.stdout.splitlines() if " -shared " in ln]
    try:
        bid = open(os.path.join(CSRC, "build_id.h")).read().split('"')[1]
    except Exception:
        bid = "?"
    print("cadm_amd build: %s; hipcc compiled %d object(s), linked %d librar(ies); build id %s"
          % ("tree was up to date" if before else "tree was stale", len(compiled), len(linked), bid))
    return {"up_to_date_before": before, "compiled": compiled, "linked": linked, "build_id": bid}


def load():
    """dlopen the library and type every entry point.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CadmError(
            "libcadm_hip.so is not built (%s missing). Run `python -c \"import __graft_entry__ as g; "
            "g.build()\"` or `make -C cadm_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    # PyTorch-ROCm bundles its own libamdhip64: it has to be in the process BEFORE this library's dependency on
    # libamdhip64.so is resolved, or two HIP runtimes coexist and the second one sees no device
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    _lib = _open(LIB_PATH, SIGNATURES)
    return _lib


def _open(path, signatures):
    lib = C.CDLL(path)
    for name, (res, args) in signatures.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.cadm_abi_version() != ABI_VERSION:
        raise CadmError("%s ABI %d != binding ABI %d" % (os.path.basename(path), lib.cadm_abi_version(), ABI_VERSION))
    return lib


def load_dev(path=None):
    """The DEVELOPER library (tests and tools/ only; no product module calls this): the product's objects plus the
    fp32-MFMA comparison kernel and the hooks of csrc/dev/dev_api.h.  A separate dlopen handle -- an engine is bound
    to it explicitly with HipEngine(..., lib=load_dev()); the product library and its engines are unaffected."""
    path = os.path.abspath(path or DEV_LIB_PATH)
    if path not in _dev_libs:
        if not os.path.exists(path):
            raise CadmError("%s is not built (make -C cadm_amd/csrc dev)" % path)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        sigs = dict(SIGNATURES)
        sigs.update(DEV_SIGNATURES)
        _dev_libs[path] = _open(path, sigs)
    return _dev_libs[path]


def check(rc, what="", lib=None):
    if rc != 0:
        msg = (lib or load()).cadm_last_error()
        raise CadmError("%s failed (code %d): %s" % (what or "libcadm_hip call", rc,
                                                     msg.decode() if msg else "?"))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())
