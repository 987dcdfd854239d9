"""On-demand builds of rollout-kernel instantiations the library does not carry.

The production rollout kernel (csrc/rollout_xdl.h) is specialised at compile time per (env kind, hidden width, number of hidden
layers, context width, hidden nonlinearity); libcadm_hip.so carries the reference's defaults (4 layers of 128 / 200 / 256 / 512
units, swish, context_out_dim 0 / 10).  For any other `--hidden_size` / `--context_out_dim` (run_cadm_pets.py:122-135), depth
or nonlinearity (dynamics.py:17-24) this module compiles csrc/rollout_jit.hip with hipcc -- one small shared object per noise
mode, ~30 s each, cached in cadm_amd/jit_cache/ keyed by the geometry and a hash of the kernel sources -- and registers it on
the engine's ctx (`cadm_register_rollout`).  hipcc is the only requirement; there is no fallback kernel.
"""
import ctypes as C
import hashlib
import os
import subprocess

from . import _lib

HIPCC = "/opt/rocm/bin/hipcc"
CACHE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jit_cache")
_SRC = ("rollout_jit.hip", "rollout_xdl.h", "rollout_env.h", "rollout_args.h", "common.h", "xdl_geo.h")
_loaded = {}


def _source_hash():
    h = hashlib.sha256()
    for f in _SRC + (os.path.join("..", "..", "include", "cadm_hip.h"),):
        h.update(open(os.path.join(_lib.CSRC, f), "rb").read())
    return h.hexdigest()[:12]


def module_path(env_kind, C_, hid, nh, act, noise):
    return os.path.join(CACHE, "rollout_e%d_c%d_h%d_n%d_a%d_z%d_%s.so" % (env_kind, C_, hid, nh, act, noise, _source_hash()))


def build(env_kind, C_, hid, nh, act, noise, verbose=False):
    """Compile one instantiation (if it is not cached) and return the path of its shared object."""
    if hid < 113:
        raise _lib.CadmError("hidden width %d is too small for the rollout kernel's 8-wave tile split (needs >= 113 units)" % hid)
    path = module_path(env_kind, C_, hid, nh, act, noise)
    if os.path.exists(path):
        return path
    if not os.path.exists(HIPCC):
        raise _lib.CadmError("the rollout kernel for hidden=%d x %d, context_out_dim=%d, nonlinearity %d is not compiled into "
                             "libcadm_hip.so and %s is not available to build it" % (hid, nh, C_, act, HIPCC))
    os.makedirs(CACHE, exist_ok=True)
    tmp = path + ".tmp%d" % os.getpid()
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc",
           "-Wno-unused-function", "-DCADM_JIT_MODULE", "-DCADM_JIT_ENV=%d" % env_kind, "-DCADM_JIT_C=%d" % C_, "-DCADM_JIT_HID=%d" % hid,
           "-DCADM_JIT_NH=%d" % nh, "-DCADM_JIT_ACT=%d" % act, "-DCADM_JIT_NOISE=%d" % noise,
           os.path.join(_lib.CSRC, "rollout_jit.hip"), "-o", tmp]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
    if r.returncode != 0:
        raise _lib.CadmError("building the rollout kernel for hidden=%d x %d, context_out_dim=%d failed:\n%s" % (hid, nh, C_, r.stdout[-2000:]))
    os.replace(tmp, path)
    return path


def ensure(engine, noise):
    """Make sure the engine's ctx can launch rollouts in this noise mode: nothing to do for compiled-in geometries, else build /
    load the side module and register it.  Idempotent per (engine, noise mode)."""
    lib = engine.lib
    if lib.cadm_rollout_builtin(engine._ctx):
        return False
    key = (_lib.ENV_KINDS[engine.env_kind], engine.C, engine.HID, engine.NH, engine.hidden_act, noise)
    path = build(*key)
    if path not in _loaded:
        mod = C.CDLL(path)
        mod.cadm_jit_describe.argtypes = [C.POINTER(C.c_int)]
        mod.cadm_jit_describe.restype = None
        _loaded[path] = mod
    mod = _loaded[path]
    desc = (C.c_int * 8)()
    mod.cadm_jit_describe(desc)
    engine._check(lib.cadm_register_rollout(engine._ctx, noise, C.cast(mod.cadm_jit_rollout, C.c_void_p), desc), "cadm_register_rollout")
    return True
