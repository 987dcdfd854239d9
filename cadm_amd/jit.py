"""On-demand builds of rollout-kernel instantiations the library does not carry.

The production rollout kernel (csrc/rollout_xdl.h) is specialised at compile time per (env kind, hidden width, number of hidden
layers, context width, hidden nonlinearity); libcadm_hip.so carries the reference's defaults (4 layers of 128 / 200 / 256 / 512
units, swish, context_out_dim 0 / 10).  For any other `--hidden_size` / `--context_out_dim` (run_cadm_pets.py:122-135), depth
or nonlinearity (dynamics.py:17-24) this module compiles csrc/rollout_jit.hip with hipcc -- one small shared object per noise
mode, cached (cache_dir(): the package's jit_cache/ when writable, else the user's cache directory) under a key made of the
geometry, a hash of the kernel sources and the compiler's version -- and registers it on the engine's ctx
(`cadm_register_rollout`).  hipcc is the only requirement ($HIPCC / $ROCM_PATH / /opt/rocm / PATH); there is no fallback
kernel.  Hidden widths below 113 need no build: they run zero-padded on the 128-wide kernel (xdl_geo.h).
"""
import ctypes as C
import fcntl
import hashlib
import json
import os
import shutil
import subprocess
import tempfile
import warnings

from . import _lib, isa_check

ARCH = "gfx950"           # the one target of this library (csrc/Makefile: ARCH); the kernels are written for its MFMA / LDS
MIN_HID, NARROW_HID = 113, 128     # xdl_geo.h: narrower nets run zero-padded on the compiled-in 128-wide kernel (no build needed)
_SRC = ("rollout_jit.hip", "rollout_xdl.h", "rollout_wt.h", "rollout_env.h", "rollout_args.h", "common.h", "xdl_geo.h")
_loaded = {}
_memo = {}


def kernel_hid(hid):
    """Hidden width of the kernel instantiation that serves a model of width `hid` (xdl_geo.h: xdl_kernel_hid)."""
    return NARROW_HID if hid < MIN_HID else hid


def hipcc():
    """The compiler: $HIPCC, else $ROCM_PATH/bin/hipcc, else /opt/rocm/bin/hipcc, else whatever `hipcc` is on PATH."""
    if "hipcc" not in _memo:
        cands = [os.environ.get("HIPCC"), os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc"), "/opt/rocm/bin/hipcc",
                 shutil.which("hipcc")]
        _memo["hipcc"] = next((c for c in cands if c and os.path.exists(c)), None)
    return _memo["hipcc"]


def cache_dir():
    """Where built modules are kept: $CADM_JIT_CACHE if set; else cadm_amd/jit_cache/ next to the package when that is writable
    (a source checkout: the modules then travel with the tree); else the user's cache directory ($XDG_CACHE_HOME or ~/.cache)
    /cadm_amd/jit -- an installed, read-only package must not be a build failure; last resort a per-user temp directory."""
    if "cache" not in _memo:
        here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jit_cache")
        cands = [os.environ.get("CADM_JIT_CACHE"), here,
                 os.path.join(os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"), "cadm_amd", "jit"),
                 os.path.join(tempfile.gettempdir(), "cadm_amd_jit_%d" % os.getuid())]
        for c in cands:
            if not c:
                continue
            try:
                os.makedirs(c, exist_ok=True)
                probe = os.path.join(c, ".w%d" % os.getpid())
                open(probe, "w").close()
                os.remove(probe)
                _memo["cache"] = c
                break
            except OSError:
                continue
        else:
            raise _lib.CadmError("no writable directory for the rollout-kernel build cache (tried %s)" % [c for c in cands if c])
    return _memo["cache"]


def _build_key():
    """Hash of everything a cached module depends on besides its geometry: the kernel sources and the compiler's version."""
    if "key" not in _memo:
        h = hashlib.sha256()
        for f in _SRC + (os.path.join("..", "..", "include", "cadm_hip.h"),):
            h.update(open(os.path.join(_lib.CSRC, f), "rb").read())
        cc = hipcc()
        if cc:
            try:
                h.update(subprocess.run([cc, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=60).stdout)
            except Exception:
                h.update(cc.encode())
        h.update(ARCH.encode())
        _memo["key"] = h.hexdigest()[:12]
    return _memo["key"]


def module_path(env_kind, C_, hid, nh, act, noise):
    return os.path.join(cache_dir(), "rollout_e%d_c%d_h%d_n%d_a%d_z%d_%s.so" % (env_kind, C_, hid, nh, act, noise, _build_key()))


def build(env_kind, C_, hid, nh, act, noise, verbose=False):
    """Compile one instantiation (if it is not cached) and return the path of its shared object.  `hid` is the KERNEL's width
    (kernel_hid(model width)).  Processes that want the same module at the same time -- the ranks of a multi-GPU job each
    construct the same model -- serialise on a lock file next to it: the first one compiles (~6-30 s), the others wait and load."""
    if hid < MIN_HID:
        raise _lib.CadmError("rollout kernel width %d: widths below %d are served by the %d-wide kernel (jit.kernel_hid)" % (hid, MIN_HID, NARROW_HID))
    path = module_path(env_kind, C_, hid, nh, act, noise)
    if os.path.exists(path):
        return path
    cc = hipcc()
    if not cc:
        raise _lib.CadmError("the rollout kernel for hidden=%d x %d, context_out_dim=%d, nonlinearity %d is not compiled into "
                             "libcadm_hip.so and no hipcc was found to build it ($HIPCC, $ROCM_PATH/bin, /opt/rocm/bin, PATH)" % (hid, nh, C_, act))
    with open(path + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)             # released when the file is closed
        if os.path.exists(path):                     # another process built it while this one waited
            return path
        tmp = path + ".tmp%d" % os.getpid()
        cmd = [cc, "--offload-arch=" + ARCH, "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc",
               "-Wno-unused-function", "-DCADM_JIT_MODULE", "-DCADM_JIT_ENV=%d" % env_kind, "-DCADM_JIT_C=%d" % C_, "-DCADM_JIT_HID=%d" % hid,
               "-DCADM_JIT_NH=%d" % nh, "-DCADM_JIT_ACT=%d" % act, "-DCADM_JIT_NOISE=%d" % noise,
               os.path.join(_lib.CSRC, "rollout_jit.hip"), "-o", tmp]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if verbose or r.returncode != 0:
            print(r.stdout)
        if r.returncode != 0:
            raise _lib.CadmError("building the rollout kernel for hidden=%d x %d, context_out_dim=%d failed:\n%s" % (hid, nh, C_, r.stdout[-2000:]))
        try:
            report = isa_scan(tmp, cc)
        except _lib.CadmError:
            os.remove(tmp)
            raise
        with open(path + ".isa.json", "w") as f:
            json.dump(report, f, indent=1)
        os.replace(tmp, path)
    return path


def isa_scan(so_path, cc=None):
    """Disassemble a freshly built module with the llvm-objdump of the compiler that built it and apply the hand-scheduling rules of
    cadm_amd/isa_check.py (resident AGPR fragments never copied, wait states around asm MFMAs / asm loads, LDS-DMA landed before the block
    barrier).  The kernels pad hazards hipcc cannot see through inline asm; whether the padding holds depends on what THIS hipcc did around
    it, so a module that breaks a rule is refused (CadmError) rather than registered.  Scratch use is a performance defect only (compiler-managed):
    reported in the returned dict and as a warning.  Without llvm-objdump the scan cannot run: a RuntimeWarning, and the report says so."""
    objdump = isa_check.find_objdump(cc or hipcc())
    if not objdump:
        warnings.warn("cadm_amd.jit: llvm-objdump not found -- the built rollout module was NOT ISA-checked (set $CADM_OBJDUMP)", RuntimeWarning)
        return {"checked": False, "reason": "llvm-objdump not found"}
    rep = isa_check.scan(so_path, objdump, which=("rollout_xdl_kernel", "rollout_wt_kernel"))
    if rep["kernels"] == 0:
        raise _lib.CadmError("ISA check of %s: no rollout kernel found in the module" % so_path)
    if rep["problems"]:
        raise _lib.CadmError("the rollout module built by %s breaks the kernels' hand-scheduling rules and was not registered (%d problem(s)):\n  %s"
                             % (cc or hipcc(), len(rep["problems"]), "\n  ".join(rep["problems"][:8])))
    if rep["scratch"]:
        warnings.warn("cadm_amd.jit: %d rollout kernel(s) of this geometry spill registers to scratch (slower, not wrong): %s"
                      % (len(rep["scratch"]), {k[-40:]: v for k, v in list(rep["scratch"].items())[:3]}), RuntimeWarning)
    return {"checked": True, "objdump": objdump, "kernels": rep["kernels"], "problems": 0, "scratch": rep["scratch"]}


def ensure(engine, noise):
    """Make sure the engine's ctx can launch rollouts in this noise mode: nothing to do for compiled-in geometries, else build /
    load the side module and register it.  Idempotent per (engine, noise mode)."""
    lib = engine.lib
    if lib.cadm_rollout_builtin(engine._ctx):
        return False
    key = (_lib.ENV_KINDS[engine.env_kind], engine.C, kernel_hid(engine.HID), engine.NH, engine.hidden_act, noise)
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
