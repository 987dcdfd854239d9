"""CPU: the C-ABI shared library builds for gfx950, loads without a GPU, and exports every symbol
that include/cadm_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from cadm_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "cadm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cadm_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from cadm_amd import _lib
    decl = _declared_symbols()
    assert len(decl) >= 20
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in decl:
        assert hasattr(raw, name), "libcadm_hip.so does not export %s" % name
        assert name in _lib.SIGNATURES, "python binding lacks %s" % name
    assert sorted(_lib.SIGNATURES) == decl


def test_abi_version_and_error_channel(lib):
    from cadm_amd import _lib
    assert lib.cadm_abi_version() == _lib.ABI_VERSION
    # argument validation happens before any HIP call: usable without a GPU
    cfg = _lib.Config()
    cfg.abi_version = 999
    ctx = ctypes.c_void_p()
    rc = lib.cadm_ctx_create(ctypes.byref(cfg), ctypes.byref(ctx))
    assert rc == -1 and b"ABI version" in lib.cadm_last_error()
    assert lib.cadm_plan_workspace_bytes(None, 1, 1) == 0


def test_config_struct_layout_matches_header():
    from cadm_amd import _lib
    # 16 int32 + 8 int32 (cp_hidden) + 2 int32 + 3 float + 1 int32 + 7 reserved = 37 words
    assert ctypes.sizeof(_lib.Config) == 37 * 4
    assert ctypes.sizeof(_lib.TrainHParams) == (6 + 9 + 9) * 4


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under cadm_amd/ may reference it."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cadm_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, "product modules import the oracle: %s" % bad


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cadm_amd import _lib
    from cadm_amd.engine import HipEngine
    with pytest.raises(_lib.CadmError, match="no CPU fallback"):
        HipEngine("halfcheetah", 5, 20, 18, 6, 18, 10, (200,) * 4, 30)
