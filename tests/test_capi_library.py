"""CPU: libmcle.so is built, loads, and exports every symbol include/mcle.h declares.
No compute call is made here (no GPU in the build container)."""
import os
import re
import subprocess

import pytest

from pyphysim_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(REPO, "include", "mcle.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mcle_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert header_functions() == _lib.exported_symbols()


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build first: make -C pyphysim_amd/csrc"
    lib = _lib.load()
    for name in header_functions():
        assert hasattr(lib, name), name
    assert lib.mcle_version() == 1
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    defined = set(re.findall(r" T (mcle_[a-z0-9_]+)", out))
    assert set(header_functions()) <= defined


def test_gfx950_code_object_present():
    out = subprocess.run(["strings", "-n", "6", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_struct_layouts_match_header():
    import ctypes
    assert ctypes.sizeof(_lib.Counters) == 64
    assert ctypes.sizeof(_lib.AwgnCfg) == 16
    assert ctypes.sizeof(_lib.FlatCfg) == 40
    assert ctypes.sizeof(_lib.MimoOfdmCfg) == 40
    assert ctypes.sizeof(_lib.IaCfg) == 56
    assert ctypes.sizeof(_lib.MimoFlatCfg) == 32
    assert ctypes.sizeof(_lib.OfdmTdlCfg) == 32 + 24 + 24 * 8 + 24 * 4
    assert ctypes.sizeof(_lib.MimoOfdmTdlCfg) == 40 + 24 + 24 * 8 + 24 * 4
    assert ctypes.sizeof(_lib.IaGeneralCfg) == 64
    assert ctypes.sizeof(_lib.BdExtIntCfg) == 64
    assert ctypes.sizeof(_lib.MuStatsCfg) == 80


def test_integration_map_names_every_entry_point():
    """INTEGRATION.md is the maintainer's map: every exported function has to appear in it."""
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    groups = {"mcle_ctx_create": "mcle_ctx_create", "mcle_malloc": "mcle_malloc", "mcle_free": "`mcle_malloc` / `free`",
              "mcle_memcpy_h2d": "memcpy_*", "mcle_memcpy_d2h": "memcpy_*", "mcle_memset": "`memset`",
              "mcle_ctx_destroy": "`destroy`", "mcle_ctx_set_stream": "`set_stream`",
              "mcle_ctx_get_stream": "`get_stream`", "mcle_ctx_sync": "`sync`",
              "mcle_ctx_device_info": "`device_info`", "mcle_timer_stop_ms": "`stop_ms`"}
    missing = [n for n in header_functions() if n not in text and groups.get(n, "\0") not in text]
    assert not missing, missing


def test_no_device_is_a_loud_error():
    lib = _lib.load()
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    from pyphysim_amd.engine import Engine
    with pytest.raises(_lib.McleError, match="no HIP device|no CPU fallback"):
        Engine(0)
