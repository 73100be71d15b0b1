"""CPU-only: the C-ABI library builds, loads and exports every symbol include/wgsassign_hip.h
declares; the product path fails loudly (no CPU fallback) when no GPU is present."""
import os
import re

import pytest

from conftest import ROOT


def declared_symbols(headers=("wgsassign_hip.h", "wgsassign_hip_debug.h")):
    names = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(wgs_[a-z0-9_]+)\s*\(", text))
    return sorted(names - {"wgs_reduce_fn"})


def test_test_hooks_live_in_their_own_header():
    """The drop-in boundary (include/wgsassign_hip.h) declares no wgs_debug_* entry point; the test hooks have their own header."""
    assert not [n for n in declared_symbols(("wgsassign_hip.h",)) if n.startswith("wgs_debug_")]
    dbg = declared_symbols(("wgsassign_hip_debug.h",))
    assert len(dbg) >= 10 and all(n.startswith("wgs_debug_") for n in dbg)


def test_library_exports_every_declared_symbol():
    from wgsassign_amd import _lib, build
    build.build()
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), name
        assert name in _lib.SIGNATURES, "binding missing for " + name
    assert sorted(_lib.SIGNATURES) == names


def test_no_cpu_fallback_without_gpu():
    import ctypes
    from wgsassign_amd import _lib
    lib = _lib.load()
    n = ctypes.c_int(0)
    have_gpu = lib.wgs_device_count(ctypes.byref(n)) == 0 and n.value > 0
    if have_gpu:
        pytest.skip("a GPU is present")
    from wgsassign_amd import device
    with pytest.raises(RuntimeError, match="HIP call failed"):
        device.Context(0)
    import numpy as np
    from wgsassign_amd import emMAF_cy
    with pytest.raises(RuntimeError):
        emMAF_cy.emMAF_update(np.zeros((4, 4), dtype=np.float32), np.zeros(4, dtype=np.float32), 1)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under wgsassign_amd/ may import, link or load it."""
    pkg = os.path.join(ROOT, "wgsassign_amd")
    for base, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(base, fn)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle", src, flags=re.M), fn
                assert "libwgs_oracle" not in src and "wgs_oracle" not in src, fn
