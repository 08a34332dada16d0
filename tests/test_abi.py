"""CPU-side checks of the drop-in boundary: the libraries load without a GPU and export every symbol
include/cz_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

from cubez_amd import lib

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "cz_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b([a-z_0-9]+)\s*\(", src)) - {"defined", "sizeof"})


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_library_exports_every_declared_symbol(prec):
    path = lib.lib_path(prec)
    assert os.path.exists(path), f"{path} not built: run __graft_entry__.build()"
    h = ctypes.CDLL(path)
    declared = _declared_symbols()
    assert len(declared) >= 40
    missing = [s for s in declared if not hasattr(h, s)]
    assert not missing, missing
    assert sorted(lib.ABI_SYMBOLS) == declared
    assert h.czhip_real_bytes() == (4 if prec == "f32" else 8)
    h.czhip_arch.restype = ctypes.c_char_p
    assert h.czhip_arch() == b"gfx950"


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(lib, "_HERE", "/nonexistent")
    monkeypatch.setattr(lib, "_LOADED", {})
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lib.load("f32")
