"""The C-ABI boundary (include/sc_engine.h): every entry point the header declares is exported by the
product library (hipcc gfx950 build; loading needs no GPU, no compute call is made here) and by the
host-emulation build the CPU tier runs, and the ctypes binding lists exactly the same set."""
import ctypes
import os
import re

import pytest

from engine_runner import emu_lib
from neuraloperator_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_entry_points():
    src = open(os.path.join(ROOT, "include", "sc_engine.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(?:int|void|size_t|const\s+char\s*\*)\s+(sc_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_declares_the_bound_entry_points():
    names = declared_entry_points()
    assert len(names) >= 15
    assert sorted(_lib.ScEngineLib.SYMBOLS) == names


def test_product_library_exports_every_entry_point():
    if not os.path.isfile(_lib.DEFAULT_LIB):
        from neuraloperator_amd.csrc import build
        build.build()                                   # hipcc cross-compiles without a GPU
    lib = ctypes.CDLL(_lib.DEFAULT_LIB)
    for name in declared_entry_points():
        assert hasattr(lib, name), name
    lib.sc_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.sc_version()


def test_emulation_library_exports_every_entry_point():
    lib = emu_lib()
    for name in declared_entry_points():
        assert hasattr(lib.lib, name), name
    assert "emulation" in lib.version()


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises((_lib.EngineError, OSError)):
        _lib.ScEngineLib(str(tmp_path / "libsc_engine_missing.so"))
