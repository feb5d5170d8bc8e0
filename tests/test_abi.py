"""The C-ABI library builds for gfx950, loads on a CPU-only host and exports
every symbol include/himut_hip.h declares.  No compute calls here."""
import ctypes
import os
import re

from himut_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "himut_hip.h")).read()
    return sorted(set(re.findall(r"^(?:int|void\*?|const char\*)\s+(himut_\w+)\(", text, flags=re.M)))


def test_library_exports_every_declared_symbol():
    path = build.build_hip()
    lib = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), n
    lib.himut_abi_version.restype = ctypes.c_int
    assert lib.himut_abi_version() == 2


def test_ffi_export_list_matches_header():
    from himut_amd import _ffi
    assert sorted(_ffi.EXPORTS) == declared_symbols()


def test_record_layout_is_64_bytes():
    from himut_amd import _ffi
    from oracle import oracle as O
    assert _ffi.RECORD_DTYPE.itemsize == 64
    assert _ffi.RECORD_DTYPE == O.RECORD_DTYPE


def test_null_context_is_rejected():
    lib = ctypes.CDLL(build.build_hip())
    lib.himut_run.restype = ctypes.c_int
    lib.himut_run.argtypes = [ctypes.c_void_p]
    assert lib.himut_run(None) != 0
