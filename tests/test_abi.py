"""CPU: the C-ABI library loads and exports every symbol include/asr_mi355x.h declares; the
ctypes mirrors of the ABI structs have the sizes the library was compiled with."""
import ctypes
import os
import re

from speech_recognition_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "asr_mi355x.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(asr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = header_functions()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/asr_mi355x.h but not exported"


def test_binding_covers_header_and_struct_sizes_match():
    lib = _lib.load()  # raises on size mismatch / missing symbol
    assert set(_lib.SIGNATURES) == set(header_functions())
    for cname, cls in _lib.STRUCTS.items():
        assert lib.asr_struct_size(cname.encode()) == ctypes.sizeof(cls)
    assert lib.asr_struct_size(b"nope") == -1


def test_bad_arguments_fail_without_touching_the_gpu():
    lib = _lib.load()
    g = _lib.RnnGeom()
    arr = (ctypes.c_int * 1)(8)
    assert lib.asr_rnn_geometry(7, 8, 1, arr, ctypes.byref(g)) == -3
    assert b"rnn_type" in lib.asr_last_error()
    assert lib.asr_rnn_geometry(0, 10, 1, arr, ctypes.byref(g)) == 0
    assert (g.Q, g.KSt, g.wp_floats) == (3, 1, 768)
