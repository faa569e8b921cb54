"""The C-ABI library loads on a CPU-only machine and exports every symbol include/crt_gpu.h declares."""
import ctypes
import os
import re

from craytracer_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "crt_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(crt_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(api.EXPORTS)


def test_library_exports_every_symbol():
    L = ctypes.CDLL(api.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(L, name), name


def test_struct_sizes():
    assert ctypes.sizeof(api.CrtShape) == 128
    assert ctypes.sizeof(api.CrtBvh4Node) == 128
    assert api.NODE2_DTYPE.itemsize == 32
    assert api.HIT_DTYPE.itemsize == 20


def test_strerror_and_bad_args():
    L = api.lib()
    assert L.crt_strerror(0) == b"ok"
    assert b"invalid" in L.crt_strerror(-1)
    h = ctypes.c_void_p()
    assert L.crt_build_scene(None, 3, ctypes.byref(h)) == -1
    assert L.crt_scene_create(0, None, ctypes.byref(h)) == -1          # rejected before any HIP call
    n = ctypes.c_int(-5)
    assert L.crt_device_count(ctypes.byref(n)) == 0 and n.value >= 0
