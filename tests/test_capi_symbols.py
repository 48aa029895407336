"""CPU-side checks of the boundary: the library loads and exports exactly what include/*.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gt4(?:hip)?_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from genometester4_amd import capi
    L = ctypes.CDLL(capi.LIB_PATH)
    declared = _declared("gt4hip.h")
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(L, name), "%s declared in include/gt4hip.h but not exported" % name
    assert sorted(capi.SYMBOLS) == declared
    # the host-side C layer: list files and the set-operations.h entry points
    for header in ("gt4_listfile.h", "gt4_set_operations.h"):
        names = _declared(header)
        assert names, header
        for name in names:
            assert hasattr(L, name), "%s declared in include/%s but not exported" % (name, header)


def test_error_strings_and_no_device_behaviour():
    from genometester4_amd import capi
    L = capi.lib()
    assert L.gt4hip_strerror(0) == b"ok"
    assert b"rule" in L.gt4hip_strerror(capi.ERULE)
    if L.gt4hip_device_count() == 0:
        # the product path must fail loudly without a GPU -- no CPU fallback
        with pytest.raises(capi.Gt4HipError) as e:
            capi.Context(0)
        assert e.value.code == capi.ENODEVICE
