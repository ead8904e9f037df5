"""CPU: the C-ABI library loads and exports every symbol include/plp.h declares
(no compute calls without a GPU); argument checking that needs no device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "plp.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(plp_[a-z_0-9]+)\s*\(", txt)))


def test_header_symbols_exported():
    from polytope_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), "libplp_hip.so does not export %s" % n
    # the python binding covers exactly the header
    assert sorted(_lib.SIGNATURES) == names
    assert lib.plp_version() >= 100


def test_no_device_is_loud_not_silent():
    from polytope_amd import _lib
    lib = _lib.load()
    if lib.plp_device_count() > 0:
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = lib.plp_ctx_create(0, C.byref(h))
    assert rc == _lib.PLP_ENODEVICE and not h
    assert b"device" in lib.plp_last_error()
    with pytest.raises(_lib.PlpError):
        _lib.context()
    import numpy as np
    import polytope_amd as pa
    with pytest.raises(_lib.PlpError):
        pa.reduce_batch(np.zeros((1, 4, 2)), np.zeros((1, 4)))


def test_oracle_is_not_reachable_from_the_product():
    """The product package must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "polytope_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "plp_oracle" not in src.replace("oracle/plp_oracle.c (test", "").replace(
                    "oracle/plp_oracle.c", ""), os.path.join(dirpath, f)
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)
