"""The C-ABI shared library: it loads, and it exports every symbol include/cplxamd.h declares
(no compute calls: there is no GPU in the CPU test tier)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cplxamd.h")
LIB = os.path.join(ROOT, "cplxmodule_amd", "libcplxamd.so")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cplxamd_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    return ctypes.CDLL(LIB)


def test_header_declares_entry_points():
    names = declared_symbols()
    assert len(names) >= 25
    for must in ("cplxamd_cgemm", "cplxamd_vd_kl_fwd", "cplxamd_lrt_reparam_fwd", "cplxamd_vd_mask",
                 "cplxamd_bn_fwd", "cplxamd_conv2d_fwd"):
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    missing = [n for n in declared_symbols() if not hasattr(lib, n)]
    assert not missing, missing


def test_python_binding_matches_header(lib):
    from cplxmodule_amd import _lib
    declared = set(declared_symbols())
    assert set(_lib.SIGNATURES) <= declared
    # every declared symbol is bound too (nothing exported that the host layer cannot reach)
    assert declared <= set(_lib.SIGNATURES), declared - set(_lib.SIGNATURES)
    assert _lib.load().cplxamd_abi_version() == _lib.ABI_VERSION


def test_no_gpu_arguments_are_rejected_not_crashed(lib):
    """Argument validation happens before any launch: callable without a GPU."""
    lib.cplxamd_vd_kl_fwd.restype = ctypes.c_int
    rc = lib.cplxamd_vd_kl_fwd(None, None, None, 0, None, None, None, ctypes.c_int64(0), None)
    assert rc == -1
    lib.cplxamd_vd_kl_ws_bytes.restype = ctypes.c_int64
    assert lib.cplxamd_vd_kl_ws_bytes() > 0
