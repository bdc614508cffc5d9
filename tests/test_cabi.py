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


# ---- ABI 19: per-call launch policy; the dispatch as a pure function (no GPU needed) ---------------------------------
def test_launch_flags_validation_needs_no_gpu(lib):
    from cplxmodule_amd import _lib
    L = _lib.load()
    both = _lib.LAUNCH_SHARED | _lib.LAUNCH_EXCLUSIVE
    # rejected before any pointer is looked at
    assert L.cplxamd_cgemm_fl(*([None] * 2), 0, 0, *([None] * 2), 0, 0, *([None] * 5), 0, 0, 0, 0, 0, 0, 0, 0, None, 0, None,
                              0, both, None) == -1
    assert L.cplxamd_gemm_plan(1, 256, 128, 128, 0, 0, _lib.BF16, 0, both, 256) == -1
    assert L.cplxamd_gemm_plan(1, 256, 128, 128, 0, 0, _lib.BF16, 0, 0x8, 256) == -1          # unknown bit
    assert L.cplxamd_conv2d_cl2_mom_chunks_fl(8, 64, 64, 64, 64, 3, 3, 1, 1, 1, 1, both) == 0


def test_gemm_dispatch_follows_the_per_call_flags(lib):
    """cplxamd_gemm_plan runs the launchers dry: 0 generic, 1 8-wave one-tile, 2 8-wave persistent, 3 one-wave-per-SIMD,
    4 / 5 split-K slabs (8-wave / w4).  The flags of ONE call decide; the deprecated process defaults only fill in what
    the flags leave open."""
    from cplxmodule_amd import _lib
    L = _lib.load()
    S, E, F, BF, F32 = _lib.LAUNCH_SHARED, _lib.LAUNCH_EXCLUSIVE, _lib.LAUNCH_FAMILY, _lib.BF16, _lib.F32
    plan = lambda *a: L.cplxamd_gemm_plan(*a, 256)  # noqa: E731       (ncu = 256: no device is asked)
    c_fwd = (1, 8192, 4096, 4096, 0, 0, BF, 0)                # the six launches of the bench step
    c_dx = (1, 8192, 4096, 4096, 0, 1, BF, 1)
    c_dw = (1, 4096, 4096, 8192, 1, 1, F32, 2)
    r_fwd = (0, 8192, 4096, 4096, 0, 0, BF, 0)
    r_dw = (0, 4096, 4096, 8192, 1, 1, F32, 2)
    for shape in (c_fwd, c_dx, c_dw, r_dw):
        assert plan(*shape, 0) == plan(*shape, S) == plan(*shape, E) == 3           # w4, one workgroup per tile
        assert plan(*shape, F(0x7f)) == 3
    # round 5: the real plain bf16 launches run the persistent form of that family when the chip is theirs
    assert [plan(*r_fwd, f) for f in (0, E, S, F(0x7f) | E)] == [6, 6, 3, 3]
    assert [plan(0, 8192, 4096, 4096, 0, 1, BF, 0, f) for f in (E, S)] == [6, 3]    # real (N,T), K = 4096
    assert plan(0, 65536, 2048, 2048, 0, 1, BF, 0, E) == 2                          # ... K = 2048: 8-wave persistent
    assert plan(0, 65536, 2048, 2048, 0, 0, BF, 0, E) == 6
    assert plan(*c_fwd, F(0xff) | E) == 3                                           # complex: not by default (slower)
    # without the w4 family: persistent where it exists, and only if the chip is this launch's
    assert [plan(*c_fwd, F(0) | f) for f in (E, S)] == [2, 1]
    assert [plan(*c_dx, F(0) | f) for f in (E, S)] == [2, 1]
    assert [plan(*r_fwd, F(0) | f) for f in (E, S)] == [2, 1]
    assert plan(*c_dw, F(0) | E) == 1                                               # (accumulate epilogue: one-tile kernel)
    # configs[3] (K = 2048): the K-depth rule of the w4 family, and its override bit 6
    cfg4_dx = (1, 65536, 2048, 2048, 0, 1, BF, 1)
    assert [plan(*cfg4_dx, f) for f in (E, S, F(0x7f) | S)] == [2, 1, 3]
    assert plan(1, 2048, 2048, 1 << 20, 1, 1, F32, 2, 0) == 5                       # split-K slabs on w4 (complex)
    assert plan(0, 2048, 2048, 1 << 20, 1, 1, F32, 2, 0) == 4                       # real slabs stay on the 8-wave kernel
    assert plan(1, 64, 128, 128, 0, 0, BF, 0, 0) == 1 and plan(1, 100, 50, 40, 0, 0, BF, 0, 0) == 0
    # the deprecated setter moves the DEFAULT only: explicit flags are unaffected
    prev = L.cplxamd_gemm_set_persistent(0)
    fam = L.cplxamd_gemm_set_family(0)
    try:
        assert plan(*c_fwd, 0) == 1 and plan(*c_fwd, E) == 2 and plan(*c_fwd, F(0x7f)) == 3
    finally:
        L.cplxamd_gemm_set_persistent(prev)
        L.cplxamd_gemm_set_family(fam)
    assert plan(*c_fwd, 0) == 3 and plan(*r_fwd, 0) == 6


def test_host_launch_policy_is_per_stream_and_windowed():
    """_lib.launch_flags(): the override registered for the current stream, else SHARED while any hook's collectives are
    in flight, else 0.  (No GPU here: one pseudo-stream; the two-stream case is tests/test_gpu_r05.py.)"""
    from cplxmodule_amd import _lib
    assert _lib.launch_flags() == 0
    with _lib.launch_policy(_lib.LAUNCH_SHARED | _lib.LAUNCH_FAMILY(0)):
        assert _lib.launch_flags() == 1 | 0x100
        with _lib.launch_policy(_lib.LAUNCH_EXCLUSIVE):
            assert _lib.launch_flags() == 2
        assert _lib.launch_flags() == 1 | 0x100
    assert _lib.launch_flags() == 0 and not _lib._policy
    owner = object()
    _lib.shared_chip_enter(owner)
    try:
        assert _lib.launch_flags() == _lib.LAUNCH_SHARED
        with _lib.launch_policy(_lib.LAUNCH_EXCLUSIVE):
            assert _lib.launch_flags() == _lib.LAUNCH_EXCLUSIVE
    finally:
        _lib.shared_chip_leave(owner)
    assert _lib.launch_flags() == 0
