import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]

    return get


# tolerances used everywhere: (rtol, atol) per dtype tag
TOL = {"f32": dict(rtol=1e-5, atol=1e-6), "f64": dict(rtol=1e-11, atol=1e-12)}


# ---- parity report (CPLXAMD_PARITY_REPORT=<file>): every np.testing.assert_allclose / assert_array_equal a test
# makes is recorded -- max ELEMENTWISE relative error (over entries with |ref| > 1e-6 max|ref|), max error relative to
# max|ref| (norm-wise), and the tolerance that was asked for -- so that the achieved accuracy can be read off next
# to the asserted one (profiles/r02_parity_report.txt).
_REPORT = os.environ.get("CPLXAMD_PARITY_REPORT")
_records = {}


def _summarise(actual, desired):
    try:
        a = np.asarray(actual, dtype=np.float64)
        d = np.asarray(desired, dtype=np.float64)
        a, d = np.broadcast_arrays(a, d)
        fin = np.isfinite(a) & np.isfinite(d)
        if not fin.any():
            return None
        err = np.abs(a - d)[fin]
        ref = np.abs(d)[fin]
        scale = ref.max() if ref.size else 0.0
        big = ref > 1e-6 * max(scale, 1e-300)
        rel_elem = float((err[big] / ref[big]).max()) if big.any() else 0.0
        return rel_elem, float(err.max() / max(scale, 1e-300)), int(fin.sum())
    except Exception:
        return None


@pytest.fixture(autouse=True)
def _parity_recorder(request):
    if not _REPORT:
        yield
        return
    orig_close, orig_equal = np.testing.assert_allclose, np.testing.assert_array_equal
    rows = _records.setdefault(request.node.nodeid, [])

    def close(actual, desired, rtol=1e-7, atol=0, *a, **k):
        s = _summarise(actual, desired)
        if s:
            rows.append(("allclose", rtol, atol) + s)
        return orig_close(actual, desired, rtol, atol, *a, **k)

    def equal(actual, desired, *a, **k):
        s = _summarise(actual, desired)
        if s:
            rows.append(("equal", 0.0, 0.0) + s)
        return orig_equal(actual, desired, *a, **k)

    np.testing.assert_allclose, np.testing.assert_array_equal = close, equal
    try:
        yield
    finally:
        np.testing.assert_allclose, np.testing.assert_array_equal = orig_close, orig_equal


def pytest_sessionfinish(session, exitstatus):
    if not _REPORT or not _records:
        return
    with open(_REPORT, "w") as fh:
        fh.write("# achieved vs asserted accuracy per test (max over the test's comparisons)\n")
        fh.write("# rel_elem = max elementwise |got-ref|/|ref| over entries with |ref| > 1e-6 max|ref|;  rel_norm = max|got-ref|/max|ref|\n")
        fh.write(f"{'test':<92}{'checks':>7}{'elements':>11}{'rel_elem':>11}{'rel_norm':>11}{'rtol_asked':>11}{'bit_exact':>10}\n")
        for node, rows in sorted(_records.items()):
            if not rows:
                continue
            n = sum(r[5] for r in rows)
            re_ = max(r[3] for r in rows)
            rn = max(r[4] for r in rows)
            rt = max(r[1] for r in rows)
            nexact = sum(1 for r in rows if r[0] == "equal")
            fh.write(f"{node[-91:]:<92}{len(rows):>7}{n:>11}{re_:>11.2e}{rn:>11.2e}{rt:>11.1e}{nexact:>10}\n")
