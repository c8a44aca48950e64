"""mulls_amd/csrc/detmath.h: the sin / cos / atan2 the host driver and the device-resident loop share (same bits on both).
Checked here: correctly rounded against 300-bit mpmath values, and how often glibc's functions (what the reference and the
oracle call) differ from them over the range of angles an ICP step produces."""
import ctypes as C
import ctypes.util
import os
import subprocess

import numpy as np
import pytest

mpmath = pytest.importorskip("mpmath")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dm(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("dm") / "libdm.so")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", os.path.join(ROOT, "tests", "detmath_harness.cpp"), "-o", so])
    return C.CDLL(so)


@pytest.fixture(scope="module")
def dm_full(tmp_path_factory):
    """the same header with the quick phase compiled out: every call takes the full double-double evaluation"""
    so = str(tmp_path_factory.mktemp("dmf") / "libdmf.so")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-DMULLS_DETMATH_NO_QUICK", "-shared", "-fPIC",
                           os.path.join(ROOT, "tests", "detmath_harness.cpp"), "-o", so])
    return C.CDLL(so)


def call(fn, *arrs):
    n = len(arrs[0])
    out = np.empty(n)
    fn(*[np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.c_void_p) for a in arrs], out.ctypes.data_as(C.c_void_p), C.c_long(n))
    return out


def libm(name, *arrs):
    m = C.CDLL(ctypes.util.find_library("m"))
    f = getattr(m, name)
    f.restype = C.c_double
    f.argtypes = [C.c_double] * len(arrs)
    return np.array([f(*[float(a[i]) for a in arrs]) for i in range(len(arrs[0]))])


def test_sin_cos_correctly_rounded(dm):
    mpmath.mp.prec = 300
    rng = np.random.default_rng(3)
    x = np.concatenate([np.exp(rng.uniform(np.log(1e-14), np.log(0.9), 3000)) * rng.choice([-1, 1], 3000), rng.uniform(-20, 20, 600),
                        rng.uniform(-1e9, 1e9, 300), [0.0, -0.0, 5e-324, 1e-300, np.pi / 4, -np.pi / 4, np.pi / 2, np.pi, 1e12]])
    s, c = call(dm.dm_sin, x), call(dm.dm_cos, x)
    for i, xi in enumerate(x):
        m = mpmath.mpf(float(xi))
        assert s[i] == float(mpmath.sin(m)) and c[i] == float(mpmath.cos(m)), xi
    assert np.signbit(call(dm.dm_sin, np.array([-0.0]))[0])
    assert np.isnan(call(dm.dm_sin, np.array([np.inf, np.nan, 1e300]))).all() and np.isnan(call(dm.dm_cos, np.array([-np.inf, np.nan, -2e13]))).all()


def test_atan2_correctly_rounded_and_special_cases(dm):
    mpmath.mp.prec = 300
    rng = np.random.default_rng(4)
    y = np.concatenate([rng.normal(size=1500), np.exp(rng.uniform(np.log(1e-15), 0, 1500))])
    x = np.concatenate([rng.normal(size=1500), np.sqrt(np.maximum(1 - y[1500:] ** 2, 0))])  # the second half: (|v|, |w|) of a unit quaternion
    a = call(dm.dm_atan2, y, x)
    for i in range(len(y)):
        assert a[i] == float(mpmath.atan2(mpmath.mpf(float(y[i])), mpmath.mpf(float(x[i])))), (y[i], x[i])
    inf = np.inf
    for py, px in [(0.0, 1.0), (-0.0, 1.0), (0.0, -1.0), (-0.0, -1.0), (1.0, 0.0), (-1.0, 0.0), (inf, 1.0), (1.0, inf), (1.0, -inf), (inf, inf), (inf, -inf),
                   (-inf, -inf), (0.0, 0.0), (0.0, -0.0), (-0.0, -0.0), (1e-300, 1.0), (1.0, 1e-300), (5e-324, 1.0), (1e-310, 1e-310)]:
        r, e = call(dm.dm_atan2, np.array([py]), np.array([px]))[0], np.arctan2(py, px)
        assert r == e and np.signbit(r) == np.signbit(e), (py, px, r, e)
    assert np.isnan(call(dm.dm_atan2, np.array([np.nan, 1.0]), np.array([1.0, np.nan]))).all()


def test_glibc_agrees_over_the_icp_range(dm):
    """glibc 2.35's sin / cos / atan2 (error bound < 1 ulp) against the correctly rounded values: equal for every angle below
    0.02 rad of this sample (an ICP step after the first iteration), one ulp apart at a few per ten thousand of the larger ones."""
    rng = np.random.default_rng(5)
    small = np.exp(rng.uniform(np.log(1e-12), np.log(0.02), 60000)) * rng.choice([-1, 1], 60000)
    large = rng.uniform(0.02, 0.8, 60000) * rng.choice([-1, 1], 60000)
    assert (call(dm.dm_sin, small) == libm("sin", small)).all() and (call(dm.dm_cos, small) == libm("cos", small)).all()
    ds = np.abs(call(dm.dm_sin, large) - libm("sin", large)) / np.spacing(np.abs(np.sin(large)))
    dc = np.abs(call(dm.dm_cos, large) - libm("cos", large)) / np.spacing(np.abs(np.cos(large)))
    assert ds.max() <= 1.0 and dc.max() <= 1.0 and (ds > 0).mean() < 5e-3 and (dc > 0).mean() < 5e-3
    y = np.exp(rng.uniform(np.log(1e-12), np.log(0.4), 60000))  # |v| of a step's quaternion; rotation angle = 2 atan2(|v|, |w|)
    w = np.sqrt(1 - y * y)
    da = np.abs(call(dm.dm_atan2, y, w) - libm("atan2", y, w)) / np.spacing(np.arctan2(y, w))
    assert da.max() <= 1.0 and (da > 0).mean() < 5e-3
    print("glibc != correctly rounded: sin %.2e cos %.2e atan2 %.2e of the calls" % ((ds > 0).mean(), (dc > 0).mean(), (da > 0).mean()))


def test_quick_phase_returns_the_bits_of_the_full_evaluation(dm, dm_full):
    """Ziv's quick phase (detmath.h) in front of the double-double evaluation: the same double on every argument — two million angles
    over the ICP range and beyond it, quaternion pairs for atan2 — and it is the path taken (the timing ratio says so)."""
    import time

    rng = np.random.default_rng(6)
    n = 1000000
    x = np.concatenate([np.exp(rng.uniform(np.log(1e-16), np.log(0.5), n)) * rng.choice([-1, 1], n), rng.uniform(-0.6, 0.6, n // 4),
                        [0.5, -0.5, 0.4999999999999999, 2.0 ** -300, 2.0 ** -301, 2.0 ** -27, 2.0 ** -26 * 1.5, 1e-8, 3e-5]])
    t0 = time.perf_counter()
    sq, cq = call(dm.dm_sin, x), call(dm.dm_cos, x)
    t1 = time.perf_counter()
    sf, cf = call(dm_full.dm_sin, x), call(dm_full.dm_cos, x)
    t2 = time.perf_counter()
    assert (sq.view(np.uint64) == sf.view(np.uint64)).all() and (cq.view(np.uint64) == cf.view(np.uint64)).all()
    y = np.concatenate([np.exp(rng.uniform(np.log(1e-16), np.log(0.2), n)) * rng.choice([-1, 1], n), rng.uniform(-1, 1, n // 4)])
    w = np.concatenate([np.sqrt(1 - y[:n] ** 2), rng.uniform(-1, 1, n // 4)])
    aq, af = call(dm.dm_atan2, y, w), call(dm_full.dm_atan2, y, w)
    assert (aq.view(np.uint64) == af.view(np.uint64)).all()
    # scaled operands: towards the ends of the exponent range the quick phase steps aside
    for k in (2.0 ** -250, 2.0 ** 250, 3.7e-120, 2.0 ** -290, 2.0 ** 295):
        ys, ws = y[:20000] * k, w[:20000] * k
        assert (call(dm.dm_atan2, ys, ws).view(np.uint64) == call(dm_full.dm_atan2, ys, ws).view(np.uint64)).all()
    print("sin+cos of %d angles: quick %.2f s, full %.2f s" % (len(x), t1 - t0, t2 - t1))  # (reported, not asserted: a wall-clock ratio is not a property of the code)
