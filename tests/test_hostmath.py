"""Host half of an ICP iteration (mulls_amd/csrc/hostmath.h: 6x6 solve + cofactor with the Euler->quaternion Jacobian,
Euler step -> matrix, rotation angle; cregistration.hpp:1924-1964, :2740-2764, :2795-2836, :1345) against the oracle's
own implementation of the same reference lines — bit for bit, on random and on ill-conditioned systems."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hm(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("hm") / "libhostmath.so")
    subprocess.check_call(["g++", "-O3", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", os.path.join(ROOT, "tests", "hostmath_harness.cpp"), "-o", so])
    lib = C.CDLL(so)
    lib.hm_rotation_angle.restype = C.c_double
    return lib


def arr(a):
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
    return (C.c_double * len(a))(*a)


def test_solve_and_cofactor_match_oracle(hm):
    rng = np.random.default_rng(0)
    ora = pyoracle.lib()
    for k in range(200):
        A = rng.normal(0, 1, (40, 6)) * rng.uniform(0.01, 100.0, 6)
        if k % 10 == 0:
            A[:, 5] = A[:, 4] * (1 + 1e-9 * rng.normal(size=40))  # nearly dependent columns: ill-conditioned normal matrix
        N = A.T @ A
        b = A.T @ rng.normal(0, 1, 40)
        x1, x2 = (C.c_double * 6)(), (C.c_double * 6)()
        c1, c2 = (C.c_double * 36)(), (C.c_double * 36)()
        ok1 = hm.hm_solve(arr(N), arr(b), x1, c1)
        ok2 = ora.mulls_oracle_solve(arr(N), arr(b), x2, c2)
        assert bool(ok1) == (ok2 == 0)  # the oracle export returns 0 for a regular system
        assert np.array_equal(np.array(x1[:]), np.array(x2[:]), equal_nan=True), k
        assert np.array_equal(np.array(c1[:]), np.array(c2[:]), equal_nan=True), k


def test_singular_system_propagates_like_oracle(hm):
    ora = pyoracle.lib()
    N = np.zeros((6, 6))
    N[0, 0] = N[1, 1] = 1.0
    x1, x2 = (C.c_double * 6)(), (C.c_double * 6)()
    c1, c2 = (C.c_double * 36)(), (C.c_double * 36)()
    assert bool(hm.hm_solve(arr(N), arr(np.ones(6)), x1, c1)) == (ora.mulls_oracle_solve(arr(N), arr(np.ones(6)), x2, c2) == 0)
    assert np.array_equal(np.isfinite(np.array(x1[:])), np.isfinite(np.array(x2[:])))


def test_euler_step_and_rotation_angle_match_oracle(hm):
    rng = np.random.default_rng(1)
    ora = pyoracle.lib()
    ora.mulls_oracle_rotation_angle.restype = C.c_double
    for k in range(300):
        scale = 10.0 ** rng.uniform(-9, 0)
        x = np.concatenate([rng.normal(0, 1, 3), rng.normal(0, scale, 3)])
        T1, T2 = (C.c_double * 16)(), (C.c_double * 16)()
        hm.hm_construct_trans(arr(x), T1)
        ora.mulls_oracle_construct_trans(arr(x), T2)
        assert T1[:] == T2[:], k
        a1, a2 = hm.hm_rotation_angle(T1), ora.mulls_oracle_rotation_angle(T2)
        assert a1 == a2 or (np.isnan(a1) and np.isnan(a2)), (k, a1, a2)


def test_invert4_is_an_inverse(hm):
    rng = np.random.default_rng(2)
    for _ in range(50):
        R = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = R, rng.normal(0, 50, 3)
        out = (C.c_double * 16)()
        hm.hm_invert4(arr(T.T), out)  # column-major in, column-major out
        Ti = np.array(out[:]).reshape(4, 4).T
        assert np.abs(Ti @ T - np.eye(4)).max() < 1e-12
