"""BASELINE configs[0] on real data: the reference's demo pair (demo_data/pcd/000000.pcd <-> 000015.pcd, script/run_mulls_reg.sh) and the
consecutive pair 000000 <-> 000001, pinned by tests/golden/demo_pair.npz — made by tests/golden/make_demo_pair_golden.py with the REFERENCE'S OWN
LINES (extract_semantic_pts with ground normal method 3 as test/mulls_reg.cpp calls it, determine_source_target_cloud, mm_lls_icp with
test/mulls_reg.cpp:194-195's arguments).  The fixture travels to the GPU box, where /root/reference does not exist.

  CPU   the oracle reproduces the fixture bit for bit (registration) / float for float (feature clouds); where oracle/_ref exists the
        reference's lines are re-run against it
  GPU   the fixture's class clouds through mulls_icp on every search tier: the reference lines' code, the oracle's iteration and
        correspondence counts, Trans1_2 within 1e-7 m / 1e-7 rad; the raw scans through mulls_extract_features (normal method 3, the
        script's flags): every cloud float for float, then the registration from the device's own clouds."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_demo_pair_golden import FIELDS, NAMES, block_of, extract_params, reg_params  # noqa: E402

from mulls_amd import abi, synth  # noqa: E402
from oracle import pyoracle, pyref  # noqa: E402

GOLD = os.path.join(HERE, "golden", "demo_pair.npz")
CASES = ("pair_0_1", "pair_0_15", "pair_0_15_init")
pytestmark = pytest.mark.skipif(not os.path.exists(GOLD), reason="tests/golden/demo_pair.npz not generated")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def records_of(a):
    """(n, 8) floats of the fixture -> 48-byte records"""
    p = np.zeros(len(a), abi.POINT_DTYPE)
    for k, f in enumerate(FIELDS):
        p[f] = a[:, k]
    return p


def fields_of(raw):
    p = abi.points_of(raw)
    return np.stack([p[f] for f in FIELDS], 1)


def scan_of(gold, k):
    s = gold["scan_%d" % k]
    return abi.make_points(s[:, :3], None, s[:, 3], None)


def fixture_clouds(gold, k):
    """the clouds of enum mulls_extract_cloud for scan k as the fixture holds them (raw / down: not stored)"""
    return [None if n in ("raw", "down") else abi.records(records_of(gold["ex_%d_%s" % (k, n)])) for n in NAMES]


def pair_of(gold, name, clouds=None):
    a, b, a_is_target = [int(v) for v in gold[name + "_scans"]]
    ex_a = clouds[a] if clouds else fixture_clouds(gold, a)
    ex_b = clouds[b] if clouds else fixture_clouds(gold, b)
    fa, da, _ = block_of(ex_a)
    fb, db, _ = block_of(ex_b)
    tgt, src = (fa, db) if a_is_target else (fb, da)
    return abi.PairData([abi.points_of(t) for t in tgt], [abi.points_of(s) for s in src], init_guess=gold[name + "_guess"], tgt_bound=list(gold[name + "_bound"]))


def unpack(row):
    return np.array(row[:16]).reshape(4, 4).T, np.array(row[16:52]).reshape(6, 6).T, float(row[52]), float(row[53]), int(row[54])


def check_against_fixture(gold, name, r, exact):
    T, info, sigma, conf, code = unpack(gold[name + "_result"])
    assert r.code == code
    if exact:
        assert list(r.T[:]) == list(gold[name + "_result"][:16]) and list(r.info[:]) == list(gold[name + "_result"][16:52])
        assert r.sigma == np.float32(sigma) and r.confidence == np.float32(conf)
    else:
        dt, dr = synth.pose_error(r.T_matrix(), T)
        assert dt <= 1e-7 and dr <= 1e-7, (name, dt, dr)
        assert abs(r.sigma - sigma) <= 1e-6 * max(1.0, abs(sigma)) and r.confidence == np.float32(conf)
        assert np.abs(r.info_matrix() - info).max() <= 1e-6 * np.abs(info).max()
    it = gold[name + "_iters"]
    assert r.iters == int(it[0]) and list(r.ncorr) == [int(v) for v in it[1:]]


def test_fixture_is_the_demo_pair(gold):
    assert gold["scan_0"].shape == (124668, 4) and gold["scan_15"].shape == (121023, 4)  # 121 023 - 124 668 points
    for name in CASES:
        T, _, sigma, conf, code = unpack(gold[name + "_result"])
        assert code == 1 and 0 < sigma < 0.5
    # the consecutive frames and the demo pair with an odometry's guess are real registrations: ~0.7 m and ~11.7 m of motion along x
    assert abs(np.linalg.norm(unpack(gold["pair_0_1_result"])[0][:3, 3]) - 0.684) < 0.01
    assert abs(np.linalg.norm(unpack(gold["pair_0_15_init_result"])[0][:3, 3]) - 11.70) < 0.05


def test_oracle_registers_the_fixture_bit_for_bit(gold):
    for name in CASES:
        r = pyoracle.icp(pair_of(gold, name), reg_params())[0]
        check_against_fixture(gold, name, r, exact=True)


@pytest.mark.timeout(900)
def test_oracle_extracts_the_fixture_clouds(gold):
    X = extract_params()
    for k in (0, 15):
        ex = pyoracle.extract_features(scan_of(gold, k), X)
        for n, c in zip(NAMES, ex):
            if n not in ("raw", "down"):
                g = gold["ex_%d_%s" % (k, n)]
                assert g.shape == (len(c), 8) and np.array_equal(fields_of(c), g), (k, n)


@pytest.mark.skipif(not pyref.available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.timeout(900)
def test_reference_lines_reproduce_the_fixture(gold):
    X, P = extract_params(), reg_params()
    ex, _ = pyref.extract_semantic_pts(scan_of(gold, 15), X)
    for n, c in zip(NAMES, ex):
        if n not in ("raw", "down"):
            assert np.array_equal(fields_of(c), gold["ex_15_%s" % n]), n
    for name in CASES:
        r = pyref.icp(pair_of(gold, name), P)[0]
        assert list(r.T[:]) == list(gold[name + "_result"][:16]) and r.code == int(gold[name + "_result"][54])


def _register_all(gold, c):
    for name in CASES:
        r = c.icp(pair_of(gold, name), reg_params())[0]
        check_against_fixture(gold, name, r, exact=False)
    # ... and as one batch (the device-resident loop / the lock-step path): the same results as one at a time
    rb = c.icp_batch([pair_of(gold, name) for name in CASES] * 2, reg_params())
    for k, name in enumerate(CASES * 2):
        check_against_fixture(gold, name, rb[k], exact=False)


@pytest.mark.gpu
def test_device_registers_the_demo_pair_on_every_tier(gold, ctx):
    _register_all(gold, ctx)  # ctx: once per search tier (resident loop, LDS grid, global-memory grid, brute force)


@pytest.mark.gpu
def test_device_registers_the_demo_pair_auto_mode(gold, ctx_auto):
    _register_all(gold, ctx_auto)


@pytest.mark.gpu
def test_device_extracts_and_registers_the_raw_scans(gold, ctx_auto):
    """raw scan -> mulls_extract_features (ground normal method 3, run_mulls_reg.sh's flags) -> mulls_icp: the device's own clouds equal the
    reference lines' float for float, and registering them gives the reference lines' transform."""
    ctx = ctx_auto
    X = extract_params()
    clouds = {}
    for k in (0, 15):
        ex = ctx.extract_features(scan_of(gold, k), X)
        for n, c in zip(NAMES, ex):
            if n not in ("raw", "down"):
                g = gold["ex_%d_%s" % (k, n)]
                assert g.shape == (len(c), 8) and np.array_equal(fields_of(c), g), (k, n)
        clouds[k] = ex
    for name in ("pair_0_15", "pair_0_15_init"):
        r = ctx.icp(pair_of(gold, name, clouds), reg_params())[0]
        check_against_fixture(gold, name, r, exact=False)
