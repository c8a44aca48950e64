"""Pins the oracle restatement against the reference's own source lines (oracle/_ref/libmulls_ref.so, built by
oracle/build_ref.sh from /root/reference against the stand-in headers in oracle/ref_shim).  The two share no code: the
oracle is an independent restatement, _ref is the upstream function bodies.  They must agree BIT FOR BIT on every
output the reference interface exposes (Trans1_2, information_matrix, sigma, confidence, return code).

Skipped where the library has not been built (it needs /root/reference at build time; the prebuilt .so travels with
the repository snapshot to the GPU box)."""
import numpy as np
import pytest

from conftest import planes_scene, transformed_copy
from mulls_amd import abi, synth
from oracle import pyoracle, pyref

pytestmark = pytest.mark.skipif(not pyref.available(), reason="oracle/_ref/libmulls_ref.so not built")


def same(ro, rr):
    assert ro.code == rr.code
    assert np.array_equal(ro.T_matrix(), rr.T_matrix(), equal_nan=True)
    assert np.array_equal(ro.info_matrix(), rr.info_matrix(), equal_nan=True)
    assert ro.sigma == rr.sigma or (np.isnan(ro.sigma) and np.isnan(rr.sigma))
    assert ro.confidence == rr.confidence or (np.isnan(ro.confidence) and np.isnan(rr.confidence))


PARAM_SETS = {
    "kitti_s2s": lambda: abi.kitti_params(dis_thre_unit=2.4),
    "kitti_fixed20": lambda: abi.kitti_params(converge_translation=0.0, converge_rotation_d=0.0),
    "defaults": lambda: abi.default_params(),
    "all6_nofilter_w1111": lambda: abi.default_params(used_feature_type="111111", apply_intersection_filter=0, weight_strategy="1111"),
    "equal_weights": lambda: abi.default_params(weight_strategy="0000", used_feature_type="111100"),
    "mulls_reg_cli": lambda: abi.default_params(max_iter_num=10, dis_thre_unit=3.0, converge_translation=0.001, dis_thre_min=0.75),
    "tight_bearing": lambda: abi.default_params(normal_bearing=10.0, weight_strategy="1110"),
}


@pytest.mark.parametrize("name", sorted(PARAM_SETS))
def test_oracle_equals_reference_bodies(pairs_small, name):
    P = PARAM_SETS[name]()
    for pair, _ in pairs_small:
        same(pyoracle.icp(pair, P)[0], pyref.icp(pair, P)[0])


def test_quirks_are_the_reference_s(pairs_small):
    """faithful=1 is what upstream computes: the pt2li off-diagonal drop and the d^2-weighted vertex residual."""
    pair, _ = pairs_small[0]
    P = abi.default_params(used_feature_type="111111", weight_strategy="1101")
    rr = pyref.icp(pair, P)[0]
    same(pyoracle.icp(pair, P)[0], rr)
    P.faithful = 0
    ru = pyoracle.icp(pair, P)[0]
    assert not np.array_equal(ru.T_matrix(), rr.T_matrix()) or ru.sigma != rr.sigma


def test_failure_codes_and_edge_cases():
    rng = np.random.default_rng(7)
    tgt = planes_scene(rng)
    good = abi.PairData(tgt, transformed_copy(tgt, np.linalg.inv(synth.se3(0.1, 0.05, 0.0, 0, 0, 0.01))))
    far = abi.PairData(tgt, transformed_copy(tgt, synth.se3(200.0, 0, 0)))
    for pr, kw in ((good, {}), (far, dict(apply_intersection_filter=0)), (good, dict(max_bearable_rotation_d=0.1)), (good, dict(sigma_thre=1e-9)),
                   (good, dict(max_iter_num=0)), (good, dict(max_iter_num=1)), (far, {})):
        P = abi.default_params(used_feature_type="111000", **kw)
        same(pyoracle.icp(pr, P)[0], pyref.icp(pr, P)[0])


def test_duplicate_gate_and_stale_lists():
    rng = np.random.default_rng(9)
    tgt = planes_scene(rng, n_per=300)
    src = transformed_copy(tgt, np.linalg.inv(synth.se3(0.3, 0.1, 0.0, 0, 0, 0.01)))
    src[abi.PILLAR] = src[abi.PILLAR][:2]
    P = abi.default_params(used_feature_type="111000", dis_thre_unit=1.0, dis_thre_min=0.05, dis_thre_update_rate=2.0, max_iter_num=8,
                           min_neccessary_corr_ratio=0.0, apply_intersection_filter=0)
    same(pyoracle.icp(abi.PairData(tgt, src), P)[0], pyref.icp(abi.PairData(tgt, src), P)[0])


def test_normal_shooting_and_undistortion_variants(pairs_small):
    pair, _ = pairs_small[2]
    for kw in (dict(normal_shooting_on=1), dict(apply_motion_undistortion=1, used_feature_type="111110")):
        P = abi.default_params(**kw)
        same(pyoracle.icp(pair, P)[0], pyref.icp(pair, P)[0])


def test_golden_fixtures_agree_with_reference():
    import importlib.util
    import os

    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(here, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    for name in sorted(mg.CASES):
        pair, P, z = mg.load(name)
        if not P.faithful:
            continue  # the reference has no "intended" mode
        rr = pyref.icp(pair, P)[0]
        assert rr.code == int(z["code"]) and np.array_equal(rr.T_matrix(), z["T"]) and np.array_equal(rr.info_matrix(), z["info"])
        assert rr.sigma == float(z["sigma"])


def test_3dof_ground_variant(pairs_small):
    """lls_icp_3dof_ground (cregistration.hpp:1443-1582): ground-only (roll, pitch, z) ICP.  The reference returns its
    process code cast to bool, so only Trans1_2 and "code != 0" are observable."""
    for pair, _ in pairs_small:
        for kw in (dict(), dict(weight_strategy="0000"), dict(max_iter_num=3), dict(dis_thre_unit=0.5, max_bearable_rotation_d=0.01)):
            P = abi.default_params(weight_strategy="1111", max_bearable_rotation_d=10.0)
            for k, v in kw.items():
                setattr(P, k, v.encode() if isinstance(v, str) else v)
            ro = pyoracle.icp_3dof_ground(pair, P)[0]
            rr = pyref.icp_3dof_ground(pair, P)[0]
            assert (ro.code != 0) == (rr.code != 0)
            assert np.array_equal(ro.T_matrix(), rr.T_matrix())


def test_4dof_global_variant(pairs_small):
    """mm_lls_icp_4dof_global (cregistration.hpp:1584-1681): 8 heading trials of 45 degrees about the source station."""
    pair, T_gt = pairs_small[0]
    # rotate the source about its station so that only one heading trial can succeed
    yaw = np.deg2rad(135.0)
    spun = [pyoracle.transform(c, synth.se3(0, 0, 0, 0, 0, yaw)) for c in pair.src]
    pr = abi.PairData(pair.tgt, spun, tgt_bound=pair.tgt_bound)
    station = (0.0, 0.0, 0.0)
    (ro,), ok_o, best = pyoracle.icp_4dof_global(pr, 45.0, station, max_iter_num=12, dis_thre_unit=2.0)
    (rr,), ok_r = pyref.icp_4dof_global(pr, 45.0, station, max_iter_num=12, dis_thre_unit=2.0)
    assert ok_o == ok_r
    assert np.array_equal(ro.T_matrix(), rr.T_matrix()) and np.array_equal(ro.info_matrix(), rr.info_matrix())
    assert ro.sigma == rr.sigma and ro.confidence == rr.confidence
    if ok_o:
        dt, dr = synth.pose_error(ro.T_matrix() @ synth.se3(0, 0, 0, 0, 0, yaw), T_gt)
        assert dt < 0.3 and dr < 0.02, (dt, dr, best)


EDGE_PARAMS = {  # the values tests/test_gpu_fuzz.py::test_edge_parameter_values_match_oracle runs on the device
    "zero_iterations": dict(max_iter_num=0),
    "negative_iterations": dict(max_iter_num=-3),
    "many_iterations": dict(max_iter_num=300, converge_translation=0.0, converge_rotation_d=0.0),
    "zero_threshold": dict(dis_thre_unit=0.0, dis_thre_min=0.0),
    "min_above_unit": dict(dis_thre_unit=0.5, dis_thre_min=2.0),
    "growing_threshold": dict(dis_thre_update_rate=0.8, max_iter_num=12),
    # zero residual windows are left out: every weight becomes 0/0, the transform NaN, and the searches that follow run on NaN queries,
    # where the kd-tree stand-in of oracle/ref_shim and the oracle's own tree need not visit nodes alike (FLANN itself is unspecified there)
    "zero_balance": dict(z_xy_balanced_ratio=0.0),
    "negative_bearing": dict(normal_bearing=-10.0),
    "nothing_used": dict(used_feature_type="000000"),
    "only_vertex": dict(used_feature_type="000001"),
    "odd_flags": dict(used_feature_type="1x1 01", weight_strategy="2a01"),
    "zero_rate": dict(dis_thre_update_rate=0.0, max_iter_num=4),
    "huge_threshold": dict(dis_thre_unit=400.0, dis_thre_min=100.0, max_iter_num=3),
}


@pytest.mark.parametrize("name", sorted(EDGE_PARAMS))
def test_edge_parameter_values_oracle_equals_reference(pairs_small, name):
    """Parameter values nobody would configure but the reference accepts without a check: the restatement follows the reference's
    own lines there too (the device is compared with the oracle on the same values)."""
    P = abi.default_params(**EDGE_PARAMS[name])
    pair, _ = pairs_small[0]
    same(pyoracle.icp(pair, P)[0], pyref.icp(pair, P)[0])


@pytest.mark.parametrize("block", range(4))
def test_random_option_points_oracle_equals_reference(pairs_small, block):
    """The seeded random points of the 23-dimensional option space that tests/test_gpu_fuzz.py runs on the device (same generator):
    here the oracle against the reference's own lines, bit for bit.  `faithful` is forced on (the reference has no other mode) and
    `keep_less_source_points` off (upstream thins with a time-seeded pcl::RandomSample; the ABI defines its own seeded selection)."""
    from test_gpu_fuzz import random_params

    rng = np.random.default_rng(900 + block)
    for k in range(12):
        base, T_gt = pairs_small[int(rng.integers(0, len(pairs_small)))]
        pert = synth.se3(*rng.normal(0, 0.25, 3), *np.deg2rad(rng.normal(0, 0.6, 3)))
        pair = abi.PairData(base.tgt, base.src, init_guess=pert @ T_gt, tgt_bound=base.tgt_bound)
        P = random_params(rng)
        P.faithful = 1
        P.keep_less_source_points = 0
        same(pyoracle.icp(pair, P)[0], pyref.icp(pair, P)[0])


@pytest.mark.parametrize("kind", ["duplicates", "one_cell", "far_origin", "collinear", "sparse_far", "ragged"])
def test_degenerate_inputs_oracle_equals_reference(kind):
    """The pathological inputs of tests/test_gpu_fuzz.py (exact duplicates, everything in one cell, coordinates around 1e5 m,
    collinear clouds, nothing within reach, empty and tiny class clouds): the reference's lines have no special cases for them,
    the oracle must follow them bit for bit."""
    import zlib

    from test_gpu_fuzz import degenerate_pair, random_params

    rng = np.random.default_rng(zlib.crc32(kind.encode()) + 1)
    for k in range(8):
        pair = degenerate_pair(rng, kind)
        P = random_params(rng)
        P.apply_motion_undistortion = 0
        P.faithful = 1
        P.keep_less_source_points = 0
        same(pyoracle.icp(pair, P)[0], pyref.icp(pair, P)[0])


@pytest.mark.parametrize("block", range(2))
def test_variants_random_oracle_equals_reference(pairs_small, block):
    """The two variants under seeded random options / headings / stations (the cases tests/test_gpu_variants.py runs on the device)."""
    from test_gpu_fuzz import degenerate_pair, random_params

    rng = np.random.default_rng(7700 + block)
    for k in range(10):  # lls_icp_3dof_ground
        if k % 3 == 2:
            pair = degenerate_pair(rng, ["duplicates", "one_cell", "collinear", "ragged", "sparse_far"][int(rng.integers(0, 5))])
        else:
            base, _ = pairs_small[int(rng.integers(0, len(pairs_small)))]
            tilt = synth.se3(0, 0, rng.normal(0, 0.1), *np.deg2rad(rng.normal(0, 0.5, 2)), 0)
            pair = abi.PairData(base.tgt, [pyoracle.transform(c, tilt) if c is not None and len(c) else c for c in base.src], tgt_bound=base.tgt_bound)
        P = random_params(rng)
        P.apply_motion_undistortion = 0
        P.keep_less_source_points = 0
        ro = pyoracle.icp_3dof_ground(pair, P)[0]
        rr = pyref.icp_3dof_ground(pair, P)[0]
        assert (ro.code != 0) == (rr.code != 0)
        assert np.array_equal(ro.T_matrix(), rr.T_matrix(), equal_nan=True)
    rng = np.random.default_rng(8800 + block)
    for k in range(4):  # mm_lls_icp_4dof_global
        base, _ = pairs_small[int(rng.integers(0, len(pairs_small)))]
        spin = synth.se3(*rng.normal(0, 0.3, 3), 0, 0, rng.uniform(-np.pi, np.pi))
        pr = abi.PairData(base.tgt, [pyoracle.transform(c, spin) for c in base.src], tgt_bound=base.tgt_bound)
        station = tuple(rng.normal(0, 1.0, 3))
        step = float(rng.choice([20.0, 45.0, 72.0, 90.0, 180.0]))
        kw = dict(max_iter_num=int(rng.integers(3, 15)), dis_thre_unit=float(rng.uniform(1.0, 3.0)))
        (ro,), ok_o, _ = pyoracle.icp_4dof_global(pr, step, station, **kw)
        (rr,), ok_r = pyref.icp_4dof_global(pr, step, station, **kw)
        assert ok_o == ok_r
        assert np.array_equal(ro.T_matrix(), rr.T_matrix(), equal_nan=True) and np.array_equal(ro.info_matrix(), rr.info_matrix(), equal_nan=True)
        assert ro.sigma == rr.sigma or (np.isnan(ro.sigma) and np.isnan(rr.sigma))
