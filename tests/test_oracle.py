"""CPU tests of the oracle itself.  The reference ships no golden vectors, so the oracle is
guarded by analytic known-answer tests and by its two independent NN back-ends agreeing."""
import numpy as np
import pytest

from conftest import planes_scene, transformed_copy
from mulls_amd import abi, synth
from oracle import pyoracle


def test_kdtree_equals_bruteforce(pairs_small):
    pair, _ = pairs_small[0]
    for c in (abi.GROUND, abi.PILLAR, abi.FACADE, abi.VERTEX):
        m0, d0 = pyoracle.nn(pair.src[c], pair.tgt[c], 0)
        m1, d1 = pyoracle.nn(pair.src[c], pair.tgt[c], 1)
        assert np.array_equal(m0, m1) and np.array_equal(d0, d1)


def test_nn_ties_resolve_to_lowest_index():
    rng = np.random.default_rng(0)
    base = rng.uniform(-5, 5, (50, 3)).astype(np.float32)
    tgt = abi.make_points(np.concatenate([base, base, base]))  # every target exists three times
    src = abi.make_points(base + np.float32(0.01))
    for mode in (0, 1):
        m, _ = pyoracle.nn(src, tgt, mode)
        assert (m < 50).all()


def test_transform_matches_float64_reference():
    rng = np.random.default_rng(1)
    pts = abi.make_points(rng.uniform(-50, 50, (1000, 3)), rng.normal(size=(1000, 3)))
    T = synth.se3(0.3, -1.2, 0.05, 0.01, -0.02, 0.3)
    out = pyoracle.transform(pts, T)
    xyz = np.column_stack([pts["x"], pts["y"], pts["z"]]).astype(np.float64)
    exp = (xyz @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    got = np.column_stack([out["x"], out["y"], out["z"]])
    # same double-precision sum order is not guaranteed by numpy: allow 1 ulp
    assert np.abs(got - exp).max() <= np.spacing(np.abs(exp).max())
    assert np.array_equal(out["intensity"], pts["intensity"])


def test_identity_transform_is_bit_exact():
    rng = np.random.default_rng(2)
    pts = abi.make_points(rng.uniform(-50, 50, (257, 3)), rng.normal(size=(257, 3)))
    out = pyoracle.transform(pts, np.eye(4))
    for k in abi.POINT_DTYPE.names:
        assert np.array_equal(out[k].view(np.uint32), pts[k].view(np.uint32))  # bit-exact, field by field


def test_single_pt2pl_correspondence_known_answer():
    """A.10-ii: one point-to-plane correspondence, 27 accumulators computed by hand in float32."""
    f = np.float32
    p = np.array([1.5, -2.0, 0.25], f)
    q = np.array([1.25, -2.5, 0.0], f)
    n = np.array([0.0, 0.6, 0.8], f)
    src = abi.make_points([p], None, [10.0])
    tgt = abi.make_points([q], [n], [30.0])
    out, w = pyoracle.accumulate(0, src, tgt, [0], [0], [0.0], 0, 1.0, 0, 0, 0, 0.1)
    a = f(n[2] * p[1]) - f(n[1] * p[2])
    b = f(n[0] * p[2]) - f(n[2] * p[0])
    c = f(n[1] * p[0]) - f(n[0] * p[1])
    d = f(f(f(f(f(f(n[0] * q[0]) + f(n[1] * q[1])) + f(n[2] * q[2])) - f(n[0] * p[0])) - f(n[1] * p[1])) - f(n[2] * p[2]))
    r = np.array([n[0], n[1], n[2], a, b, c], f)
    exp = []
    for i in range(6):
        for j in range(i, 6):
            lo, hi = (r[i], r[j])
            # reference writes w*r_j*r_i with the later component first for mixed terms; products commute exactly
            exp.append(float(f(f(f(1.0) * hi) * lo)) if j >= 3 > i else float(f(f(f(1.0) * lo) * hi)))
    exp += [float(f(f(f(1.0) * d) * r[i])) for i in range(6)]
    assert w[0] == 1.0
    assert np.allclose(out, exp, rtol=0, atol=1e-7)
    # the packed 21-term layout is symmetric-complete: rebuild N and check N = r r^T
    N = np.zeros((6, 6))
    k = 0
    for i in range(6):
        for j in range(i, 6):
            N[i, j] = N[j, i] = out[k]
            k += 1
    assert np.allclose(N, np.outer(r, r).astype(np.float64), atol=1e-6)


def test_weight_chain_order_and_values():
    f = np.float32
    src = abi.make_points([[3.0, 4.0, 0.5]], None, [100.0])
    tgt = abi.make_points([[3.0, 4.0, 0.0]], [[0, 0, 1]], [40.0])
    # dist weight only: iter 0 -> b = 0.7, dist = 5 -> 0.7 + 0.3*5/30 = 0.75
    _, w = pyoracle.accumulate(0, src, tgt, [0], [0], [0.0], 0, 1.0, 1, 0, 0, 0.1)
    assert w[0] == f(0.7 + (1.0 - float(f(0.7))) * 5.0 / 30.0)
    # residual weight: |d| = 0.5 > 0.1 -> (2*0.5*0.1 - 0.01)/0.25 = 0.36
    _, w = pyoracle.accumulate(0, src, tgt, [0], [0], [0.0], 3, 1.0, 0, 1, 0, 0.1)
    assert abs(float(w[0]) - 0.36) < 1e-6
    # intensity weight exp(-|100-40|/255)
    _, w = pyoracle.accumulate(0, src, tgt, [0], [0], [0.0], 0, 1.0, 0, 0, 1, 0.1)
    assert abs(float(w[0]) - np.exp(-60.0 / 255.0)) < 1e-6


def test_pt2pt_does_not_write_weight():
    src = abi.make_points([[1.0, 2.0, 3.0]], None, [0.0])
    tgt = abi.make_points([[1.1, 2.0, 3.0]], None, [0.0])
    _, w = pyoracle.accumulate(2, src, tgt, [0], [0], [0.0123], 0, 1.0, 1, 0, 1, 0.1)
    assert w[0] == np.float32(0.0123)  # the union still holds the squared distance


def test_duplicate_gate_499_vs_500():
    """A.10-iii: duplicate rule and permanent compaction only when |source| >= 500."""
    rng = np.random.default_rng(3)
    tgt = abi.make_points(rng.uniform(-20, 20, (50, 3)), np.tile([0, 0, 1], (50, 1)))
    for n, gated in ((499, False), (500, True)):
        base = rng.integers(0, 50, n)
        xyz = np.column_stack([tgt["x"], tgt["y"], tgt["z"]])[base] + rng.normal(0, 0.01, (n, 3)).astype(np.float32)
        src = abi.make_points(xyz, np.tile([0, 0, 1], (n, 1)))
        match, d2, flags = pyoracle.correspond(src, tgt, 1.0, True, 45.0)
        assert (match >= 0).all()
        alive = (flags & 1).astype(bool)
        if gated:
            # exactly one survivor per distinct target: the lowest source index
            assert alive.sum() == len(np.unique(match))
            for t in np.unique(match):
                assert alive[np.nonzero(match == t)[0][0]]
        else:
            assert alive.all() and ((flags & 2) > 0).all()


def test_radius_and_distance_rejectors():
    tgt = abi.make_points([[0, 0, 0], [10, 0, 0], [20, 0, 0]], np.tile([0, 0, 1], (3, 1)))
    # thr = 1: NN radius 2.5, distance rejector 1.0
    src = abi.make_points([[0.5, 0, 0], [12.0, 0, 0], [23.0, 0, 0], [10.0, 0.999, 0]], np.tile([0, 0, 1], (4, 1)))
    match, d2, flags = pyoracle.correspond(src, tgt, 1.0, True, 45.0)
    assert list(match) == [0, 1, -1, 1]
    assert list(flags & 2) == [2, 0, 0, 2]


def test_normal_check_uses_abs_cosine():
    tgt = abi.make_points([[0, 0, 0], [5, 0, 0], [10, 0, 0]], [[0, 0, 1], [0, 0, 1], [0, 0, 1]])
    c30, s30 = np.cos(np.deg2rad(30)), np.sin(np.deg2rad(30))
    src = abi.make_points([[0.1, 0, 0], [5.1, 0, 0], [10.1, 0, 0]], [[0, 0, -1], [s30, 0, c30], [1, 0, 0]])
    _, _, flags = pyoracle.correspond(src, tgt, 1.0, True, 45.0)
    assert list(flags & 2) == [2, 2, 0]
    _, _, flags = pyoracle.correspond(src, tgt, 1.0, True, 20.0)
    assert list(flags & 2) == [2, 0, 0]
    _, _, flags = pyoracle.correspond(src, tgt, 1.0, False, 20.0)
    assert list(flags & 2) == [2, 2, 2]


def test_recovers_known_offset():
    """A.10-i: planes + poles, source = target moved by a known small motion -> ICP recovers its inverse."""
    rng = np.random.default_rng(4)
    tgt = planes_scene(rng)
    T_true = synth.se3(0.20, -0.15, 0.10, 0.01, -0.008, 0.02)  # source -> target
    src = transformed_copy(tgt, np.linalg.inv(T_true))
    pair = abi.PairData(tgt, src)
    P = abi.default_params(used_feature_type="111000", weight_strategy="1000", converge_translation=1e-7, converge_rotation_d=1e-6,
                           max_iter_num=30)
    r = pyoracle.icp(pair, P, trace_cap=40)[0]
    assert r.code == 1
    dt, dr = synth.pose_error(r.T_matrix(), T_true)
    assert dt < 2e-5 and dr < 2e-6, (dt, dr)
    assert r.sigma < 1e-4
    assert r.iters <= 12


def test_pillar_only_normal_matrix_is_diagonal_when_faithful():
    """A.10-iv: the mirror overwrites pt2li's off-diagonal terms."""
    rng = np.random.default_rng(5)
    tgt = planes_scene(rng, n_per=600)
    src = transformed_copy(tgt, np.linalg.inv(synth.se3(0.05, 0.02, 0.0, 0, 0, 0.004)))
    pair = abi.PairData(tgt, src)
    P = abi.default_params(used_feature_type="010000", weight_strategy="0000", min_neccessary_corr_ratio=0.0, max_iter_num=2)
    r = pyoracle.icp(pair, P, trace_cap=4)[0]
    assert r.trace_len >= 1
    N = np.array(r.trace[0].atpa[:]).reshape(6, 6)
    assert np.count_nonzero(N - np.diag(np.diag(N))) == 0
    P.faithful = 0
    r = pyoracle.icp(pair, P, trace_cap=4)[0]
    N = np.array(r.trace[0].atpa[:]).reshape(6, 6)
    assert np.count_nonzero(N - np.diag(np.diag(N))) > 0 and np.allclose(N, N.T)


def test_vertex_residual_weighted_by_squared_distance():
    """A.10-v: vertex-only registration -> sigma^2 = sum(d2_k * |r_k|^2)/(3K-6) in faithful mode."""
    rng = np.random.default_rng(6)
    xyz = rng.uniform(-10, 10, (300, 3))
    tgt = [None] * 5 + [abi.make_points(xyz)]
    src = [None] * 5 + [abi.make_points(xyz + rng.normal(0, 0.02, xyz.shape))]
    # needs >= 20 "necessary" correspondences: give facade clouds too, but keep them out of the residual by a tiny set
    fac = abi.make_points(rng.uniform(-10, 10, (60, 3)) * [1, 0, 1] + [0, 9, 0], np.tile([0, -1, 0], (60, 1)))
    tgt[abi.FACADE] = fac
    src[abi.FACADE] = fac.copy()
    pair = abi.PairData(tgt, src)
    P = abi.default_params(used_feature_type="001001", weight_strategy="0000", max_iter_num=1, apply_intersection_filter=0)
    r1 = pyoracle.icp(pair, P)[0]
    P.faithful = 0
    r0 = pyoracle.icp(pair, P)[0]
    assert r1.code in (1, -3) and r0.code in (1, -3)
    # d2 ~ 1e-3 scale weights make the faithful sigma much smaller than the intended one
    assert r1.sigma < 0.2 * r0.sigma


def test_process_codes():
    rng = np.random.default_rng(7)
    tgt = planes_scene(rng)
    src = transformed_copy(tgt, np.linalg.inv(synth.se3(0.1, 0, 0)))
    pair = abi.PairData(tgt, src)
    ok = pyoracle.icp(pair, abi.default_params(used_feature_type="111000"))[0]
    assert ok.code == 1
    # too few correspondences: sources far away
    far = transformed_copy(tgt, synth.se3(200.0, 0, 0))
    r = pyoracle.icp(abi.PairData(tgt, far), abi.default_params(used_feature_type="111000", apply_intersection_filter=0))[0]
    assert r.code == -2 and r.iters == 1 and np.allclose(r.info_matrix(), np.eye(6)) and r.sigma == 1.0
    # step too large: tiny allowed rotation
    r = pyoracle.icp(abi.PairData(tgt, transformed_copy(tgt, synth.se3(0, 0, 0, 0, 0, 0.05))),
                     abi.default_params(used_feature_type="111000", max_bearable_rotation_d=0.1))[0]
    assert r.code == -1
    # sigma too large
    r = pyoracle.icp(pair, abi.default_params(used_feature_type="111000", sigma_thre=1e-9))[0]
    assert r.code == -3
    # loop never runs
    r = pyoracle.icp(pair, abi.default_params(max_iter_num=0))[0]
    assert r.code == 0 and r.iters == 0 and np.allclose(r.T_matrix(), np.eye(4))


def test_host_pieces_against_numpy():
    rng = np.random.default_rng(8)
    x = np.array([0.3, -0.2, 0.1, 0.02, -0.03, 0.5])
    T = pyoracle.construct_trans(x)
    assert np.allclose(T, synth.se3(*x), atol=1e-15)
    assert abs(pyoracle.rotation_angle(T) - np.arccos((np.trace(T[:3, :3]) - 1) / 2)) < 1e-12
    A = rng.normal(size=(20, 6))
    N = A.T @ A
    b = rng.normal(size=6)
    rc, xs, cof = pyoracle.solve(N, b)
    assert rc == 0 and np.allclose(xs, np.linalg.solve(N, b), rtol=1e-10)
    assert np.allclose(cof[:3, :3], np.linalg.inv(N)[:3, :3], rtol=1e-9)


def test_omp_sections_do_not_change_results(pairs_small):
    pair, _ = pairs_small[1]
    P = abi.kitti_params(dis_thre_unit=2.4)
    a = pyoracle.icp(pair, P, use_omp=1)[0]
    b = pyoracle.icp(pair, P, use_omp=0)[0]
    c = pyoracle.icp(pair, P, nn_mode=1)[0]
    assert a.T[:] == b.T[:] == c.T[:] and a.code == b.code == c.code and a.sigma == b.sigma == c.sigma


def test_synthetic_pairs_converge_near_ground_truth(pairs_small):
    for pair, T_gt in pairs_small:
        r = pyoracle.icp(pair, abi.kitti_params(dis_thre_unit=2.4))[0]
        assert r.code == 1
        dt, dr = synth.pose_error(r.T_matrix(), T_gt)
        assert dt < 0.05 and dr < 2e-3


def test_keep_less_source_points_is_seeded_and_order_preserving(pairs_small):
    pair, _ = pairs_small[0]
    P = abi.default_params(keep_less_source_points=1, rng_seed=5, max_iter_num=2)
    a = pyoracle.icp(pair, P)[0]
    b = pyoracle.icp(pair, P)[0]
    assert a.T[:] == b.T[:] and list(a.nsrc0) == list(b.nsrc0)
    assert a.ntgt0[abi.GROUND] == a.cropped * 0 + (pyoracle.icp(pair, abi.default_params(max_iter_num=2))[0].ntgt0[abi.GROUND] // 2)
    P.rng_seed = 6
    c = pyoracle.icp(pair, P)[0]
    assert list(c.ntgt0) == list(a.ntgt0) and c.T[:] != a.T[:]
