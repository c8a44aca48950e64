"""Third-party operators the path relies on and that cannot be checked against PCL / FLANN / Eigen sources in this image
(SURVEY.md Appendix C restates them from memory of PCL 1.8-1.10): where two plausible readings of the upstream code exist,
count how often the bench workload lands on an input that tells them apart.

  pcl/registration/impl/correspondence_estimation.hpp  CorrespondenceEstimation::determineCorrespondences(corrs, double max_distance)
      `if (distance[0] > max_dist_sqr) continue;` with max_dist_sqr = max_distance * max_distance in double  — restated as is;
      differs from `>=` only when (double)d2 == max_dist_sqr exactly                              -> census "radius_equal"
  pcl/registration/correspondence_rejection_distance.{h,cpp}  CorrespondenceRejectorDistance::getRemainingCorrespondences
      setMaximumDistance(float d) stores d * d; PCL 1.7-1.12 keep a correspondence when `distance < max_distance_` (NaN dropped) —
      the library's default, mulls_params.rejector_strict = 1 —; SURVEY A.4-3 had written `!(distance > max)` (NaN kept), still
      selectable as rejector_strict = 0                                                           -> census "rejector_equal", "rejector_nan"
  flann/algorithms/kdtree_single_index.h  KDTreeSingleIndex::findNeighbors (exact, epsilon 0, L2_Simple<float>)
      the order among targets at exactly the same distance is implementation-defined; here: lowest index     -> census "brute_ties"
  Eigen/src/LU/PartialPivLU.h  inverse() of a fixed 6x6 — row-pivoted LU, columns solved against the identity; restated in
      hostmath.h / the oracle; last-bit differences in the back substitution order would not change any integer output.

The counts are reported, and bounded: should a workload ever hit these inputs at a rate that matters, the test fails and says so."""
import numpy as np
import pytest

from mulls_amd import abi, synth
from oracle import pyoracle


def _bench_like_pairs(n):
    import bench

    scenes = bench.build_scenes(n, False, 4)
    return [s[0] for s in scenes]


@pytest.mark.timeout(600)
def test_operator_census_on_the_bench_workload():
    pairs = _bench_like_pairs(6)
    P = abi.kitti_params(converge_translation=0.0, converge_rotation_d=0.0)
    pyoracle.census(reset=True)
    for p in pairs:
        r = pyoracle.icp(p, P, nn_mode=1)[0]  # brute force: the tie counter sees every target of every query
        assert r.code == 1 and r.iters == 20
    c = pyoracle.census(reset=True)
    print("operator census over %d registrations x 20 iterations: %s" % (len(pairs), c))
    assert c["radius_tests"] > 200000 and c["rejector_tests"] > 100000 and c["brute_queries"] == c["radius_tests"]
    # none of the boundary inputs occurs: the two readings of each operator give the same result on this workload
    assert c["radius_equal"] == 0
    assert c["rejector_equal"] == 0 and c["rejector_nan"] == 0
    # exact ties of the nearest distance need two targets at the same float distance from a query: rare, reported
    assert c["brute_ties"] <= c["brute_queries"] * 1e-4, c


def test_strict_rejector_differs_only_on_the_boundary():
    """rejector_strict = 1 (`distance < max^2`) and 0 (`distance <= max^2`): identical results unless a correspondence sits exactly
    on the threshold — shown on the bench-like pairs (identical) and on a constructed boundary case (one correspondence apart)."""
    pairs = _bench_like_pairs(2)
    P0 = abi.kitti_params(max_iter_num=6, rejector_strict=0)
    P1 = abi.kitti_params(max_iter_num=6)
    assert P1.rejector_strict == 1  # the default is PCL's operator
    for p in pairs:
        a, b = pyoracle.icp(p, P0)[0], pyoracle.icp(p, P1)[0]
        assert list(a.T) == list(b.T) and list(a.ncorr) == list(b.ncorr) and a.iters == b.iters
    # boundary: a target grid, sources displaced by exactly thr along x (float-exact numbers): distance == thr^2
    thr = 0.5
    g = np.arange(-8, 9, dtype=np.float32) * 2.0
    X, Y = np.meshgrid(g, g)
    tgt = np.stack([X.ravel(), Y.ravel(), np.zeros(X.size, np.float32)], 1)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (len(tgt), 1))
    src = tgt.copy()
    src[:, 0] += thr
    tp = abi.make_points(tgt, nrm, np.full(len(tgt), 10, np.float32), np.zeros(len(tgt), np.float32))
    sp = abi.make_points(src, nrm, np.full(len(tgt), 10, np.float32), np.zeros(len(tgt), np.float32))
    m0, d0, f0 = pyoracle.correspond(sp, tp, thr)
    assert (d0 == np.float32(thr * thr)).all() and not (f0 & 2).any()  # the stage entry point runs the default `<`: none kept
    pair = boundary_pair()
    kw = dict(used_feature_type="100000", max_iter_num=1, dis_thre_unit=thr, apply_intersection_filter=0)
    assert pyoracle.icp(pair, abi.default_params(rejector_strict=0, **kw))[0].ncorr[0] == 289  # `<=`: every correspondence kept
    assert pyoracle.icp(pair, abi.default_params(**kw))[0].ncorr[0] == 0  # `<` (default): none


def boundary_pair(thr=0.5):
    """289 ground correspondences whose float distance equals the float threshold thr * thr exactly."""
    g = np.arange(-8, 9, dtype=np.float32) * 2.0
    X, Y = np.meshgrid(g, g)
    tgt = np.stack([X.ravel(), Y.ravel(), np.zeros(X.size, np.float32)], 1)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (len(tgt), 1))
    src = tgt.copy()
    src[:, 0] += thr
    mk = lambda xyz: abi.make_points(xyz, nrm, np.full(len(xyz), 10, np.float32), np.zeros(len(xyz), np.float32))
    return abi.PairData([mk(tgt)] + [None] * 5, [mk(src)] + [None] * 5)


@pytest.mark.gpu
def test_strict_rejector_on_the_device(ctx):
    """Both forms of the distance rejector on the device, against the oracle, on the constructed boundary case and on a scan pair."""
    pair = boundary_pair()
    kw = dict(used_feature_type="100000", max_iter_num=1, dis_thre_unit=0.5, apply_intersection_filter=0)
    for strict in (0, 1):
        P = abi.default_params(rejector_strict=strict, **kw)
        rg, ro = ctx.icp(pair, P)[0], pyoracle.icp(pair, P)[0]
        assert list(rg.ncorr) == list(ro.ncorr) and rg.code == ro.code and rg.ncorr[0] == (0 if strict else 289)
    sp, _ = synth.make_pair(11, n_beams=32, n_az=900, src_counts={0: 600, 1: 300, 2: 700}, tgt_counts={0: 2500, 1: 900, 2: 3000}, vertex_count=0)
    P = abi.kitti_params(dis_thre_unit=2.4, rejector_strict=1)
    rg, ro = ctx.icp(sp, P)[0], pyoracle.icp(sp, P)[0]
    assert rg.code == ro.code == 1 and rg.iters == ro.iters and list(rg.ncorr) == list(ro.ncorr)
    dt, dr = synth.pose_error(np.array(rg.T[:]).reshape(4, 4).T, np.array(ro.T[:]).reshape(4, 4).T)
    assert dt <= 1e-7 and dr <= 1e-7


def test_ransac_stopping_rule_is_pcl_s_and_never_near_its_boundary():
    """pcl::RandomSampleConsensus::computeModel stops on `iterations_ < k`, k = log(1 - probability_) / log(1 - pow(w, 3)) (ransac.hpp).  The oracle evaluates
    exactly that (std::pow, std::log); the device evaluates the libm-free equivalent p_no_outliers^iterations > 1 - probability_.  The two can only part when
    `iterations` is within rounding of k: counted on the demo scan's and synthetic scans' ground cells (normal method 3, every shipped configuration) — no
    evaluation within 1e-9 of the boundary, no disagreement of the two forms."""
    import os

    from mulls_amd import synth

    scans = []
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "demo_pair.npz")
    if os.path.exists(gold):
        g = np.load(gold)
        for k in ("scan_0", "scan_15"):
            s = g[k]
            scans.append(abi.make_points(s[:, :3], np.zeros_like(s[:, :3]), s[:, 3], np.zeros(len(s), np.float32)))
    for seed in (3, 4):
        scene = synth.Scene(seed)
        sc = synth.raycast(scene, synth.se3(0, 0, scene.sensor_height), 64, 1200, seed=seed)
        scans.append(abi.make_points(sc["xyz"], np.zeros_like(sc["xyz"]), sc["intensity"], sc["t"]))
    G = abi.ground_params(estimate_ground_normal_method=3)
    pyoracle.census(reset=True)
    cells = 0
    for pts in scans:
        ground = pyoracle.ground_filter(pts, G)[0]
        cells += len(ground) > 0
    c = pyoracle.census(reset=True)
    print("plane RANSAC stopping test over %d scans: %s" % (len(scans), {k: v for k, v in c.items() if k.startswith("ransac")}))
    assert cells == len(scans) and c["ransac_tests"] > 1000
    assert c["ransac_near_boundary"] == 0 and c["ransac_forms_disagree"] == 0


def test_census_of_what_flann_and_eigen_leave_open():
    """pcl::KdTreeFLANN / pcl::PCA are restated (oracle/pcl_restated.h, "parity unpinned"): where could a run with the real libraries differ?  Counted on the
    non-ground clouds of the demo scans and of two synthetic scans through classify_nground's whole chain (neighbourhood PCA, labels, non_max_suppress):
      * a max_nn cut that falls inside a group of equal distances (FLANN decides which of them stay),
      * a candidate exactly on the search radius (`<` against `<=`),
      * neighbours at equal distances (their order is the order of the PCA's float sums),
      * a label decision whose eigenvalue ratio / eigenvector component lies within 1e-5 of its threshold (Eigen's float solver is good to ~1e-6).
    The counts bound the exposure: they are reported, and the first two — which would change a neighbourhood — have to stay rare."""
    import os

    from mulls_amd import synth

    scans = []
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "demo_pair.npz")
    if os.path.exists(gold):
        g = np.load(gold)
        for k in ("scan_0", "scan_15"):
            s = g[k]
            scans.append(abi.make_points(s[:, :3], np.zeros_like(s[:, :3]), s[:, 3], np.zeros(len(s), np.float32)))
    for seed in (3, 4):
        scene = synth.Scene(seed)
        sc = synth.raycast(scene, synth.se3(0, 0, scene.sensor_height), 64, 1200, seed=seed)
        scans.append(abi.make_points(sc["xyz"], np.zeros_like(sc["xyz"]), sc["intensity"], sc["t"]))
    G, P = abi.ground_params(), abi.classify_params()
    pyoracle.search_census(reset=True), pyoracle.classify_census(reset=True)
    n_pts = 0
    for pts in scans:
        unground = pyoracle.ground_filter(pts, G)[2]
        out, _ = pyoracle.classify_nground(unground, P)
        n_pts += len(unground)
        assert sum(len(c) for c in out) > 0
    s, c = pyoracle.search_census(reset=True), pyoracle.classify_census(reset=True)
    print("restated searches on %d scans (%d non-ground points): %s; label decisions: %s" % (len(scans), n_pts, s, c))
    assert s["searches"] > 10000 and c["decisions"] > 1000
    # a neighbourhood that could differ with the real FLANN: fewer than one search in a thousand
    assert s["cut_in_tie"] + s["on_radius"] <= s["searches"] // 1000
    # labels within reach of the eigen-solver's accuracy: fewer than one decision in a thousand
    assert c["within_1e5"] <= max(1, c["decisions"] // 1000)
