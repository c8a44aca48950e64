"""Randomised differential test: seeded random points in the 23-dimensional option space of mm_lls_icp, on small pairs with
random initial guesses, HIP (every search tier) against the oracle.  Same comparison as tests/test_gpu_icp.py::compare."""
import zlib

import numpy as np
import pytest

from mulls_amd import abi, synth
from oracle import pyoracle
from test_gpu_icp import compare

pytestmark = pytest.mark.gpu


def random_params(rng):
    used = "".join(rng.choice(["0", "1"], p=[0.3, 0.7]) for _ in range(6))
    if used[:3].count("1") == 0:
        used = "1" + used[1:]
    kw = dict(
        max_iter_num=int(rng.integers(1, 26)),
        dis_thre_unit=float(rng.uniform(0.6, 3.0)),
        converge_translation=float(rng.choice([0.0, 0.0005, 0.002, 0.01])),
        converge_rotation_d=float(rng.choice([0.0, 0.001, 0.01, 0.05])),
        dis_thre_min=float(rng.uniform(0.2, 0.8)),
        dis_thre_update_rate=float(rng.uniform(1.02, 1.4)),
        used_feature_type=used,
        weight_strategy="".join(rng.choice(["0", "1"]) for _ in range(4)),
        z_xy_balanced_ratio=float(rng.uniform(0.5, 2.0)),
        pt2pt_residual_window=float(rng.uniform(0.02, 0.3)),
        pt2pl_residual_window=float(rng.uniform(0.02, 0.3)),
        pt2li_residual_window=float(rng.uniform(0.02, 0.3)),
        apply_intersection_filter=int(rng.random() < 0.7),
        apply_motion_undistortion=int(rng.random() < 0.15),
        normal_shooting_on=int(rng.random() < 0.15),
        normal_bearing=float(rng.uniform(10.0, 60.0)),
        keep_less_source_points=int(rng.random() < 0.15),
        faithful=int(rng.random() < 0.8),
        sigma_thre=float(rng.choice([0.05, 0.35, 0.5, 5.0])),
        min_neccessary_corr_ratio=float(rng.choice([0.0, 0.03, 0.3])),
        max_bearable_rotation_d=float(rng.choice([0.5, 10.0, 45.0])),
        rng_seed=int(rng.integers(0, 2**31)),
    )
    return abi.default_params(**kw)


@pytest.mark.parametrize("block", range(4))
def test_random_option_points_match_oracle(ctx, pairs_small, block):
    rng = np.random.default_rng(900 + block)
    codes = set()
    for k in range(12):
        base, T_gt = pairs_small[int(rng.integers(0, len(pairs_small)))]
        pert = synth.se3(*rng.normal(0, 0.25, 3), *np.deg2rad(rng.normal(0, 0.6, 3)))
        pair = abi.PairData(base.tgt, base.src, init_guess=pert @ T_gt, tgt_bound=base.tgt_bound)
        P = random_params(rng)
        ro = pyoracle.icp(pair, P, trace_cap=32)[0]
        rg = ctx.icp(pair, P, trace_cap=32)[0]
        compare(ro, rg)
        codes.add(ro.code)
    assert 1 in codes


@pytest.mark.parametrize("block", range(3))
def test_random_options_large_batch_equals_single(ctx_auto, pairs_small, block):
    """Batches big enough for class-level LDS jobs (duplicate rule and rejection chain inside the search kernel) and, from
    2048 pairs, two sub-batches in flight: per-pair results must be bit-identical to the single-pair path (chunk-level
    jobs, separate k_filter) for random option points, including the options that switch the fused path off."""
    rng = np.random.default_rng(1300 + block)
    n = [700, 1100, 2300][block]
    pairs = []
    for k in range(n):
        base, T_gt = pairs_small[int(rng.integers(0, len(pairs_small)))]
        pert = synth.se3(*rng.normal(0, 0.25, 3), *np.deg2rad(rng.normal(0, 0.6, 3)))
        pairs.append(abi.PairData(base.tgt, base.src, init_guess=pert @ T_gt, tgt_bound=base.tgt_bound))
    for trial in range(2):
        P = random_params(rng)
        if block == 0 and trial == 1:
            P.normal_shooting_on = 1
        rb = ctx_auto.icp_batch(pairs, P)
        for i in rng.choice(n, 25, replace=False):
            r1 = ctx_auto.icp(pairs[int(i)], P)[0]
            assert (r1.code, r1.iters, list(r1.ncorr), r1.singular) == (rb[i].code, rb[i].iters, list(rb[i].ncorr), rb[i].singular), (trial, i)
            assert np.array_equal(np.array(r1.T[:]), np.array(rb[i].T[:]), equal_nan=True), (trial, i)
            assert np.array_equal(np.array(r1.info[:]), np.array(rb[i].info[:]), equal_nan=True), (trial, i)
            assert r1.sigma == rb[i].sigma or (np.isnan(r1.sigma) and np.isnan(rb[i].sigma))


def degenerate_pair(rng, kind, nt=None, ns=None):
    """Small pathological inputs (finite values): the reference has no special cases for them, neither may the device.
    nt / ns: six target / source sizes (tools/gpu_fuzz_mixed.py: class clouds of every search tier in one pair); default one small size for all."""
    n_t, n_s = int(rng.integers(3, 900)), int(rng.integers(3, 700))
    nt = [n_t] * 6 if nt is None else [int(v) for v in nt]
    ns = [n_s] * 6 if ns is None else [int(v) for v in ns]

    def cloud(n, spread, offset=0.0):
        xyz = rng.normal(0, spread, (n, 3)) + offset
        nrm = rng.normal(0, 1, (n, 3))
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        return abi.make_points(xyz, nrm, rng.uniform(0, 255, n))

    if kind == "duplicates":  # many identical points: distance ties everywhere, the lowest target index must win
        base = cloud(7, 3.0)
        tgt = [base[rng.integers(0, 7, nt[c])] for c in range(6)]
        src = [base[rng.integers(0, 7, ns[c])] for c in range(6)]
    elif kind == "one_cell":  # everything inside a few centimetres
        tgt = [cloud(nt[c], 0.01) for c in range(6)]
        src = [cloud(ns[c], 0.01) for c in range(6)]
    elif kind == "far_origin":  # coordinates around 1e5 m: float cell arithmetic at its coarsest
        tgt = [cloud(nt[c], 5.0, 1e5) for c in range(6)]
        src = [cloud(ns[c], 5.0, 1e5) for c in range(6)]
    elif kind == "collinear":
        def line(n):
            t = rng.uniform(-20, 20, n)
            return abi.make_points(np.column_stack([t, 0.5 * t, 0 * t]), np.tile([0, 0, 1.0], (n, 1)), rng.uniform(0, 255, n))
        tgt = [line(nt[c]) for c in range(6)]
        src = [line(ns[c]) for c in range(6)]
    elif kind == "sparse_far":  # nothing within any search radius
        tgt = [cloud(nt[c], 2.0) for c in range(6)]
        src = [cloud(ns[c], 2.0, 500.0) for c in range(6)]
    else:  # "ragged": empty and tiny clouds mixed with normal ones
        sizes_t = rng.choice([0, 1, 2, 3, 50, 600], 6)
        sizes_s = rng.choice([0, 1, 2, 3, 40, 550], 6)
        tgt = [cloud(int(k), 4.0) if k else None for k in sizes_t]
        src = [cloud(int(k), 4.0) if k else None for k in sizes_s]
    return abi.PairData(tgt, src)


def noisy_copy(rng, nt, ns):
    """Source class clouds = samples of the target's, moved by a small rigid motion plus noise: registrations that run their iterations (hints,
    certificates, duplicate losers, the rejection chain) on class clouds of the given sizes."""
    T = synth.se3(*rng.normal(0, 0.1, 3), *np.deg2rad(rng.normal(0, 0.3, 3)))
    Ti = np.linalg.inv(T)
    tgt, src = [], []
    for c in range(6):
        xyz = rng.uniform(-25, 25, (nt[c], 3)) * np.array([1.0, 1.0, 0.15])
        nrm = rng.normal(0, 1, (nt[c], 3))
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        tgt.append(abi.make_points(xyz, nrm, rng.uniform(0, 255, nt[c])))
        k = rng.integers(0, nt[c], ns[c])
        sx = (xyz[k] + rng.normal(0, 0.03, (ns[c], 3))) @ Ti[:3, :3].T + Ti[:3, 3]
        src.append(abi.make_points(sx, nrm[k] @ Ti[:3, :3].T, rng.uniform(0, 255, ns[c])))
    return abi.PairData(tgt, src)


MIXED_KINDS = ("copy", "duplicates", "copy", "one_cell", "far_origin", "copy", "collinear", "sparse_far")
MIXED_T_SIZES = [(3, 64), (200, 900), (3000, 7088), (7089, 9000), (16000, 30000)]  # brute force, LDS tier (small / at its limit), global-memory tier
MIXED_S_SIZES = [(3, 64), (200, 700), (1400, 1600), (3900, 4300), (9000, 14000)]  # around 1536 (class-level job limit) and 4096 (LDS tier's source limit), chunked


def mixed_tier_pair(rng, kind):
    """A pair whose six class clouds draw their sizes from the ranges of all three search tiers (and sources from both sides of the class-level job
    limit), on the degenerate shapes above or as a noisy copy: one batch of them holds class-level and chunk-level jobs of both grid tiers."""
    nt = [int(rng.integers(*MIXED_T_SIZES[rng.integers(0, len(MIXED_T_SIZES))])) for _ in range(6)]
    ns = [int(rng.integers(*MIXED_S_SIZES[rng.integers(0, len(MIXED_S_SIZES))])) for _ in range(6)]
    return noisy_copy(rng, nt, ns) if kind == "copy" else degenerate_pair(rng, kind, nt, ns)


def mixed_tier_block(rng, per):
    """One option point and `per` mixed-tier pairs (tools/gpu_fuzz_mixed.py runs many blocks)."""
    P = random_params(rng)
    P.apply_motion_undistortion = 0
    P.normal_shooting_on = 0
    P.max_iter_num = int(rng.integers(2, 9))
    P.max_bearable_rotation_d = 45.0
    k0 = int(rng.integers(0, len(MIXED_KINDS)))
    return P, [mixed_tier_pair(rng, MIXED_KINDS[(k0 + i) % len(MIXED_KINDS)]) for i in range(per)]


@pytest.mark.parametrize("block", [0, 1, 2, 3])
def test_mixed_tier_degenerate_batches_match_oracle(ctx_auto, block):
    """Auto mode (a tier per class cloud) and the global-memory tier forced on every cloud, on batches whose class clouds span every search tier,
    pathological shapes included: each pair equals the oracle's.  (The LDS tier and the resident loop refuse clouds of that size when forced.)"""
    from mulls_amd import lib

    rng = np.random.default_rng(block * 15485863 + 11)
    P, pairs = mixed_tier_block(rng, 10)
    ro = [pyoracle.icp(pair, P)[0] for pair in pairs]
    for r, g in zip(ro, ctx_auto.icp_batch(pairs, P)):
        compare(r, g, check_trace=False, x_tol=1e-6)
    forced = lib.Context(0)
    try:
        forced.set_nn_mode(2)
        for r, g in zip(ro, forced.icp_batch(pairs, P)):
            compare(r, g, check_trace=False, x_tol=1e-6)
    finally:
        forced.close()


@pytest.mark.parametrize("kind", ["duplicates", "one_cell", "far_origin", "collinear", "sparse_far", "ragged"])
def test_degenerate_inputs_match_oracle(ctx, kind):
    rng = np.random.default_rng(zlib.crc32(kind.encode()))  # str hashes are salted per process: not a seed
    for k in range(10):
        pair = degenerate_pair(rng, kind)
        P = random_params(rng)
        P.apply_motion_undistortion = 0  # needs time stamps in the curvature field
        ro = pyoracle.icp(pair, P, trace_cap=32)[0]
        rg = ctx.icp(pair, P, trace_cap=32)[0]
        compare(ro, rg, x_tol=1e-6)  # these normal matrices are badly conditioned by construction


@pytest.mark.timeout(90, method="thread")
def test_non_finite_inputs_do_not_take_the_device_down(ctx, pairs_small):
    """NaN / inf / 1e38 coordinates are outside the contract (results are unspecified), but every kernel must stay inside
    its buffers and terminate: the grids are sized from the finite points only and saturate instead of overflowing."""
    base = pairs_small[0][0]
    rng = np.random.default_rng(5)

    def poison(clouds, vals, fields):
        out = []
        for c in clouds:
            c = c.copy()
            if len(c) > 10:
                for v in vals:
                    for f in fields:
                        c[f][rng.integers(0, len(c), 3)] = v
            out.append(c)
        return out

    cases = [
        abi.PairData(base.tgt, poison(base.src, [np.nan], "xyz")),
        abi.PairData(poison(base.tgt, [np.nan], "xyz"), base.src),
        abi.PairData(base.tgt, poison(base.src, [np.inf, -np.inf], "xyz")),
        abi.PairData(poison(base.tgt, [np.inf, -np.inf], "xyz"), base.src, tgt_bound=base.tgt_bound),
        abi.PairData(poison(base.tgt, [3e38, -3e38], "xyz"), base.src, tgt_bound=base.tgt_bound),
        abi.PairData(poison(base.tgt, [np.nan], ["nx", "ny", "nz"]), poison(base.src, [np.nan], ["nx", "ny", "nz"])),
    ]
    for pair in cases:
        for P in (abi.kitti_params(), abi.default_params(used_feature_type="111111", apply_intersection_filter=0),
                  abi.default_params(normal_shooting_on=1)):
            r = ctx.icp(pair, P)[0]
            rb = ctx.icp_batch([pair] * 12, P)
            assert r.code in (1, -1, -2, -3, 0) and all(x.code in (1, -1, -2, -3, 0) for x in rb)


EDGE_PARAMS = {
    "zero_iterations": dict(max_iter_num=0),
    "negative_iterations": dict(max_iter_num=-3),
    "many_iterations": dict(max_iter_num=300, converge_translation=0.0, converge_rotation_d=0.0),
    "zero_threshold": dict(dis_thre_unit=0.0, dis_thre_min=0.0),
    "min_above_unit": dict(dis_thre_unit=0.5, dis_thre_min=2.0),
    "growing_threshold": dict(dis_thre_update_rate=0.8, max_iter_num=12),
    "zero_windows": dict(pt2pt_residual_window=0.0, pt2pl_residual_window=0.0, pt2li_residual_window=0.0),
    "zero_balance": dict(z_xy_balanced_ratio=0.0),
    "negative_bearing": dict(normal_bearing=-10.0),
    "bearing_180": dict(normal_bearing=180.0, normal_shooting_on=1),
    "nothing_used": dict(used_feature_type="000000"),
    "only_vertex": dict(used_feature_type="000001"),
    "odd_flags": dict(used_feature_type="1x1 01", weight_strategy="2a01"),
    "zero_rate": dict(dis_thre_update_rate=0.0, max_iter_num=4),
    "huge_threshold": dict(dis_thre_unit=400.0, dis_thre_min=100.0, max_iter_num=3),
}


@pytest.mark.timeout(120, method="thread")
@pytest.mark.parametrize("name", sorted(EDGE_PARAMS))
def test_edge_parameter_values_match_oracle(ctx, pairs_small, name):
    """Parameter values nobody would configure but the reference accepts without a check: same outcome on the device."""
    P = abi.default_params(**EDGE_PARAMS[name])
    for pair, _ in pairs_small[:2]:
        ro = pyoracle.icp(pair, P, trace_cap=32)[0]
        rg = ctx.icp(pair, P, trace_cap=32)[0]
        compare(ro, rg, x_tol=1e-6)


@pytest.mark.parametrize("block", range(2))
def test_resident_batch_random_option_sequence(ctx_auto, pairs_small, block):
    """One staged batch of 700 pairs (class-level LDS jobs: on-chip duplicate rule, hints and cost classes kept from iteration to
    iteration, correspondence records) run with a sequence of random option points: whatever the previous run left in the scratch
    arenas (hints, records, flags, cost bits) must not leak — every run equals a fresh single-pair call bit for bit."""
    rng = np.random.default_rng(2600 + block)
    pairs = []
    for k in range(700):
        base, T_gt = pairs_small[int(rng.integers(0, len(pairs_small)))]
        pert = synth.se3(*rng.normal(0, 0.25, 3), *np.deg2rad(rng.normal(0, 0.6, 3)))
        pairs.append(abi.PairData(base.tgt, base.src, init_guess=pert @ T_gt, tgt_bound=base.tgt_bound))
    b = ctx_auto.batch(pairs)
    for trial in range(4):
        P = random_params(rng)
        rb = b.run(P)
        for i in rng.choice(len(pairs), 12, replace=False):
            r1 = ctx_auto.icp(pairs[int(i)], P)[0]
            assert (r1.code, r1.iters, list(r1.ncorr), r1.singular) == (rb[i].code, rb[i].iters, list(rb[i].ncorr), rb[i].singular), (trial, i)
            assert np.array_equal(np.array(r1.T[:]), np.array(rb[i].T[:]), equal_nan=True), (trial, i)
            assert np.array_equal(np.array(r1.info[:]), np.array(rb[i].info[:]), equal_nan=True), (trial, i)
    b.close()
