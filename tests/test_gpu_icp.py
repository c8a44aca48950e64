"""GPU parity of the whole path (mm_lls_icp) against the oracle through mulls_icp / mulls_icp_batch / mulls_batch_run.

Tolerances (BASELINE.json north_star): transform within 1e-4 m / 1e-4 rad.  Observed agreement is ~1e-12; the tests
pin 1e-7 so a real regression cannot hide inside the contractual tolerance.  Integer outputs (process code, iteration
count, per-iteration correspondence and live-source counts) must be identical."""
import numpy as np
import pytest

from conftest import planes_scene, transformed_copy
from mulls_amd import abi, synth
from oracle import pyoracle

pytestmark = pytest.mark.gpu

TOL_T, TOL_R = 1e-7, 1e-7


def compare(ro, rg, check_trace=True, x_tol=1e-9):
    assert ro.code == rg.code and ro.iters == rg.iters
    assert list(ro.ncorr) == list(rg.ncorr)
    assert list(ro.nsrc0) == list(rg.nsrc0) and list(ro.ntgt0) == list(rg.ntgt0)
    assert ro.singular == rg.singular
    assert ro.cropped == rg.cropped and list(ro.crop_box) == list(rg.crop_box)
    if np.isnan(ro.T_matrix()).any():
        # singular normal matrix: the reference lets inf/NaN propagate (SURVEY B-11); both must agree on where
        assert np.array_equal(np.isnan(ro.T_matrix()), np.isnan(rg.T_matrix()))
    else:
        dt, dr = synth.pose_error(rg.T_matrix(), ro.T_matrix())
        assert dt <= TOL_T and dr <= TOL_R, (dt, dr)
    # sigma^2 = VTPV / (n - 6) goes negative / infinite with fewer than seven observations: NaN and inf propagate in both
    assert (np.isnan(ro.sigma) and np.isnan(rg.sigma)) or ro.sigma == rg.sigma or abs(ro.sigma - rg.sigma) <= 1e-6 * max(1.0, abs(ro.sigma))
    assert ro.confidence == rg.confidence or (np.isnan(ro.confidence) and np.isnan(rg.confidence))
    io, ig = ro.info_matrix(), rg.info_matrix()
    assert np.array_equal(np.isfinite(io), np.isfinite(ig))
    if np.isfinite(io).all():
        assert np.abs(io - ig).max() <= 1e-6 * np.abs(io).max()
    if check_trace:
        assert ro.trace_len == rg.trace_len
        for k in range(ro.trace_len):
            a, b = ro.trace[k], rg.trace[k]
            assert list(a.ncorr) == list(b.ncorr) and list(a.nsrc) == list(b.nsrc), k
            assert list(a.thr) == list(b.thr)
            if any(a.atpa[:]):
                A, Bm = np.array(a.atpa[:]), np.array(b.atpa[:])
                assert np.array_equal(np.isfinite(A), np.isfinite(Bm))  # 0/0 weights etc. turn up in the same places
                fin = np.isfinite(A)
                if fin.any():
                    assert np.abs(A[fin] - Bm[fin]).max() <= 1e-10 * np.abs(A[fin]).max()
                if np.isfinite(np.array(a.x[:])).all():
                    # the solve amplifies the 1e-12 differences of the sums by the condition number of the normal matrix
                    assert np.abs(np.array(a.x[:]) - np.array(b.x[:])).max() <= x_tol * max(1.0, np.abs(np.array(a.x[:])).max())
                else:
                    assert not np.isfinite(np.array(b.x[:])).all()


PARAM_SETS = {
    "kitti_s2s": dict(base="kitti", dis_thre_unit=2.4),
    "kitti_fixed20": dict(base="kitti", converge_translation=0.0, converge_rotation_d=0.0),
    "defaults_5classes": dict(base="default"),
    "all6_nofilter": dict(base="default", used_feature_type="111111", apply_intersection_filter=0, weight_strategy="1111"),
    "unfaithful": dict(base="default", used_feature_type="111111", faithful=0),
    "equal_weights": dict(base="default", weight_strategy="0000", used_feature_type="111100"),
}


@pytest.mark.parametrize("name", sorted(PARAM_SETS))
def test_icp_matches_oracle(ctx, pairs_small, name):
    kw = dict(PARAM_SETS[name])
    base = kw.pop("base")
    P = abi.kitti_params(**kw) if base == "kitti" else abi.default_params(**kw)
    for pair, _ in pairs_small:
        ro = pyoracle.icp(pair, P, trace_cap=48)[0]
        rg = ctx.icp(pair, P, trace_cap=48)[0]
        compare(ro, rg)


def test_full_size_kitti_pair(ctx):
    """BASELINE config #2 at full size: 64-beam ~120k-point scans, source 800/400/1200, 20 iterations."""
    pair, T_gt = synth.make_pair(101)
    assert pair.n_raw[0] > 100000
    P = abi.kitti_params(converge_translation=0.0, converge_rotation_d=0.0)
    ro = pyoracle.icp(pair, P, trace_cap=24)[0]
    rg = ctx.icp(pair, P, trace_cap=24)[0]
    assert rg.iters == 20
    compare(ro, rg)
    dt, dr = synth.pose_error(rg.T_matrix(), T_gt)
    assert dt < 0.05 and dr < 2e-3


def test_batch_equals_single(ctx, pairs_small):
    P = abi.kitti_params(dis_thre_unit=2.4)
    pairs = [p for p, _ in pairs_small] * 3
    rb = ctx.icp_batch(pairs, P)
    for i, pr in enumerate(pairs):
        r1 = ctx.icp(pr, P)[0]
        assert r1.code == rb[i].code and r1.iters == rb[i].iters
        assert r1.T[:] == rb[i].T[:] and r1.info[:] == rb[i].info[:] and r1.sigma == rb[i].sigma  # bit-identical


def test_large_batch_two_sub_batches_in_flight(ctx, pairs_small):
    """n >= 2048: one launch set per iteration for the whole batch (device step), or two sub-batches pipelined on the stream when the host
    steps (driver.cpp): per-pair results stay bit-identical to the single-pair call, wherever a pair lands and whenever its neighbours
    converge or fail."""
    rng = np.random.default_rng(3)
    tgt = planes_scene(rng)
    far = abi.PairData(tgt, transformed_copy(tgt, synth.se3(200.0, 0, 0)))
    base = [p for p, _ in pairs_small] + [far]
    P = abi.kitti_params(dis_thre_unit=2.4)
    single = [ctx.icp(pr, P)[0] for pr in base]
    assert len({r.iters for r in single}) > 1  # sub-batches do not finish together
    order = [int(k) for k in rng.integers(0, len(base), 2100)]
    order[0], order[1049], order[1050], order[2099] = 3, 3, 3, 3
    for host_step in (0, 1):  # one launch set per iteration with the step on the device; two sub-batches with the host stepping them
        ctx.set_option(abi.OPT_HOST_STEP, host_step)
        rb = ctx.icp_batch([base[k] for k in order], P)
        ctx.set_option(abi.OPT_HOST_STEP, 0)
        for i, k in enumerate(order):
            r1 = single[k]
            assert (r1.code, r1.iters, list(r1.ncorr)) == (rb[i].code, rb[i].iters, list(rb[i].ncorr)), i
            assert r1.T[:] == rb[i].T[:] and r1.info[:] == rb[i].info[:] and r1.sigma == rb[i].sigma, i


def test_device_step_equals_host_step(ctx, pairs_small):
    """The lock-step loop steps on the device (k_finish_step: count test, 6x6 solve, convergence tests, residual — icp_step.h's functions) unless
    traces are asked for, which the host half of the loop collects (or MULLS_OPT_HOST_STEP): every output bit-identical between the two, for
    healthy pairs, failing ones (-1, -2, -3), loops of one to three iterations, and so are the profile's point counters."""
    rng = np.random.default_rng(11)
    tgt = planes_scene(rng)
    far = abi.PairData(tgt, transformed_copy(tgt, synth.se3(200.0, 0, 0)))
    empty = abi.PairData(tgt, [None] * 6)
    turned = abi.PairData(tgt, transformed_copy(tgt, np.linalg.inv(synth.se3(0.05, 0, 0, 0, 0, 0.03))))
    plist = [p for p, _ in pairs_small] + [far, empty, turned] + [p for p, _ in pairs_small]
    sets = [abi.kitti_params(dis_thre_unit=2.4), abi.default_params(), abi.default_params(used_feature_type="111000", max_bearable_rotation_d=0.1),
            abi.default_params(used_feature_type="111000", sigma_thre=1e-9), abi.kitti_params(dis_thre_unit=2.4, max_iter_num=1),
            abi.kitti_params(dis_thre_unit=2.4, max_iter_num=2), abi.kitti_params(dis_thre_unit=2.4, max_iter_num=3),
            abi.kitti_params(dis_thre_unit=2.4, apply_motion_undistortion=1), abi.kitti_params(dis_thre_unit=2.4, apply_intersection_filter=0)]
    b = ctx.batch(plist)
    codes = set()
    for P in sets:
        out, prof = {}, {}
        for host in (0, 1, 0):
            ctx.set_option(abi.OPT_HOST_STEP, host)
            r = b.run(P)
            got = [(x.code, x.iters, tuple(x.ncorr), tuple(x.nsrc0), tuple(x.ntgt0), x.cropped, tuple(x.crop_box), np.array(x.T[:]).tobytes(),
                    np.array(x.info[:]).tobytes(), np.float32(x.sigma).tobytes(), np.float32(x.confidence).tobytes(), x.singular) for x in r]  # (NaN-safe: bytes)
            pf = ctx.profile()
            cnt = (pf.nn_src_pts, pf.nn_tgt_pts, pf.nn_tgt_unique, pf.nn_pair_evals, pf.iterations)
            if host in out:
                assert out[host] == got and prof[host] == cnt  # and run after run
            out[host], prof[host] = got, cnt
        assert out[0] == out[1]
        assert prof[0] == prof[1]
        codes |= {x[0] for x in out[0]}
    ctx.set_option(abi.OPT_HOST_STEP, 0)
    assert codes >= {1, -1, -2, -3}
    b.close()


def test_lock_step_launch_forms_are_bit_identical(ctx, pairs_small):
    """The lock-step iteration exists in two launch forms, picked by batch size: four launches (one accumulation launch for every trip length;
    k_finish_step = finish + step + publication behind an arrival ticket) for small batches, seven for large ones.  Same bits from both, healthy
    and failing pairs, all six classes (every metric's term set), the residual pass, run after run."""
    rng = np.random.default_rng(5)
    tgt = planes_scene(rng)
    far = abi.PairData(tgt, transformed_copy(tgt, synth.se3(200.0, 0, 0)))
    empty = abi.PairData(tgt, [None] * 6)
    plist = ([p for p, _ in pairs_small] + [far, empty]) * 7
    wave_min = ctx.get_option(abi.OPT_ACCUM_WAVE_MIN_TRIPS)
    ctx.set_option(abi.OPT_ACCUM_WAVE_MIN_TRIPS, 1)
    b = ctx.batch(plist)
    # ... and batches of SPLIT_MIN .. SPLIT_MAX pairs iterate as two sub-batches on two streams (own launch sets, epoch word, ticket, work list)
    # ... and (round 5) the normal equations summed by one wave per trip (k_accum_wave: ACCUM_WAVE_MIN_TRIPS = 1) instead of one workgroup per trip (0)
    # ... and (round 6) one wave per pair that sums the trip partials and steps the pair (k_sum_step: SUM_STEP = 1) instead of k_finish followed by k_step (0)
    forms = {"few": (384, 1 << 30, 0, 1), "separate": (0, 1 << 30, 0, 1), "few split": (384, 2, 0, 1), "separate split": (0, 2, 0, 1), "few wave": (384, 1 << 30, 1, 1),
             "separate wave": (0, 1 << 30, 1, 1), "separate split wave": (0, 2, 1, 1),
             # (STEP_LAUNCH_MAX_PAIRS = 0: finish, step and publication as the separate launches of large batches)
             "large": (0, 1 << 30, 0, 1, 0), "large two kernels": (0, 1 << 30, 0, 0, 0), "large split wave": (0, 2, 1, 1, 0), "large split wave two kernels": (0, 2, 1, 0, 0)}
    split_min = ctx.get_option(abi.OPT_SPLIT_MIN_PAIRS)
    step_max = ctx.get_option(abi.OPT_STEP_LAUNCH_MAX_PAIRS)
    few_max = ctx.get_option(abi.OPT_FEW_LAUNCHES_MAX_PAIRS)
    for P in (abi.kitti_params(dis_thre_unit=2.4), abi.default_params(), abi.default_params(used_feature_type="111111", faithful=0),
              abi.kitti_params(dis_thre_unit=2.4, max_iter_num=1)):
        got = {}
        for name in list(forms) * 2:
            ctx.set_option(abi.OPT_FEW_LAUNCHES_MAX_PAIRS, forms[name][0])
            ctx.set_option(abi.OPT_SPLIT_MIN_PAIRS, forms[name][1])
            ctx.set_option(abi.OPT_ACCUM_WAVE_MIN_TRIPS, forms[name][2])
            ctx.set_option(abi.OPT_SUM_STEP, forms[name][3])
            ctx.set_option(abi.OPT_STEP_LAUNCH_MAX_PAIRS, forms[name][4] if len(forms[name]) > 4 else step_max)
            r = b.run(P)
            rows = [(x.code, x.iters, tuple(x.ncorr), tuple(x.nsrc0), np.array(x.T[:]).tobytes(), np.array(x.info[:]).tobytes(), np.float32(x.sigma).tobytes()) for x in r]
            assert got.setdefault(name, rows) == rows
        assert all(got[name] == got["few"] for name in forms)
    ctx.set_option(abi.OPT_FEW_LAUNCHES_MAX_PAIRS, few_max)
    ctx.set_option(abi.OPT_SPLIT_MIN_PAIRS, split_min)
    ctx.set_option(abi.OPT_ACCUM_WAVE_MIN_TRIPS, wave_min)
    ctx.set_option(abi.OPT_SUM_STEP, 1)
    ctx.set_option(abi.OPT_STEP_LAUNCH_MAX_PAIRS, step_max)
    b.close()


def test_k_candidate_certificates_are_bit_identical(ctx, pairs_small):
    """Round 5: a point whose hinted target fails the certificate evaluates the few nearest targets its last search's other lanes saw (k-candidate
    certificates, MULLS_OPT_KCERT) before it is searched again.  Certificates never change a result: on, off, and taken from list lengths of 1 and 64
    on, every output is the same bits — healthy, failing and empty pairs, every class, run after run on the same batch (records of an earlier run carry
    an older epoch and must not be believed), and with the duplicate-table epoch next to its wrap (the records are cleared with the table)."""
    rng = np.random.default_rng(23)
    tgt = planes_scene(rng)
    far = abi.PairData(tgt, transformed_copy(tgt, synth.se3(200.0, 0, 0)))
    empty = abi.PairData(tgt, [None] * 6)
    plist = ([p for p, _ in pairs_small] + [far, empty]) * 5
    keep = (ctx.get_option(abi.OPT_KCERT), ctx.get_option(abi.OPT_KCERT_MIN), ctx.get_option(abi.OPT_FEW_LAUNCHES_MAX_PAIRS))
    ctx.set_option(abi.OPT_FEW_LAUNCHES_MAX_PAIRS, 0)  # k_cert + k_nn_lds as separate launches: the 512-entry leftover lists that take the look
    b = ctx.batch(plist)
    for P in (abi.kitti_params(dis_thre_unit=2.4), abi.default_params(), abi.default_params(used_feature_type="111111", faithful=0)):
        got = None
        for kc, kmin in ((0, 64), (1, 1), (1, 64), (0, 1), (1, 1), (1, 512)):
            ctx.set_option(abi.OPT_KCERT, kc)
            ctx.set_option(abi.OPT_KCERT_MIN, kmin)
            r = b.run(P)
            rows = [(x.code, x.iters, tuple(x.ncorr), tuple(x.nsrc0), np.array(x.T[:]).tobytes(), np.array(x.info[:]).tobytes(), np.float32(x.sigma).tobytes()) for x in r]
            if got is None:
                got = rows
            assert rows == got, (kc, kmin)
    b.close()
    # the epoch counter of a fresh batch next to its wrap: the fourth run crosses it (the candidate records are cleared with the duplicate table)
    ctx.set_option(abi.OPT_KCERT, 1)
    ctx.set_option(abi.OPT_KCERT_MIN, 1)
    ctx.set_option(abi.OPT_DEBUG_TICK, 0xfffffff0 - 3 * 22 - 5)
    b = ctx.batch(plist)
    P = abi.default_params(used_feature_type="111111", faithful=0)
    for _ in range(6):
        rows = [(x.code, x.iters, tuple(x.ncorr), tuple(x.nsrc0), np.array(x.T[:]).tobytes(), np.array(x.info[:]).tobytes(), np.float32(x.sigma).tobytes()) for x in b.run(P)]
        assert rows == got
    b.close()
    ctx.set_option(abi.OPT_DEBUG_TICK, 0)
    ctx.set_option(abi.OPT_KCERT, keep[0])
    ctx.set_option(abi.OPT_KCERT_MIN, keep[1])
    ctx.set_option(abi.OPT_FEW_LAUNCHES_MAX_PAIRS, keep[2])


def test_fused_target_setup_is_bit_identical(ctx, pairs_small):
    """LDS tier: k_tgt_grid (crop + grid build of a target class cloud in one pass, no cropped copy: the records a correspondence needs are gathered
    from the staged cloud through the crop's map) against k_crop + k_grid_build_sort: same bits, with and without the intersection filter, healthy
    and failing pairs, lock-step and resident loop (the ctx fixture's tiers), single calls and batches."""
    rng = np.random.default_rng(17)
    tgt = planes_scene(rng)
    far = abi.PairData(tgt, transformed_copy(tgt, synth.se3(200.0, 0, 0)))
    empty_t = abi.PairData([None] * 6, transformed_copy(tgt, synth.se3(0.1, 0, 0)))
    plist = ([p for p, _ in pairs_small] + [far, empty_t]) * 3
    b = ctx.batch(plist)
    for P in (abi.kitti_params(dis_thre_unit=2.4), abi.default_params(), abi.default_params(used_feature_type="111111", apply_intersection_filter=0),
              abi.kitti_params(dis_thre_unit=2.4, apply_motion_undistortion=1)):
        got = {}
        for fused in (1, 0, 1):
            ctx.set_option(abi.OPT_FUSED_TGT_SETUP, fused)
            r = list(b.run(P)) + [ctx.icp(plist[0], P)[0]]
            rows = [(x.code, x.iters, tuple(x.ncorr), tuple(x.nsrc0), tuple(x.ntgt0), x.cropped, tuple(x.crop_box), np.array(x.T[:]).tobytes(),
                     np.array(x.info[:]).tobytes(), np.float32(x.sigma).tobytes()) for x in r]
            assert got.setdefault(fused, rows) == rows
        assert got[0] == got[1]
    ctx.set_option(abi.OPT_FUSED_TGT_SETUP, 1)
    b.close()


def test_one_pass_class_walk_is_bit_identical(ctx, pairs_small):
    """k_cert takes a whole class cloud that fits its lanes' registers through certificates, duplicate rule and rejection chain in one pass
    (cert_class_flat); MULLS_OPT_DEBUG_STOP = 9 sends every class cloud through the general three-walk form instead.  Same bits: healthy and
    failing pairs, every class, gate on and off (clouds above and below 500 live points), the iteration where nothing matches."""
    rng = np.random.default_rng(23)
    tgt = planes_scene(rng)
    far = abi.PairData(tgt, transformed_copy(tgt, synth.se3(200.0, 0, 0)))
    near_far = abi.PairData(tgt, transformed_copy(tgt, synth.se3(3.0, 0.5, 0, 0, 0, 0.05)))
    plist = ([p for p, _ in pairs_small] + [far, near_far]) * 5
    b = ctx.batch(plist)
    for P in (abi.kitti_params(dis_thre_unit=2.4), abi.default_params(), abi.default_params(used_feature_type="111111", faithful=0, rejector_strict=0),
              abi.kitti_params(dis_thre_unit=0.4, max_iter_num=6)):
        got = {}
        for stop in (0, 9, 10, 0):  # 10: light and heavy pass as two launches (k_cert + k_nn_lds) where a small batch runs them as one (k_cert_nn)
            ctx.set_option(abi.OPT_DEBUG_STOP, stop)
            r = list(b.run(P)) + [ctx.icp(plist[1], P)[0]]
            rows = [(x.code, x.iters, tuple(x.ncorr), tuple(x.nsrc0), np.array(x.T[:]).tobytes(), np.array(x.info[:]).tobytes(), np.float32(x.sigma).tobytes()) for x in r]
            assert got.setdefault(stop, rows) == rows
        assert got[0] == got[9] == got[10]
    ctx.set_option(abi.OPT_DEBUG_STOP, 0)
    b.close()


def test_resident_batch_is_repeatable(ctx, pairs_small):
    """mulls_batch_run re-clones the staged clouds every run: identical results run after run, run-to-run deterministic."""
    P = abi.kitti_params(dis_thre_unit=2.4)
    b = ctx.batch([p for p, _ in pairs_small])
    r1 = b.run(P)
    first = [(r.code, r.iters, tuple(r.T), tuple(r.info), r.sigma) for r in r1]
    for _ in range(3):
        r2 = b.run(P)
        assert [(r.code, r.iters, tuple(r.T), tuple(r.info), r.sigma) for r in r2] == first
    P2 = abi.default_params()  # different class set on the same staged batch
    r3 = b.run(P2)
    ro = pyoracle.icp(pairs_small[0][0], P2)[0]
    compare(ro, r3[0], check_trace=False)
    b.close()


def test_profiling_levels_do_not_change_results(ctx, pairs_small):
    """mulls_set_profiling: 0 off, 1 events around every launch group, 2 around the correspondence search only — same results, and the
    profile says what was bracketed."""
    P = abi.kitti_params(dis_thre_unit=2.4)
    b = ctx.batch([p for p, _ in pairs_small])
    out = {}
    for level in (0, 2, 1):
        ctx.set_profiling(level)
        out[level] = [(r.code, r.iters, tuple(r.T), tuple(r.info), r.sigma) for r in b.run(P)]
        pf = ctx.profile()
        if level:
            assert pf.ms_nn > 0 and pf.launches_nn > 0
        if level == 2:
            assert pf.ms_accum == 0 and pf.ms_setup == 0
    ctx.set_profiling(0)
    assert out[0] == out[1] == out[2]
    b.close()


def test_resident_loop_equals_lock_step(ctx_auto, pairs_small):
    """The device-resident loop (one launch, 6x6 solve on the device) and the lock-step path (host-stepped launches) run the same
    arithmetic in the same order: every output bit-identical, trace included."""
    from mulls_amd import lib

    plist = [p for p, _ in pairs_small] * 4  # 12 pairs: more than the 8 below which auto mode keeps the global-memory tier
    for P in (abi.kitti_params(dis_thre_unit=2.4), abi.default_params(), abi.kitti_params(dis_thre_unit=2.4, faithful=0, weight_strategy="1011")):
        out = {}
        for mode in (4, 3, 2):
            c = lib.Context(0)
            c.set_nn_mode(mode)
            r = c.icp_batch(plist, P, trace_cap=24)
            out[mode] = [(x.code, x.iters, tuple(x.ncorr), tuple(x.nsrc0), tuple(x.ntgt0), tuple(x.T), tuple(x.info), x.sigma, x.confidence, x.trace_len,
                          [(t.iter, tuple(t.ncorr), tuple(t.nsrc), tuple(t.thr), tuple(t.atpa), tuple(t.atpb), tuple(t.x)) for t in x.trace[:x.trace_len]])
                         for x in r]
            c.close()
        assert out[4] == out[3] == out[2]
        assert any(x[0] == 1 for x in out[4])


def test_duplicate_table_epoch_wrap(ctx, pairs_small):
    """The duplicate table's 32-bit epoch counter: runs on either side of the wrap give the results of a fresh batch
    (stale winner entries of older epochs must not beat the keys of the restarted count)."""
    P = abi.kitti_params(dis_thre_unit=2.4)
    plist = [p for p, _ in pairs_small]
    b0 = ctx.batch(plist)
    want = [(r.code, r.iters, tuple(r.ncorr), tuple(r.T)) for r in b0.run(P)]
    b0.close()
    ctx.set_option(abi.OPT_DEBUG_TICK, 0xfffffff0 - 3 * 22 - 5)  # the fourth run of 22 epochs crosses the limit
    b = ctx.batch(plist)
    for _ in range(6):
        assert [(r.code, r.iters, tuple(r.ncorr), tuple(r.T)) for r in b.run(P)] == want
    b.close()
    ctx.set_option(abi.OPT_DEBUG_TICK, 0)


def test_mixed_outcomes_in_one_batch(ctx):
    """Pairs that fail early (-2), step too far (-1), exceed sigma (-3) or never iterate sit next to healthy ones."""
    rng = np.random.default_rng(7)
    tgt = planes_scene(rng)
    good = abi.PairData(tgt, transformed_copy(tgt, np.linalg.inv(synth.se3(0.1, 0.05, 0.0, 0, 0, 0.01))))
    far = abi.PairData(tgt, transformed_copy(tgt, synth.se3(200.0, 0, 0)))
    empty = abi.PairData(tgt, [None] * 6)
    P = abi.default_params(used_feature_type="111000", apply_intersection_filter=0)
    pairs = [good, far, good, empty, far]
    rb = ctx.icp_batch(pairs, P, trace_cap=24)
    for i, pr in enumerate(pairs):
        ro = pyoracle.icp(pr, P, trace_cap=24)[0]
        compare(ro, rb[i])
    assert [r.code for r in rb] == [1, -2, 1, -2, -2]
    for kw, code in ((dict(max_bearable_rotation_d=0.1), -1), (dict(sigma_thre=1e-9), -3), (dict(max_iter_num=0), 0)):
        P2 = abi.default_params(used_feature_type="111000", **kw)
        pr = abi.PairData(tgt, transformed_copy(tgt, np.linalg.inv(synth.se3(0.05, 0, 0, 0, 0, 0.03))))
        ro, rg = pyoracle.icp(pr, P2)[0], ctx.icp(pr, P2)[0]
        assert ro.code == code
        compare(ro, rg, check_trace=False)


def test_analytic_recovery_on_gpu(ctx):
    rng = np.random.default_rng(4)
    tgt = planes_scene(rng)
    T_true = synth.se3(0.20, -0.15, 0.10, 0.01, -0.008, 0.02)
    pair = abi.PairData(tgt, transformed_copy(tgt, np.linalg.inv(T_true)))
    P = abi.default_params(used_feature_type="111000", weight_strategy="1000", converge_translation=1e-7, converge_rotation_d=1e-6,
                           max_iter_num=30)
    r = ctx.icp(pair, P)[0]
    dt, dr = synth.pose_error(r.T_matrix(), T_true)
    assert r.code == 1 and dt < 2e-5 and dr < 2e-6


def test_stale_correspondences_and_tiny_clouds(ctx):
    """SURVEY B-4: classes with < 3 points are skipped (their correspondence list stays empty), classes that lose all
    matches keep the stale list; both implementations must agree on counts every iteration."""
    rng = np.random.default_rng(9)
    tgt = planes_scene(rng, n_per=300)
    src = transformed_copy(tgt, np.linalg.inv(synth.se3(0.3, 0.1, 0.0, 0, 0, 0.01)))
    src[abi.PILLAR] = src[abi.PILLAR][:2]  # below K_min
    tgt2 = list(tgt)
    tgt2[abi.GROUND] = tgt2[abi.GROUND][:2]
    P = abi.default_params(used_feature_type="111000", dis_thre_unit=1.0, dis_thre_min=0.05, dis_thre_update_rate=2.0, max_iter_num=8,
                           min_neccessary_corr_ratio=0.0, apply_intersection_filter=0)
    for pr in (abi.PairData(tgt, src), abi.PairData(tgt2, src)):
        ro = pyoracle.icp(pr, P, trace_cap=16)[0]
        rg = ctx.icp(pr, P, trace_cap=16)[0]
        compare(ro, rg)


@pytest.mark.parametrize("used", ["111110", "111111", "111000"])
def test_motion_undistortion(ctx, pairs_small, used):
    """apply_motion_undistortion_while_registration (cregistration.hpp:1248-1258, cfilter.hpp:493-549): the five
    non-vertex source clouds are regenerated from block2->pc_*_down with per-point slerp, the intersection filter is
    skipped, and the vertex cloud receives the initial guess twice (reference quirk)."""
    P = abi.default_params(apply_motion_undistortion=1, used_feature_type=used)
    for pair, _ in pairs_small:
        assert pair.src[0]["curvature"].max() > 0.5  # time stamps present
        ro = pyoracle.icp(pair, P, trace_cap=32)[0]
        rg = ctx.icp(pair, P, trace_cap=32)[0]
        compare(ro, rg)
    # use_more_points: src = un-down-sampled clouds, src_down = the down-sampled ones the regeneration starts from
    pair, _ = pairs_small[0]
    rng = np.random.default_rng(0)
    down = [c[np.sort(rng.choice(len(c), size=max(len(c) // 2, min(len(c), 3)), replace=False))] if len(c) else c for c in pair.src]
    pr = abi.PairData(pair.tgt, pair.src, init_guess=pair.init_guess, tgt_bound=pair.tgt_bound, src_down=down)
    P = abi.default_params(apply_motion_undistortion=1, use_more_points=1)
    ro = pyoracle.icp(pr, P, trace_cap=32)[0]
    rg = ctx.icp(pr, P, trace_cap=32)[0]
    compare(ro, rg)
    assert list(rg.nsrc0) == [len(c) for c in pair.src]


@pytest.mark.parametrize("seed", [0, 7, 123456789])
def test_keep_less_source_points(ctx, pairs_small, seed):
    """keep_less_source_pts (cregistration.hpp:2866-2892) as used by the map-to-map registrations (test/mulls_slam.cpp:477-482):
    halves target ground/facade, caps every source class relative to its target.  Seeded selection sampling, same
    definition in oracle and device."""
    P = abi.default_params(keep_less_source_points=1, use_more_points=1, rng_seed=seed, max_iter_num=6)
    for pair, _ in pairs_small[:2]:
        # map-to-map shape: source as dense as the target
        pr = abi.PairData(pair.tgt, [pyoracle.transform(t, np.linalg.inv(pair.init_guess)) for t in pair.tgt], init_guess=pair.init_guess,
                          tgt_bound=pair.tgt_bound)
        ro = pyoracle.icp(pr, P, trace_cap=16)[0]
        rg = ctx.icp(pr, P, trace_cap=16)[0]
        compare(ro, rg)
        assert rg.ntgt0[abi.GROUND] <= (len(pair.tgt[abi.GROUND]) + 1) // 2 and rg.nsrc0[abi.GROUND] <= rg.ntgt0[abi.GROUND] // 4 + 1


@pytest.mark.parametrize("used", ["111110", "101010"])
def test_normal_shooting(ctx, pairs_small, used):
    """normal_shooting_on: planar classes use PCL's CorrespondenceEstimationNormalShooting (k = 10)."""
    P = abi.default_params(normal_shooting_on=1, used_feature_type=used, min_neccessary_corr_ratio=0.0 if used == "101010" else 0.03)
    for pair, _ in pairs_small:
        ro = pyoracle.icp(pair, P, trace_cap=32)[0]
        rg = ctx.icp(pair, P, trace_cap=32)[0]
        compare(ro, rg)


def test_normal_shooting_survives_nan_queries(ctx, pairs_small):
    """A class set that leaves the normal matrix singular (facades only) makes the reference propagate inf / NaN into the
    transform (SURVEY B-11); the next iteration's normal-shooting search then runs on NaN queries.  The 10-NN list stays
    empty there (as in the oracle's kd-tree) — it used to be read as if it were full."""
    P = abi.default_params(max_iter_num=6, dis_thre_unit=1.95, used_feature_type="001000", weight_strategy="0010", normal_shooting_on=1,
                           normal_bearing=33.0, min_neccessary_corr_ratio=0.0)
    for pair, _ in pairs_small:
        ro = pyoracle.icp(pair, P, trace_cap=16)[0]
        rg = ctx.icp(pair, P, trace_cap=16)[0]
        compare(ro, rg)
    rb = ctx.icp_batch([p for p, _ in pairs_small] * 40, P)
    assert [r.code for r in rb[:3]] == [ctx.icp(p, P)[0].code for p, _ in pairs_small]


def test_option_values_are_validated(ctx):
    """mulls_set_option refuses what a cast to a count or a size could not take (non-finite, negative, absurd) and keeps the old value; the array stagger is
    rounded down to a multiple of 256 bytes (the per-point arrays hold 16-byte records)."""
    from mulls_amd import lib

    before = ctx.get_option(abi.OPT_STAGGER)
    for bad in (float("nan"), float("inf"), -1.0, 1e18):
        for opt in (abi.OPT_STAGGER, abi.OPT_SPLIT_MIN_PAIRS, abi.OPT_CERT_SLACK_MIN, abi.OPT_GRID_H0):
            with pytest.raises(lib.MullsError):
                ctx.set_option(opt, bad)
    assert ctx.get_option(abi.OPT_STAGGER) == before
    ctx.set_option(abi.OPT_STAGGER, 4400)
    assert ctx.get_option(abi.OPT_STAGGER) == 4352
    ctx.set_option(abi.OPT_STAGGER, before)
    with pytest.raises(lib.MullsError):
        ctx.set_option(abi.OPT_COUNT, 1)
