"""Batches whose class clouds sit on different search tiers (auto mode since round 4: CloudDesc::tier per (pair, class)).

A mixed batch must return, pair by pair, the bits a batch on one tier returns — whatever its neighbours in the batch are — and the oracle's integer outputs:
the LDS tier's class-level jobs (k_cert / k_nn_lds), the global-memory tier's class-level jobs (a down-sampled scan against a target above the LDS tier's
size: k_cert_big resolves the duplicate rule and the rejection chain itself) and its chunk-level jobs (+ k_filter) all run inside one launch set."""
import numpy as np
import pytest

from mulls_amd import abi, synth
from oracle import pyoracle

pytestmark = pytest.mark.gpu


def same_bits(a, b):
    return (a.code, a.iters, list(a.ncorr), list(a.nsrc0), list(a.ntgt0)) == (b.code, b.iters, list(b.ncorr), list(b.nsrc0), list(b.ntgt0)) and list(a.T[:]) == list(b.T[:]) and \
        list(a.info[:]) == list(b.info[:]) and a.sigma == b.sigma


def oracle_equal(ro, rg):
    assert (ro.code, ro.iters, list(ro.ncorr), list(ro.nsrc0), list(ro.ntgt0)) == (rg.code, rg.iters, list(rg.ncorr), list(rg.nsrc0), list(rg.ntgt0))
    dt, dr = synth.pose_error(rg.T_matrix(), ro.T_matrix())
    assert dt <= 1e-7 and dr <= 1e-7, (dt, dr)


@pytest.fixture(scope="module")
def zoo():
    """pairs of every shape: small everywhere (LDS tier), an 11.5 k / 8 k-point ground or facade target (global-memory tier, class-level jobs, next to LDS-tier
    clouds of the same pair), a 3 000-point source against a big target (chunk-level jobs), a source above 4 096 points on a small target, an empty class"""
    out = []
    small_src = {abi.GROUND: 800, abi.PILLAR: 400, abi.FACADE: 1200, abi.BEAM: 200, abi.ROOF: 100}
    shapes = [
        (small_src, {abi.GROUND: 5000, abi.PILLAR: 1500, abi.FACADE: 6000, abi.BEAM: 600, abi.ROOF: 400}),
        (small_src, {abi.GROUND: 11500, abi.PILLAR: 1500, abi.FACADE: 5500, abi.BEAM: 900, abi.ROOF: 600}),
        (small_src, {abi.GROUND: 7089, abi.PILLAR: 7088, abi.FACADE: 9728, abi.BEAM: 0, abi.ROOF: 600}),
        ({abi.GROUND: 3000, abi.PILLAR: 600, abi.FACADE: 1537, abi.BEAM: 200, abi.ROOF: 100}, {abi.GROUND: 20000, abi.PILLAR: 2500, abi.FACADE: 12000, abi.BEAM: 900, abi.ROOF: 600}),
        ({abi.GROUND: 5000, abi.PILLAR: 400, abi.FACADE: 1200, abi.BEAM: 200, abi.ROOF: 100}, {abi.GROUND: 4000, abi.PILLAR: 1500, abi.FACADE: 6000, abi.BEAM: 600, abi.ROOF: 400}),
    ]
    for k, (src, tgt) in enumerate(shapes):
        pair, T_gt = synth.make_pair(700 + k, n_beams=128, n_az=1875, elev_deg=(-25.0, 15.0), src_counts=src, tgt_counts=tgt, vertex_count=300)
        out.append((pair, T_gt))
    return out


PARAMS = {
    "kitti": lambda: abi.kitti_params(),
    "all6_fixed12": lambda: abi.default_params(used_feature_type="111111", weight_strategy="1111", max_iter_num=12, converge_translation=0.0, converge_rotation_d=0.0, dis_thre_unit=1.4),
    "nofilter": lambda: abi.default_params(used_feature_type="101100", apply_intersection_filter=0, max_iter_num=8),
}


@pytest.mark.parametrize("pname", sorted(PARAMS))
def test_mixed_batch_equals_single_tier_batches_and_the_oracle(ctx_auto, zoo, pname):
    from mulls_amd import lib

    P = PARAMS[pname]()
    rng = np.random.default_rng(5)
    pairs = []
    for k in range(40):  # every shape several times with its own guess, shuffled: a pair's neighbours in the batch are of other tiers
        base, T_gt = zoo[k % len(zoo)]
        pert = synth.se3(*rng.normal(0, 0.15, 3), *np.deg2rad(rng.normal(0, 0.3, 3)))
        pairs.append(abi.PairData(base.tgt, base.src, init_guess=pert @ T_gt, tgt_bound=base.tgt_bound))
    order = rng.permutation(len(pairs))
    pairs = [pairs[i] for i in order]
    mixed = ctx_auto.icp_batch(pairs, P)
    # the same pairs with one tier for the whole batch (rounds 1 - 3: the global-memory grid, every cloud as chunk-level jobs), and with the first-iterations
    # form of the mixed batch in every iteration / in none
    c = lib.Context(0)
    try:
        c.set_nn_mode(2)
        one_tier = c.icp_batch(pairs, P)
        c.set_nn_mode(0)
        c.set_option(abi.OPT_BIG_EARLY_SETS, 1000)
        early = c.icp_batch(pairs, P)
        c.set_option(abi.OPT_BIG_EARLY_SETS, 0)
        late = c.icp_batch(pairs + pairs, P)  # (80 pairs: enough class-level jobs for the class-level form from iteration 0 on)
        c.set_option(abi.OPT_MIXED_TIERS, 0)
        unmixed = c.icp_batch(pairs, P)
    finally:
        c.close()
    for i in range(len(pairs)):
        assert same_bits(mixed[i], one_tier[i]), i
        assert same_bits(mixed[i], early[i]) and same_bits(mixed[i], late[i]) and same_bits(mixed[i], late[len(pairs) + i]) and same_bits(mixed[i], unmixed[i]), i
    for i in (0, 7, 13, 21, 39):
        oracle_equal(pyoracle.icp(pairs[i], P)[0], mixed[i])
        assert same_bits(ctx_auto.icp(pairs[i], P)[0], mixed[i])  # ... and alone


def test_mixed_batch_with_traces_and_device_step(ctx_auto, zoo):
    """host-stepped (traces) and device-stepped loops launch the same mixed sets: same results, traces equal to the oracle's"""
    P = abi.kitti_params(dis_thre_unit=2.0)
    pairs = [z[0] for z in zoo]
    dev = ctx_auto.icp_batch(pairs, P)
    host = ctx_auto.icp_batch(pairs, P, trace_cap=24)
    for i, p in enumerate(pairs):
        assert same_bits(dev[i], host[i])
        ro = pyoracle.icp(p, P, trace_cap=24)[0]
        oracle_equal(ro, host[i])
        assert ro.trace_len == host[i].trace_len
        for k in range(ro.trace_len):
            assert list(ro.trace[k].ncorr) == list(host[i].trace[k].ncorr) and list(ro.trace[k].nsrc) == list(host[i].trace[k].nsrc), (i, k)


def test_small_mixed_batch_one_launch_for_both_tiers(zoo):
    """A small mixed batch runs the class clouds of both tiers in ONE launch (k_cert_mixed); MULLS_OPT_DEBUG_STOP = 22 launches the tiers one after the other
    as large batches do.  Same bits either way, equal to the oracle."""
    from mulls_amd import lib

    P = abi.kitti_params()
    c = lib.Context(0)
    try:
        for n in (1, 3):
            pairs = [z[0] for z in zoo[1 : 1 + n]] if n > 1 else [zoo[1][0]]
            together = c.icp_batch(pairs, P)
            c.set_option(abi.OPT_DEBUG_STOP, 22)
            apart = c.icp_batch(pairs, P)
            c.set_option(abi.OPT_DEBUG_STOP, 0)
            for i, p in enumerate(pairs):
                assert same_bits(together[i], apart[i]), (n, i)
                oracle_equal(pyoracle.icp(p, P)[0], together[i])
    finally:
        c.close()
