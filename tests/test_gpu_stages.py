"""GPU parity, stage by stage, through the C ABI (mulls_stage_* run the same kernels the ICP driver launches).
Integer / index / flag outputs must be bit-exact against the oracle; double accumulators agree to 1e-12 relative
(the device sums in a fixed tree order, the reference serially)."""
import numpy as np
import pytest

from mulls_amd import abi, synth
from oracle import pyoracle

pytestmark = pytest.mark.gpu


def test_transform_bit_exact(ctx):
    rng = np.random.default_rng(0)
    for n in (1, 63, 64, 65, 1000, 4097):
        pts = abi.make_points(rng.uniform(-80, 80, (n, 3)), rng.normal(size=(n, 3)), rng.uniform(0, 255, n), rng.uniform(0, 1, n))
        T = synth.se3(*rng.normal(0, 1, 3), *rng.normal(0, 0.1, 3))
        a = ctx.transform(pts, T)
        b = pyoracle.transform(pts, T)
        for k in abi.POINT_DTYPE.names:
            assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), (n, k)


@pytest.mark.parametrize("cls", [abi.GROUND, abi.PILLAR, abi.FACADE, abi.VERTEX])
@pytest.mark.parametrize("thr", [2.4, 1.0, 0.5])
def test_correspondences_index_exact(ctx, pairs_small, cls, thr):
    pair, T_gt = pairs_small[0]
    src = pyoracle.transform(pair.src[cls], pair.init_guess)
    tgt = pair.tgt[cls]
    nc = cls != abi.VERTEX
    m0, d0, f0 = pyoracle.correspond(src, tgt, thr, nc, 20.0, nn_mode=1)
    m1, d1, f1 = ctx.correspond(src, tgt, thr, nc, 20.0)
    assert np.array_equal(m0, m1)
    assert np.array_equal(d0[m0 >= 0].view(np.uint32), d1[m0 >= 0].view(np.uint32))
    assert np.array_equal(f0, f1)
    assert (f0 & 2).sum() > 10


def test_correspondences_full_size_and_ragged_edges(ctx):
    """KITTI-sized class clouds (above and below the 500-point gate), odd sizes around wave / tile boundaries."""
    rng = np.random.default_rng(1)
    for ns, nt in ((499, 2049), (500, 2048), (1200, 6000), (513, 4097), (3, 3), (2, 100), (100, 2), (1025, 17)):
        tx = rng.uniform(-30, 30, (nt, 3))
        tgt = abi.make_points(tx, rng.normal(size=(nt, 3)))
        sx = tx[rng.integers(0, nt, ns)] + rng.normal(0, 0.3, (ns, 3))
        sx[: ns // 10] += 50.0  # a few sources with no neighbour inside the radius
        src = abi.make_points(sx, rng.normal(size=(ns, 3)))
        m0, d0, f0 = pyoracle.correspond(src, tgt, 0.8, True, 60.0, nn_mode=1)
        m1, d1, f1 = ctx.correspond(src, tgt, 0.8, True, 60.0)
        assert np.array_equal(m0, m1), (ns, nt)
        assert np.array_equal(d0[m0 >= 0].view(np.uint32), d1[m0 >= 0].view(np.uint32)), (ns, nt)
        assert np.array_equal(f0, f1), (ns, nt)


def test_correspondences_exact_ties_pick_lowest_index(ctx):
    rng = np.random.default_rng(2)
    base = rng.uniform(-5, 5, (700, 3)).astype(np.float32)
    tgt = abi.make_points(np.concatenate([base, base, base]), np.tile([0, 0, 1], (2100, 1)))
    src = abi.make_points(base + np.float32(0.01), np.tile([0, 0, 1], (700, 1)))
    m0, _, f0 = pyoracle.correspond(src, tgt, 1.0, True, 45.0, nn_mode=1)
    m1, _, f1 = ctx.correspond(src, tgt, 1.0, True, 45.0)
    assert (m1 < 700).all() and np.array_equal(m0, m1) and np.array_equal(f0, f1)


def test_duplicate_rule_first_source_wins(ctx):
    rng = np.random.default_rng(3)
    tgt = abi.make_points(rng.uniform(-20, 20, (64, 3)), np.tile([0, 0, 1], (64, 1)))
    txyz = np.column_stack([tgt["x"], tgt["y"], tgt["z"]])
    for n in (499, 500, 1500):
        src = abi.make_points(txyz[rng.integers(0, 64, n)] + rng.normal(0, 0.01, (n, 3)), np.tile([0, 0, 1], (n, 1)))
        m0, _, f0 = pyoracle.correspond(src, tgt, 1.0, True, 45.0, nn_mode=1)
        m1, _, f1 = ctx.correspond(src, tgt, 1.0, True, 45.0)
        assert np.array_equal(m0, m1) and np.array_equal(f0, f1)
        if n >= 500:
            assert (f1 & 1).sum() == len(np.unique(m1))


def test_empty_inputs(ctx):
    empty = np.zeros(0, abi.POINT_DTYPE)
    some = abi.make_points(np.random.default_rng(0).uniform(-1, 1, (10, 3)))
    m, d, f = ctx.correspond(empty, some, 1.0)
    assert len(m) == 0
    m, d, f = ctx.correspond(some, empty, 1.0)
    assert (m == -1).all() and (f & 2).sum() == 0


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("flags", [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 1)])
def test_accumulate_matches_oracle(ctx, pairs_small, metric, flags):
    pair, _ = pairs_small[1]
    cls = (abi.FACADE, abi.PILLAR, abi.VERTEX)[metric]
    src = pyoracle.transform(pair.src[cls], pair.init_guess)
    tgt = pair.tgt[cls]
    m, d2, f = pyoracle.correspond(src, tgt, 1.0, cls != abi.VERTEX, 30.0, nn_mode=1)
    cs = np.nonzero(f & 2)[0].astype(np.int32)
    ct = m[cs]
    assert len(cs) > 20
    it = 5
    a, wa = pyoracle.accumulate(metric, src, tgt, cs, ct, d2[cs], it, 0.37, flags[0], flags[1], flags[2], 0.05)
    b, wb = ctx.accumulate(metric, src, tgt, cs, ct, d2[cs], it, 0.37, flags[0], flags[1], flags[2], 0.05)
    scale = np.abs(a).max()
    assert np.abs(a - b).max() <= 1e-12 * scale
    # weights are float32 products of identical operands; exp() may differ by one ulp between libm and ocml
    assert np.abs(wa - wb).max() <= 1.2e-7 * np.abs(wa).max()


def test_accumulate_single_correspondence_is_bit_exact(ctx):
    """No reduction involved -> the float products and the double conversion must agree bit for bit."""
    src = abi.make_points([[1.5, -2.0, 0.25]], None, [10.0])
    tgt = abi.make_points([[1.25, -2.5, 0.0]], [[0.0, 0.6, 0.8]], [30.0])
    for metric in (0, 1, 2):
        a, _ = pyoracle.accumulate(metric, src, tgt, [0], [0], [0.1], 0, 1.0, 1, 0, 0, 0.1)
        b, _ = ctx.accumulate(metric, src, tgt, [0], [0], [0.1], 0, 1.0, 1, 0, 0, 0.1)
        assert np.array_equal(a, b), metric
