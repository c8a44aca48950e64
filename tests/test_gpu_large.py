"""Full-size parity cases for the large BASELINE.json configurations (they select the global-memory grid tier):

  #5  synthetic 128-beam ~240k-point clouds, regime D (every return assigned to a class, nothing down-sampled), all six
      classes, 40 iterations;
  #3  scan-to-local-map with a ~1M-point multi-frame submap as target.

Both are compared directly against the oracle (its kd-tree handles these sizes in seconds) and through size-independent
properties: registering a cloud onto itself is the identity, and A->B composed with B->A returns to the start."""
import numpy as np
import pytest

from mulls_amd import abi, synth
from oracle import pyoracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dense_pair():
    none = {c: None for c in range(abi.NCLASS)}
    pair, T_gt = synth.make_pair(301, n_beams=128, n_az=1875, elev_deg=(-25.0, 15.0), src_counts=none, tgt_counts=none, vertex_count=2000)
    return pair, T_gt


@pytest.fixture(scope="module")
def submap_pair():
    """target = 8 consecutive 64-beam scans merged in the frame of the first one (~0.9M feature points), source = the next scan."""
    scene = synth.Scene(302, extent=80.0)
    h = scene.sensor_height
    step = synth.se3(1.1, 0.02, 0.0, 0, 0, np.deg2rad(0.5))
    pose = synth.se3(0, 0, h)
    none = {c: None for c in range(abi.NCLASS)}
    tgt = [[] for _ in range(abi.NCLASS)]
    first_inv = np.linalg.inv(pose)
    for k in range(8):
        scan = synth.raycast(scene, pose, 64, 1900, seed=900 + k)
        clouds = synth.class_clouds(scan, none, seed=k, vertex_count=400)
        rel = first_inv @ pose
        for c in range(abi.NCLASS):
            if len(clouds[c]):
                tgt[c].append(pyoracle.transform(clouds[c], rel))
        pose = pose @ step
    tgt = [np.concatenate(t) if t else None for t in tgt]
    scan = synth.raycast(scene, pose, 64, 1900, seed=999)
    src = synth.class_clouds(scan, synth.R_SOURCE, seed=77, vertex_count=300)
    T_gt = first_inv @ pose
    guess = synth.se3(0.2, -0.1, 0.05, 0, 0, np.deg2rad(0.3)) @ T_gt
    return abi.PairData(tgt, src, init_guess=guess), T_gt


def close(ro, rg):
    assert ro.code == rg.code and ro.iters == rg.iters
    assert list(ro.ncorr) == list(rg.ncorr) and list(ro.nsrc0) == list(rg.nsrc0) and list(ro.ntgt0) == list(rg.ntgt0)
    for k in range(ro.trace_len):
        assert list(ro.trace[k].ncorr) == list(rg.trace[k].ncorr) and list(ro.trace[k].nsrc) == list(rg.trace[k].nsrc), k
    dt, dr = synth.pose_error(rg.T_matrix(), ro.T_matrix())
    assert dt <= 1e-7 and dr <= 1e-7, (dt, dr)
    assert abs(ro.sigma - rg.sigma) <= 1e-6


def test_config5_dense_128beam_all_classes(ctx_auto, dense_pair):
    pair, T_gt = dense_pair
    assert pair.n_raw[0] > 200000 and sum(len(c) for c in pair.src) > 200000
    P = abi.default_params(used_feature_type="111111", weight_strategy="1111", max_iter_num=40, dis_thre_unit=1.4, dis_thre_min=0.5,
                           converge_translation=0.0005, converge_rotation_d=0.001, normal_bearing=20.0, sigma_thre=0.35)
    ro = pyoracle.icp(pair, P, trace_cap=48)[0]
    rg = ctx_auto.icp(pair, P, trace_cap=48)[0]
    close(ro, rg)
    assert rg.code == 1
    dt, dr = synth.pose_error(rg.T_matrix(), T_gt)
    assert dt < 0.05 and dr < 2e-3


def test_config3_scan_to_submap(ctx_auto, submap_pair):
    pair, T_gt = submap_pair
    assert sum(len(c) for c in pair.tgt) > 700000
    P = abi.kitti_params(dis_thre_unit=1.4)
    ro = pyoracle.icp(pair, P, trace_cap=32)[0]
    rg = ctx_auto.icp(pair, P, trace_cap=32)[0]
    close(ro, rg)
    assert rg.code == 1
    dt, dr = synth.pose_error(rg.T_matrix(), T_gt)
    assert dt < 0.05 and dr < 2e-3


def test_self_registration_is_identity(ctx_auto, dense_pair):
    pair, _ = dense_pair
    self_pair = abi.PairData(pair.tgt, pair.tgt)
    P = abi.default_params(used_feature_type="111111", max_iter_num=6)
    r = ctx_auto.icp(self_pair, P)[0]
    dt, dr = synth.pose_error(r.T_matrix(), np.eye(4))
    assert r.code == 1 and dt < 1e-6 and dr < 1e-7 and r.sigma < 1e-5
    assert r.ncorr[abi.GROUND] == r.nsrc0[abi.GROUND]  # every point is its own neighbour at distance 0


def test_forward_backward_round_trip(ctx_auto):
    pair, T_gt = synth.make_pair(303)
    P = abi.kitti_params(dis_thre_unit=2.4)
    fwd = ctx_auto.icp(pair, P)[0]
    back_pair = abi.PairData(pair.src, [c[np.sort(np.random.default_rng(1).choice(len(c), size=min(len(c), n), replace=False))] if len(c) else c
                                       for c, n in zip(pair.tgt, (800, 400, 1200, 200, 100, 300))], init_guess=np.linalg.inv(pair.init_guess))
    bwd = ctx_auto.icp(back_pair, P)[0]
    assert fwd.code == 1 and bwd.code == 1
    dt, dr = synth.pose_error(fwd.T_matrix() @ bwd.T_matrix(), np.eye(4))
    assert dt < 0.05 and dr < 2e-3  # two independent registrations of noisy scans: within the noise floor


@pytest.fixture(scope="module")
def lds_limit_pair():
    """Class clouds at the edges of the LDS tier: targets of exactly MULLS_LDS_MAXPTS = 9728 points (the cell table then gets the
    minimum budget), source class clouds of 2500 / 5000 / 1030 queries (3, 5 and 2 equal chunks of <= 1024)."""
    src = {abi.GROUND: 2500, abi.PILLAR: 1030, abi.FACADE: 5000, abi.BEAM: 300, abi.ROOF: 200}
    tgt = {abi.GROUND: 9728, abi.PILLAR: 3000, abi.FACADE: 9728, abi.BEAM: 900, abi.ROOF: 600}
    pair, T_gt = synth.make_pair(303, n_beams=128, n_az=1875, elev_deg=(-25.0, 15.0), src_counts=src, tgt_counts=tgt, vertex_count=1500)
    assert len(pair.tgt[abi.GROUND]) == 9728 and len(pair.tgt[abi.FACADE]) == 9728 and len(pair.src[abi.FACADE]) == 5000
    return pair, T_gt


@pytest.mark.parametrize("used", ["111000", "111111"])
def test_lds_tier_at_its_limits(lds_limit_pair, used):
    """Forced LDS tier (mode 3): one pair (512-query jobs) and 200 copies (class-level jobs: duplicate rule, rejection chain, hints,
    cost order and correspondence records on chip) against the oracle; one point more in a target is refused in mode 3."""
    from mulls_amd import lib

    pair, _ = lds_limit_pair
    P = abi.default_params(used_feature_type=used, max_iter_num=12, converge_translation=0.0, converge_rotation_d=0.0)
    ro = pyoracle.icp(pair, P, trace_cap=32)[0]
    assert ro.iters == 12
    c = lib.Context(0)
    c.set_nn_mode(3)
    try:
        close(ro, c.icp(pair, P, trace_cap=32)[0])
        rb = c.icp_batch([pair] * 200, P)
        for i in (0, 57, 199):
            assert (rb[i].code, rb[i].iters, list(rb[i].ncorr)) == (ro.code, ro.iters, list(ro.ncorr))
            dt, dr = synth.pose_error(rb[i].T_matrix(), ro.T_matrix())
            assert dt <= 1e-7 and dr <= 1e-7
        big = list(pair.tgt)
        big[abi.GROUND] = np.concatenate([big[abi.GROUND], big[abi.GROUND][:1]])
        with pytest.raises(lib.MullsError):
            c.icp(abi.PairData(big, pair.src, init_guess=pair.init_guess, tgt_bound=pair.tgt_bound), P)
    finally:
        c.close()
