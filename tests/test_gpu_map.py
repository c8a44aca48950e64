"""GPU parity of the device-resident local map (mulls_map_*) against the oracle's update_local_map restatement, and of a
scan-to-map registration whose target never leaves the device."""
import numpy as np
import pytest

from mulls_amd import abi, synth
from oracle import pyoracle
from test_map import linear_frames, same_cloud, small_frames

pytestmark = pytest.mark.gpu


def drive(ctx_auto, frames, params_of):
    """Run the same frame sequence through the device map and the oracle; compare after every update."""
    dev = ctx_auto.local_map(frames[0][0], frames[0][1])
    clouds, pose = [c.copy() for c in frames[0][0]], frames[0][1]
    reps = []
    for k, (fc, fp) in enumerate(frames[1:], 1):
        P = params_of(k)
        clouds, appended, ro = pyoracle.map_update(clouds, pose, fc, fp, P)
        pose = fp
        rg = dev.update(fc, fp, P)
        assert list(rg.n) == list(ro.n) and list(rg.frame_n) == list(ro.frame_n)
        assert rg.feature_point_num == ro.feature_point_num and rg.dynamic_removal_ran == ro.dynamic_removal_ran
        assert list(rg.local_bound) == list(ro.local_bound) and list(rg.bound) == list(ro.bound)
        for c in range(6):
            same_cloud(dev.download(c), clouds[c])
            same_cloud(dev.frame_download(c), appended[c])
            if P.recalculate_feature_on and c in (abi.PILLAR, abi.BEAM):  # linearity in normal[3]
                assert np.array_equal(abi.normal3(dev.download(c)), abi.normal3(clouds[c]))
        assert np.array_equal(dev.pose(), fp)
        reps.append(rg)
    return dev, clouds, reps


@pytest.mark.parametrize("used", ["111110", "101000", "111111"])
def test_update_matches_oracle(ctx_auto, used):
    frames = small_frames(40, n_frames=4)
    dev, _, _ = drive(ctx_auto, frames, lambda k: abi.map_params(used_feature_type=used, max_num_pts=10**7, kept_vertex_num=10**6,
                                                                local_map_radius=45.0))
    dev.close()


def test_update_with_thinning_and_radius(ctx_auto):
    """The seeded selection sampling is the ABI's definition (shared with the oracle): the thinned maps are bit-identical."""
    frames = small_frames(60, n_frames=5)
    dev, clouds, reps = drive(ctx_auto, frames, lambda k: abi.map_params(max_num_pts=1800, kept_vertex_num=130, local_map_radius=35.0,
                                                                        rng_seed=1000 + k))
    assert reps[-1].n[5] == 130 and reps[-1].feature_point_num <= 1805
    dev.close()


@pytest.mark.parametrize("tree_mode", [1, 2])
def test_dynamic_removal_matches_oracle(ctx_auto, tree_mode):
    frames = small_frames(50, n_frames=4)
    box = [-28.0, -14.0, -3.0, 30.0, 14.0, 8.0]

    def P(k):
        return abi.map_params(max_num_pts=6000, kept_vertex_num=10**6, map_based_dynamic_removal_on=1, dynamic_removal_center_radius=25.0,
                              dynamic_dist_thre_min=0.25, dynamic_dist_thre_max=1.0, near_dist_thre=0.05, tree_mode=tree_mode,
                              tree_used="111110" if k != 2 else "101000", tree_box=box)

    dev, _, reps = drive(ctx_auto, frames, P)
    assert all(r.dynamic_removal_ran == 1 for r in reps)
    assert sum(int(len(frames[k + 1][0][c]) - reps[k].frame_n[c]) for k in range(len(reps)) for c in (1, 2, 3)) > 0
    dev.close()


def test_scan_to_map_registration_with_resident_target(ctx_auto):
    """mm_lls_icp against the device-resident map == the same registration with the map downloaded and re-uploaded."""
    frames = small_frames(80, n_frames=3)
    dev, clouds, _ = drive(ctx_auto, frames, lambda k: abi.map_params(max_num_pts=10**7, kept_vertex_num=10**6))
    src_pair, T_gt = synth.make_pair(83, n_beams=32, n_az=700, src_counts={abi.GROUND: 500, abi.PILLAR: 200, abi.FACADE: 600, abi.BEAM: 120,
                                                                           abi.ROOF: 60}, vertex_count=150)
    src = [src_pair.src[c] for c in range(6)]
    host_pair = abi.PairData(clouds, src)
    P = abi.kitti_params(dis_thre_unit=2.4)
    r_host = ctx_auto.icp(host_pair, P)[0]
    r_dev = dev.icp(src, P, tgt_bound=host_pair.tgt_bound)[0]
    r_ora = pyoracle.icp(host_pair, P)[0]
    assert (r_dev.code, r_dev.iters, list(r_dev.ncorr)) == (r_host.code, r_host.iters, list(r_host.ncorr))
    assert r_dev.T[:] == r_host.T[:] and r_dev.info[:] == r_host.info[:] and r_dev.sigma == r_host.sigma
    assert (r_ora.code, r_ora.iters, list(r_ora.ncorr)) == (r_dev.code, r_dev.iters, list(r_dev.ncorr))
    dt, dr = synth.pose_error(r_dev.T_matrix(), r_ora.T_matrix())
    assert dt <= 1e-7 and dr <= 1e-7
    # the registration's crop box is what the next update's dynamic removal uses as tree contents
    rep = dev.update(src, frames[-1][1], abi.map_params(max_num_pts=6000, map_based_dynamic_removal_on=1, tree_mode=2 if r_dev.cropped else 1,
                                                       tree_used="111000", tree_box=list(r_dev.crop_box)))
    assert rep.dynamic_removal_ran == 1
    dev.close()


@pytest.mark.parametrize("thin", [False, True])
def test_pca_refresh_matches_oracle(ctx_auto, thin):
    """recalculate_feature_on: the device's neighbourhood PCA is the oracle's arithmetic operation by operation (float sums in
    neighbour order, the same Jacobi rotations in double), so kept sets, directions and linearities are bit-identical; the
    refreshed directions then feed the next frame's update."""
    frames = linear_frames(310, n_frames=4)
    kw = dict(max_num_pts=2500, kept_vertex_num=60, rng_seed=77) if thin else dict(max_num_pts=10**7, kept_vertex_num=10**6)
    dev, clouds, reps = drive(ctx_auto, frames, lambda k: abi.map_params(recalculate_feature_on=1, local_map_radius=60.0,
                                                                        used_feature_type="111110" if k != 2 else "110010", **kw))
    assert 0 < reps[-1].n[abi.PILLAR] and 0 < reps[0].n[abi.BEAM]
    assert np.all(np.abs(dev.download(abi.PILLAR)["nz"]) > 0.80) and np.all(np.abs(dev.download(abi.BEAM)["nz"]) < 0.25)
    dev.close()


def test_map_argument_errors(ctx_auto):
    m = ctx_auto.local_map()
    assert all(len(m.download(c)) == 0 for c in range(6))
    frames = small_frames(90, n_frames=2)
    with pytest.raises(Exception):
        m.update(frames[1][0], frames[1][1], abi.map_params(used_feature_type="111"))
    rep = m.update(frames[1][0], frames[1][1], abi.map_params())  # empty map + first frame
    assert rep.n[0] == len(frames[1][0][0])
    m.close()


def test_scan_to_map_odometry_loop(ctx_auto):
    """The scan-to-map loop of test/mulls_slam.cpp on a synthetic drive: register every new frame against the
    device-resident local map, then fold it into the map (with map-based dynamic removal driven by that registration's
    crop box).  The oracle runs the same loop on the host from the same poses; registrations and maps must agree."""
    counts = {abi.GROUND: 500, abi.PILLAR: 200, abi.FACADE: 600, abi.BEAM: 120, abi.ROOF: 60}
    frames = synth.drive(5, 6, counts=counts)
    dev = ctx_auto.local_map(frames[0][0], np.eye(4))
    host_map, host_pose = [c.copy() for c in frames[0][0]], np.eye(4)
    P = abi.kitti_params(dis_thre_unit=2.0, used_feature_type="111110")
    pose = np.eye(4)  # estimated pose of the last frame folded into the map
    for k in range(1, len(frames)):
        clouds, gt = frames[k]
        guess = np.linalg.inv(frames[k - 1][1]) @ gt @ synth.se3(0.15, -0.1, 0.02, 0, 0, np.deg2rad(0.4))  # perturbed relative motion
        bound = abi.PairData(host_map, clouds).tgt_bound
        rg = dev.icp(clouds, P, init_guess=guess, tgt_bound=bound)[0]
        ro = pyoracle.icp(abi.PairData(host_map, clouds, init_guess=guess, tgt_bound=bound), P)[0]
        assert (rg.code, rg.iters, list(rg.ncorr)) == (ro.code, ro.iters, list(ro.ncorr)) and rg.code == 1
        dt, dr = synth.pose_error(rg.T_matrix(), ro.T_matrix())
        assert dt <= 1e-7 and dr <= 1e-7
        assert list(rg.crop_box) == list(ro.crop_box) and rg.cropped == ro.cropped
        pose = pose @ rg.T_matrix()  # both sides continue from the device's estimate (bit-identical inputs for the map update)
        et, er = synth.pose_error(pose, gt)
        assert et < 0.15 and er < 0.01, (k, et, er)
        M = abi.map_params(max_num_pts=4000, kept_vertex_num=300, local_map_radius=60.0, map_based_dynamic_removal_on=1, rng_seed=k,
                           dynamic_dist_thre_min=0.3, dynamic_dist_thre_max=1.5, tree_mode=2 if rg.cropped else 1,
                           tree_used="".join("1" if (P.used_feature_type[c:c + 1] == b"1" and rg.ntgt0[c] > 0) else "0" for c in range(6)),
                           tree_box=list(rg.crop_box))
        rep = dev.update(clouds, pose, M)
        host_map, _, rep_o = pyoracle.map_update(host_map, host_pose, clouds, pose, M)
        host_pose = pose
        assert list(rep.n) == list(rep_o.n) and list(rep.frame_n) == list(rep_o.frame_n) and rep.dynamic_removal_ran == rep_o.dynamic_removal_ran
        for c in range(6):
            same_cloud(dev.download(c), host_map[c])
    assert rep.feature_point_num <= 4005
    dev.close()


@pytest.mark.parametrize("block", range(3))
def test_random_update_sequences_match_oracle(ctx_auto, block):
    """Seeded random sequences: frames with empty / tiny / duplicated class clouds, random radii, caps small enough to thin,
    dynamic removal with every tree state, class sets with holes.  Device and oracle must agree bit for bit after every update."""
    rng = np.random.default_rng(4200 + block)
    frames = small_frames(100 + block, n_frames=5)

    def mutate(clouds):
        out = []
        for c in clouds:
            roll = rng.random()
            if roll < 0.15:
                out.append(c[:0])
            elif roll < 0.3:
                out.append(c[: int(rng.integers(1, 12))])
            elif roll < 0.4:
                out.append(np.concatenate([c, c[: len(c) // 2]]))  # exact duplicates
            else:
                out.append(c)
        return out

    def params(k):
        used = "".join(rng.choice(["0", "1"], p=[0.25, 0.75]) for _ in range(6))
        box = sorted(rng.uniform(-40, 40, 2)) + sorted(rng.uniform(-20, 20, 2)) + sorted(rng.uniform(-4, 8, 2))
        return abi.map_params(used_feature_type=used, max_num_pts=int(rng.choice([300, 1500, 6000, 10**7])),
                              kept_vertex_num=int(rng.choice([0, 50, 10**6])), local_map_radius=float(rng.uniform(15, 80)),
                              map_based_dynamic_removal_on=int(rng.random() < 0.7), dynamic_removal_center_radius=float(rng.uniform(5, 40)),
                              dynamic_dist_thre_min=float(rng.uniform(0.1, 0.6)), dynamic_dist_thre_max=float(rng.uniform(0.2, 3.0)),
                              near_dist_thre=float(rng.uniform(0.0, 0.1)), rng_seed=int(rng.integers(0, 2**31)), tree_mode=int(rng.integers(0, 3)),
                              tree_used="".join(rng.choice(["0", "1"]) for _ in range(6)), tree_box=[box[0], box[2], box[4], box[1], box[3], box[5]])

    frames = [(mutate(fc), fp) for fc, fp in frames]
    drive(ctx_auto, frames, params)[0].close()
