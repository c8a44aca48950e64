"""LiDAR odometry front end on the device, frame by frame as test/mulls_slam.cpp runs it: raw scan -> extract_semantic_pts
(mulls_extract_features) -> scan-to-map mm_lls_icp against the device-resident local map -> update_local_map (map-based dynamic removal, PCA
refresh of the linear features every 5th frame).  Synthetic drive through one scene; prints the time per frame by stage and the drift against
the ground truth.  usage: gpu_odometry.py [frames] [--check N: run the first N frames through the oracle too and compare] [--host: the frame's
feature clouds come back to the host and are uploaded again by the registration and the map update (the round-2 form); default: they stay in a
device-resident feature block (mulls_extract_features_resident), only the scan, selection indices and the pose cross PCIe]
[--motion-compensation 1: test/mulls_slam.cpp:703-712 — the frame's clouds are moved by their time-stamp fraction of the estimated motion before they enter the
map (mulls_block_motion_compensate on the resident block; on in the reference's 32- and 128-beam configurations)]"""
import sys, time, warnings
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
import numpy as np
from mulls_amd import abi, lib, synth
from oracle import pyoracle


def raw_drive(seed, n_frames, n_beams=64, n_az=1900, step=0.9):
    rng = np.random.default_rng(seed)
    scene = synth.Scene(seed)
    world0 = synth.se3(0, 0, scene.sensor_height)
    pose = np.eye(4)
    frames = []
    for k in range(n_frames):
        s = synth.raycast(scene, world0 @ pose, n_beams, n_az, seed=seed * 7 + k)
        frames.append((abi.make_points(s["xyz"], np.zeros_like(s["xyz"]), s["intensity"], s["t"]), pose.copy()))
        pose = pose @ synth.se3(rng.uniform(0.7, 1.3) * step, rng.normal(0, 0.03), rng.normal(0, 0.01), 0, 0, np.deg2rad(rng.normal(0, 1.0)))
    return frames


def block_of(ex):
    """(full clouds, down clouds) in class order ground, pillar, facade, beam, roof, vertex from the clouds of mulls_extract_features"""
    c = ex[abi.EX_PILLAR:]
    full = [ex[abi.EX_GROUND], c[abi.CL_PILLAR], c[abi.CL_FACADE], c[abi.CL_BEAM], c[abi.CL_ROOF], c[abi.CL_VERTEX]]
    down = [ex[abi.EX_GROUND_DOWN], c[abi.CL_PILLAR_DOWN], c[abi.CL_FACADE_DOWN], c[abi.CL_BEAM_DOWN], c[abi.CL_ROOF_DOWN], c[abi.CL_VERTEX]]
    return [abi.points_of(x) for x in full], [abi.points_of(x) for x in down]


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    n_frames = int(argv[0]) if argv and not argv[0].startswith("-") else 12
    n_check = int(argv[argv.index("--check") + 1]) if "--check" in argv else 0
    resident = "--host" not in argv
    mocomp = "--motion-compensation" in argv and argv[argv.index("--motion-compensation") + 1] not in ("0", "false")
    del frames_pose[1:]
    frames = raw_drive(11, n_frames)
    X = abi.extract_params(ground=abi.ground_params(fixed_num_downsampling=1, down_ground_fixed_num=800, rng_seed=1),
                           classify=abi.classify_params(neighbor_searching_radius=0.7, neighbor_k=25, neigh_k_min=7, curvature_thre=0.08, fixed_num_downsampling=1,
                                                        unground_down_fixed_num=20000, pillar_down_fixed_num=400, facade_down_fixed_num=1200, beam_down_fixed_num=200,
                                                        roof_down_fixed_num=200, rng_seed=1),
                           apply_dist_filter=1, min_dist_used=1.5, max_dist_used=120.0)  # lo_gflag_list_kitti_urban.txt: --apply_dist_filter=true, 1.5 m .. 120 m
    P = abi.kitti_params(dis_thre_unit=1.5, used_feature_type="111110")
    ctx = lib.Context(0)
    t_feat = t_reg = t_map = 0.0
    ex = ctx.extract_features(frames[0][0], X)
    full, down = block_of(ex)
    dev = ctx.local_map(down, np.eye(4))  # the map is made of the frames' down-sampled clouds (update_local_map appends pc_*_down)
    host_map = [abi.points_of(abi.records(c).copy()) for c in down]  # (.copy() of a record array drops the bytes between its fields)
    bound = list(abi.PairData(down, down).tgt_bound)
    pose, prev_rel = np.eye(4), np.eye(4)
    worst = (0.0, 0.0)
    blk = ctx.block() if resident else None
    for k in range(1, n_frames):
        scan, gt = frames[k]
        t0 = time.time()
        if resident:
            blk.extract(scan, X)
            down = blk.class_clouds(down=True)  # device clouds: the registration's source and the map update's frame
        else:
            ex = ctx.extract_features(scan, X)
            full, down = block_of(ex)
        t1 = time.time()
        if resident and k <= n_check:  # (the check's copy of the block, before the motion compensation moves its clouds)
            ex = [None if n in (abi.EX_RAW, abi.EX_DOWN) else blk.download(n) for n in range(abi.EX_COUNT)]
        rg = dev.icp(down, P, init_guess=prev_rel, tgt_bound=bound)[0]  # constant-velocity guess
        t2 = time.time()
        rel = rg.T_matrix()
        pose = pose @ rel
        if mocomp:
            adjacent = np.linalg.inv(rel)  # adjacent_pose_out: frame k -> frame k + 1 (test/mulls_slam.cpp:699)
            if resident:
                blk.motion_compensate(adjacent)
            else:
                down = [ctx.motion_compensate(c, adjacent) if i < 5 else c for i, c in enumerate(down)]
        MP = abi.map_params(max_num_pts=20000, map_based_dynamic_removal_on=1, tree_mode=2 if rg.cropped else 1, tree_used="111000", tree_box=list(rg.crop_box),
                            recalculate_feature_on=1 if k % 5 == 0 else 0, rng_seed=k)
        rep = dev.update(down, pose, MP)
        t3 = time.time()
        if k > 1:  # the first frame pays the allocations
            t_feat += t1 - t0; t_reg += t2 - t1; t_map += t3 - t2
        et, er = synth.pose_error(pose, gt)
        worst = (max(worst[0], et), max(worst[1], er))
        if k <= n_check:
            exo = pyoracle.extract_features(scan, X)
            if resident:
                assert all(a is None or np.array_equal(a, b) for a, b in zip(ex, exo)), "features differ from the oracle's at frame %d" % k
            else:
                assert all(np.array_equal(a, b) for a, b in zip(ex, exo)), "features differ from the oracle's at frame %d" % k
            fo, do = block_of(exo)
            ro = pyoracle.icp(abi.PairData(host_map, do, init_guess=prev_rel, tgt_bound=bound), P)[0]
            dt, dr = synth.pose_error(rg.T_matrix(), ro.T_matrix())
            assert rg.code == ro.code and dt <= 1e-6 and dr <= 1e-6, (k, rg.code, ro.code, dt, dr)
            if mocomp:
                do = [pyoracle.motion_compensate(c, np.linalg.inv(rel)) if i < 5 else c for i, c in enumerate(do)]
            host_map, _, _ = pyoracle.map_update(host_map, frames_pose[-1], do, pose, MP)
            for c in range(6):
                a, b = abi.records(dev.download(c)), abi.records(host_map[c])
                if a.shape != b.shape or not np.array_equal(a, b):
                    cols = sorted(set(np.nonzero(a != b)[1].tolist())) if a.shape == b.shape else "sizes %s %s" % (a.shape, b.shape)
                    raise AssertionError("map class %d differs at frame %d: byte columns %s" % (c, k, cols))
        frames_pose.append(pose.copy())
        prev_rel = rel
        bound = list(rep.local_bound)
        assert rg.code == 1, (k, rg.code)
    m = max(n_frames - 2, 1)
    print(("motion compensation on; " if mocomp else "") + ("feature block resident in HBM: " if resident else "feature clouds through the host: ") + "%d frames of %d returns: features %.2f ms, scan-to-map registration %.2f ms, map update %.2f ms per frame -> %.1f frames/s; drift after %.0f m: %.3f m, %.4f rad"
          % (n_frames, len(frames[0][0]), t_feat / m * 1e3, t_reg / m * 1e3, t_map / m * 1e3, m / max(t_feat + t_reg + t_map, 1e-9),
             np.linalg.norm(frames[-1][1][:3, 3]), worst[0], worst[1]))
    if n_check:
        print("first %d frames: features, registration and local map identical to the oracle's chain" % n_check)
    return worst


frames_pose = [np.eye(4)]
if __name__ == "__main__":
    main()
