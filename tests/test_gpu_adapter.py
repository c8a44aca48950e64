"""End-to-end drop-in check on the GPU box: oracle/_ref/adapter_check calls, on one lo::constraint_t, the reference's own
CRegistration<Point_T>::mm_lls_icp (its source lines, CPU) and the adapter lo::hip::mm_lls_icp<Point_T> from
include/cregistration_hip.hpp (-> C ABI -> MI355X), with the positional arguments of test/mulls_slam.cpp:642-648 and
test/mulls_reg.cpp:194-195.  Both write Trans1_2 / information_matrix / sigma / confidence into their constraint_t and
return the process code; the kd-tree side effect on block1 must cover the same number of points."""
import json
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

from mulls_amd import abi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "adapter_check")

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/adapter_check not built")]


def dump(pair, path):
    with open(path, "wb") as f:
        for clouds in (pair.tgt, pair.src):
            for c in clouds:
                f.write(struct.pack("<I", len(c)))
                f.write(np.ascontiguousarray(c).tobytes())
        f.write(np.asarray(pair.init_guess, np.float64).T.tobytes())  # column-major
        f.write(np.asarray(pair.tgt_bound, np.float64).tobytes())


@pytest.mark.parametrize("mode", ["kitti", "reg"])
def test_adapter_matches_reference_class(pairs_small, mode):
    for pair, T_gt in pairs_small:
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "pair.bin")
            dump(pair, path)
            out = subprocess.check_output([BIN, path, mode], timeout=120).decode().strip().split("\n")
        ref, hip = json.loads(out[0]), json.loads(out[1])
        assert ref["who"] == "reference" and hip["who"] == "hip"
        assert ref["code"] == hip["code"] == 1
        Tr, Th = np.array(ref["T"]).reshape(4, 4).T, np.array(hip["T"]).reshape(4, 4).T
        dt, dr = synth.pose_error(Th, Tr)
        assert dt <= 1e-7 and dr <= 1e-7, (dt, dr)  # contract: 1e-4 m / 1e-4 rad
        Ir, Ih = np.array(ref["info"]), np.array(hip["info"])
        assert np.abs(Ir - Ih).max() <= 1e-6 * np.abs(Ir).max()
        assert abs(ref["sigma"] - hip["sigma"]) <= 1e-6 and ref["confidence"] == hip["confidence"]
        assert ref["tree_points"] == hip["tree_points"] > 0
        gt_t, gt_r = synth.pose_error(Th, T_gt)
        assert gt_t < 0.05 and gt_r < 2e-3


def test_adapter_variants_match_reference_members(pairs_small):
    """lo::hip::lls_icp_3dof_ground / mm_lls_icp_4dof_global vs the reference members, same constraint_t, same arguments."""
    pair, _ = pairs_small[1]
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "pair.bin")
        dump(pair, path)
        out = subprocess.check_output([BIN, path, "variants"], timeout=300).decode().strip().split("\n")
    rows = {json.loads(l)["who"]: json.loads(l) for l in out}
    for v in ("3dof", "4dof"):
        ref, hip = rows["reference_" + v], rows["hip_" + v]
        assert ref["code"] == hip["code"]
        Tr, Th = np.array(ref["T"]).reshape(4, 4).T, np.array(hip["T"]).reshape(4, 4).T
        dt, dr = synth.pose_error(Th, Tr)
        assert dt <= 1e-7 and dr <= 1e-7, (v, dt, dr)
        if v == "4dof" and ref["code"]:
            assert abs(ref["sigma"] - hip["sigma"]) <= 1e-6 and ref["confidence"] == hip["confidence"]


def test_adapter_local_map_matches_reference_map_manager(pairs_small):
    """Scan-to-map step: mm_lls_icp against the local map, then MapManager::update_local_map with map-based dynamic removal —
    the reference class with the kd-trees its registration left on block1, the bridge with the device-resident mirror."""
    for pair, _ in pairs_small[:2]:
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "pair.bin")
            dump(pair, path)
            out = subprocess.check_output([BIN, path, "map"], timeout=300).decode().strip().split("\n")
        rows = {json.loads(l)["who"]: json.loads(l) for l in out}
        ref, hip = rows["reference_map"], rows["hip_map"]
        assert ref["code"] == hip["code"] == 1
        for key in ("n", "frame_n", "hash", "frame_hash", "feature_point_num", "local_bound", "bound"):
            assert ref[key] == hip[key], key
        assert sum(ref["frame_n"]) < sum(len(pair.src[c]) for c in range(5))  # the removal filtered something


def test_adapter_feature_extraction_matches_reference_members():
    """lo::hip::fast_ground_filter / classify_nground_pts vs the reference's CFilter members on one raw scan: every output cloud the same bytes."""
    from test_ground_filter import raw_scan

    from test_classify import with_ego_and_ghost_points

    scan = with_ego_and_ghost_points(raw_scan(14, n_beams=48, n_az=1400))
    # Semantic-KITTI labels in the curvature field (only the semantic_assisted runs read them; every run's clouds carry them along)
    labels = np.array([0, 1, 10, 40, 44, 48, 50, 51, 71, 80, 99, 249, 250, 252, 255, 259], np.float32)
    scan = scan.copy()
    scan["curvature"] = labels[np.random.default_rng(21).integers(0, len(labels), len(scan))]
    n_semantic_kept = int(((scan["curvature"] < 250) & (scan["curvature"] != 1)).sum())
    empty = np.zeros(0, abi.POINT_DTYPE)
    pair = abi.PairData([scan] + [empty] * 5, [empty] * 6)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "scan.bin")
        dump(pair, path)
        out = subprocess.check_output([BIN, path, "features"], timeout=300).decode().strip().split("\n")
    ref, hip = json.loads(out[0]), json.loads(out[1])
    assert ref["who"] == "reference" and hip["who"] == "hip"
    assert ref["sizes"] == hip["sizes"] and ref["sums"] == hip["sums"]
    assert ref["sizes"][2] > 1000  # (the scan carries underground ghost points: without the scanner filter the ground filter finds little ground)
    # ... and through the chain's one entry point, CFilter::extract_semantic_pts vs lo::hip::extract_semantic_pts, scanner filter on, on cloudblocks
    refb, hipb = json.loads(out[2]), json.loads(out[3])
    assert refb["who"] == "reference_block" and hipb["who"] == "hip_block"
    assert refb["sizes"] == hipb["sizes"] and refb["sums"] == hipb["sums"] and refb["down_feature_point_num"] == hipb["down_feature_point_num"] > 0
    assert refb["sizes"][0] == len(scan) - 1200 and refb["sizes"][1] == refb["sizes"][0] and 0 < refb["sizes"][2] <= 1024  # pc_raw filtered, pc_down = pc_raw, pc_sketch
    assert refb["sizes"][3] > 1000 and refb["sizes"][8] > 1000 and sum(refb["sizes"][10:14]) > 100  # ground, facade, the *_down clouds
    # ... and with voxel_downsample ahead of the ground filter and the adaptive parameter update after it
    refv, hipv = json.loads(out[4]), json.loads(out[5])
    assert refv["who"] == "reference_block_voxels" and hipv["who"] == "hip_block_voxels"
    assert refv["sizes"] == hipv["sizes"] and refv["sums"] == hipv["sums"] and refv["down_feature_point_num"] == hipv["down_feature_point_num"] > 0
    assert refv["sizes"][0] == refb["sizes"][0] and 1000 < refv["sizes"][1] < refv["sizes"][0]  # pc_raw as before, pc_down one point per voxel
    assert refv["rates"] == hipv["rates"] and refv["rates"][0] == 10 and 1 <= refv["rates"][1] < 20  # 20 - 200 / (facade_down + pillar_down)
    assert refb["rates"] == hipb["rates"] == [10, 3]
    # ... and with semantic_assisted (round 5): the label pre-filter filter_with_dynamic_object_mask_pre in place of the scanner filter (upstream's `else if`),
    # the reference's own lines against the bridge's host statement of them + the device chain
    refs, hips = json.loads(out[6]), json.loads(out[7])
    assert refs["who"] == "reference_block_semantic" and hips["who"] == "hip_block_semantic"
    assert refs["sizes"] == hips["sizes"] and refs["sums"] == hips["sums"] and refs["down_feature_point_num"] == hips["down_feature_point_num"] > 0
    assert refs["sizes"][0] == n_semantic_kept < len(scan) and refs["sizes"][1] == refs["sizes"][0]  # moving objects (>= 250) and outliers (1) left pc_raw; pc_down = pc_raw
    assert sum(refs["sizes"][3:]) > 1000 and refs["sizes"] != refb["sizes"]  # (the underground ghost points stay without the scanner filter: little ground, as in the first run)


def test_adapter_motion_compensation_matches_reference_members(pairs_small):
    """test/mulls_slam.cpp:703-712 through the bridge: lo::hip::apply_motion_compensation / batch_apply_motion_compensation against the reference's CFilter
    members on the same cloudblock_t (time stamps in the curvature field), float bit patterns compared (one ulp on at most 1e-4 of the coordinates: the
    device's acos / sin against glibc's)."""
    pair, T_gt = pairs_small[0]
    rng = np.random.default_rng(5)
    src = []
    for c in pair.src:
        c = c.copy()
        c["curvature"] = rng.uniform(-0.05, 1.05, len(c)).astype(np.float32)
        src.append(c)
    stamped = abi.PairData(pair.tgt, src, init_guess=synth.se3(1.1, 0.04, -0.02, 0.003, -0.002, 0.03), tgt_bound=pair.tgt_bound)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "pair.bin")
        dump(stamped, path)
        out = subprocess.check_output([BIN, path, "motion"], timeout=120).decode().strip().split("\n")
    ref, hip = json.loads(out[0]), json.loads(out[1])
    assert ref["who"] == "reference" and hip["who"] == "hip" and ref["sizes"] == hip["sizes"] and sum(ref["sizes"]) > 1000
    a, b = np.array(ref["xyz"], np.int64), np.array(hip["xyz"], np.int64)
    d = np.abs(a - b)
    assert d.max() <= 1 and (d != 0).sum() <= max(1, len(a) // 10000)
    moved = np.array(ref["xyz"], np.uint32).view(np.float32).reshape(-1, 3)[: ref["sizes"][0]]
    assert np.abs(moved - np.stack([src[0]["x"], src[0]["y"], src[0]["z"]], 1)).max() > 0.5  # the ground cloud really moved
