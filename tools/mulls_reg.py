"""Pairwise registration of two point clouds with the library, stage for stage what test/mulls_reg.cpp does (script/run_mulls_reg.sh):
read -> voxel_downsample (cloud_*_down_res, 0 = off as in run_mulls_reg.sh) -> fast_ground_filter -> classify_nground_pts per cloud -> the
cloud with more down-sampled feature points is the target -> mm_lls_icp -> the source's pc_down, transformed, written out.  Flags carry the
reference's names and defaults (test/mulls_reg.cpp:24-60).  Not here: the global coarse registration (TEASER / RANSAC on key-point
correspondences: --is_global_reg must be false, the initial guess is the identity), the viewers.

    python tools/mulls_reg.py --point_cloud_1_path a.pcd --point_cloud_2_path b.pcd --output_point_cloud_path b_reg.pcd --is_global_reg=false
"""
import argparse
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from mulls_amd import abi, lib  # noqa: E402


def flags(argv=None):
    def boolean(v):
        return str(v).lower() in ("1", "true", "yes", "on")

    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--point_cloud_1_path", required=True)
    p.add_argument("--point_cloud_2_path", required=True)
    p.add_argument("--output_point_cloud_path", default="")
    p.add_argument("--cloud_1_down_res", type=float, default=0.0)
    p.add_argument("--cloud_2_down_res", type=float, default=0.0)
    p.add_argument("--gf_grid_size", type=float, default=2.0)
    p.add_argument("--gf_in_grid_h_thre", type=float, default=0.3)
    p.add_argument("--gf_neigh_grid_h_thre", type=float, default=2.2)
    p.add_argument("--gf_max_h", type=float, default=3.0e38)
    p.add_argument("--gf_ground_down_rate", type=int, default=10)
    p.add_argument("--gf_nonground_down_rate", type=int, default=3)
    p.add_argument("--dist_inverse_sampling_method", type=int, default=0)
    p.add_argument("--unit_dist", type=float, default=15.0)
    p.add_argument("--ground_normal_method", type=int, default=3, help="extract_semantic_pts' default (cfilter.hpp:2304): 3 = plane RANSAC per grid cell")
    p.add_argument("--pca_distance_adpative_on", type=boolean, default=False)
    p.add_argument("--pca_neighbor_radius", type=float, default=1.0)
    p.add_argument("--pca_neighbor_count", type=int, default=30)
    p.add_argument("--linearity_thre", type=float, default=0.6)
    p.add_argument("--planarity_thre", type=float, default=0.6)
    p.add_argument("--curvature_thre", type=float, default=0.1)
    p.add_argument("--corr_dis_thre", type=float, default=2.0)
    p.add_argument("--reg_max_iter_num", type=int, default=25)
    p.add_argument("--converge_tran", type=float, default=0.001)
    p.add_argument("--converge_rot_d", type=float, default=0.01)
    p.add_argument("--is_global_reg", type=boolean, default=True)
    p.add_argument("--device", type=int, default=0)
    a, _ = p.parse_known_args(argv)  # glog / viewer flags of the reference's script are accepted and ignored
    return a


def read_cloud(path):
    """DataIo::read_pc_cloud_block(block, normalize_intensity_or_not = true) (dataio.hpp:1732-1756, test/mulls_reg.cpp:130-131): the intensity
    rescaled to 0 - 255 with the reference's float expressions"""
    p = lib.read_kitti_bin(path) if path.endswith(".bin") else lib.read_pcd(path)
    p = p.copy()
    inten = p["intensity"].astype(np.float32)
    if len(inten):
        lo, hi = np.float32(inten.min()), np.float32(inten.max())
        with np.errstate(divide="ignore", invalid="ignore"):
            p["intensity"] = (inten - lo) * np.float32(255.0 / float(hi - lo))
    return p


def scan_bound(p):
    """block->local_bound = the bounding box of pc_raw (get_cloud_bbx_cpt, dataio.hpp:1736)"""
    return [float(p[k].min()) for k in ("x", "y", "z")] + [float(p[k].max()) for k in ("x", "y", "z")]


def extract_semantic_pts(ctx, scan, F, vf_downsample_resolution):
    """CFilter::extract_semantic_pts (cfilter.hpp:2294-2413) as test/mulls_reg.cpp:134-143 calls it (its default ground normal method 3: the per-cell plane RANSAC).
    Returns (class clouds, their *_down clouds, pc_down)."""
    GP = abi.ground_params(min_grid_pt_num=8, grid_resolution=F.gf_grid_size, max_height_difference=F.gf_in_grid_h_thre,
                           neighbor_height_diff=F.gf_neigh_grid_h_thre, max_ground_height=F.gf_max_h, ground_random_down_rate=F.gf_ground_down_rate,
                           ground_random_down_down_rate=2, nonground_random_down_rate=F.gf_nonground_down_rate, reliable_neighbor_grid_num_thre=0,
                           estimate_ground_normal_method=F.ground_normal_method, distance_weight_downsampling_method=F.dist_inverse_sampling_method,
                           standard_distance=F.unit_dist, fixed_num_downsampling=0, down_ground_fixed_num=500, intensity_thre=3.0e38,
                           apply_grid_wise_outlier_filter=0)
    CP = abi.classify_params(neighbor_searching_radius=F.pca_neighbor_radius, neighbor_k=F.pca_neighbor_count, neigh_k_min=8, pca_down_rate=1,
                             edge_thre=F.linearity_thre, planar_thre=F.planarity_thre, edge_thre_down=F.linearity_thre + 0.1,
                             planar_thre_down=F.planarity_thre + 0.1, curvature_thre=F.curvature_thre,
                             vertex_curvature_non_max_radius=1.5 * F.pca_neighbor_radius, use_distance_adaptive_pca=int(F.pca_distance_adpative_on))
    pc_down = ctx.voxel_downsample(scan, vf_downsample_resolution) if vf_downsample_resolution >= 0.001 else abi.records(scan)
    ground, ground_down, unground = ctx.ground_filter(abi.points_of(pc_down), GP)
    c = ctx.classify_nground(unground, CP)
    full = [ground, c[abi.CL_PILLAR], c[abi.CL_FACADE], c[abi.CL_BEAM], c[abi.CL_ROOF], c[abi.CL_VERTEX]]
    down = [ground_down, c[abi.CL_PILLAR_DOWN], c[abi.CL_FACADE_DOWN], c[abi.CL_BEAM_DOWN], c[abi.CL_ROOF_DOWN], c[abi.CL_VERTEX]]
    return full, down, pc_down


def register(ctx, scan1, scan2, F):
    """Returns (abi.Result, which scan is the source: 1 or 2, that scan's pc_down)."""
    # test/mulls_reg.cpp:80-81, :134-143: block 1 is down-sampled with cloud_2_down_res, block 2 with cloud_1_down_res
    f1, d1, p1 = extract_semantic_pts(ctx, scan1, F, F.cloud_2_down_res)
    f2, d2, p2 = extract_semantic_pts(ctx, scan2, F, F.cloud_1_down_res)
    n1, n2 = sum(len(x) for x in d1), sum(len(x) for x in d2)
    # determine_source_target_cloud (cregistration.hpp:857-870): block1 (target) = the one with more down-sampled feature points
    (tgt_full, src_down, source) = (f1, d2, 2) if n1 > n2 else (f2, d1, 1)
    pair = abi.PairData([abi.points_of(t) for t in tgt_full], [abi.points_of(s) for s in src_down],
                        tgt_bound=scan_bound(abi.as_points(scan1 if source == 2 else scan2)) if len(scan1) and len(scan2) else None)
    # mm_lls_icp(reg_con, max_iter, thre, converge_tran, converge_rot_d, 0.25 * thre, 1.1, "111110", "1101", 1.0, 0.1, 0.1, 0.1, init_mat)
    P = abi.default_params(max_iter_num=F.reg_max_iter_num, dis_thre_unit=F.corr_dis_thre, converge_translation=F.converge_tran,
                           converge_rotation_d=F.converge_rot_d, dis_thre_min=0.25 * F.corr_dis_thre, dis_thre_update_rate=1.1,
                           used_feature_type="111110", weight_strategy="1101", z_xy_balanced_ratio=1.0, pt2pt_residual_window=0.1,
                           pt2pl_residual_window=0.1, pt2li_residual_window=0.1)
    return ctx.icp(pair, P)[0], source, (p1, p2)[source - 1]


def main(argv=None):
    F = flags(argv)
    if F.is_global_reg:
        sys.exit("the global coarse registration is out of scope: pass --is_global_reg=false (the initial guess is the identity)")
    ctx = lib.Context(F.device)
    scans = [read_cloud(F.point_cloud_1_path), read_cloud(F.point_cloud_2_path)]
    res, source, source_down = register(ctx, scans[0], scans[1], F)
    T = res.T_matrix()
    print("process code %d after %d iterations; source = point cloud %d; Trans1_2 =" % (res.code, res.iters, source))
    print(np.array2string(T, precision=6, suppress_small=True))
    if F.output_point_cloud_path:
        moved = ctx.transform(abi.points_of(source_down), T)  # pcl::transformPointCloud(*block2->pc_down, *pc_s_tran, Trans1_2)
        lib.write_pcd(F.output_point_cloud_path, moved)
    return res, source


if __name__ == "__main__":
    main()
