#!/usr/bin/env bash
# build_ref.sh — compile the reference's OWN hot-path function bodies into oracle/_ref/libmulls_ref.so.
#
# The MULLS-ICP path lives in header-only class templates whose headers include PCL, Eigen, glog, VTK viewers and the
# koide3 baselines (cregistration.hpp:15-55) — none of which exist in this image — so the headers cannot be included as
# they are.  This recipe instead reads the line ranges that hold the path's classes and member functions straight out
# of /root/reference at build time (into a temporary directory; nothing from the reference is copied into the
# repository), wraps them in the class shells of oracle/ref_driver.cpp and compiles them against oracle/ref_shim/shim.hpp,
# a stand-in for the handful of Eigen/PCL/boost/glog APIs those lines use.  Output: oracle/_ref/ only (git-ignored,
# travels to the GPU box).  Without /root/reference the script exits 0 and leaves any existing library in place.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${MULLS_REFERENCE:-/root/reference}"
INC="$REF/include/common"
if [ ! -f "$INC/cregistration.hpp" ]; then
	echo "build_ref: $REF not present, skipping"
	exit 0
fi
TMP="$(mktemp -d /tmp/mulls_ref.XXXXXX)"
trap 'rm -rf "$TMP"' EXIT

# extract <file> <first> <last> <expected substring of first line> <out>
extract() {
	local first_line src="$INC/$1"
	[[ "$1" == /* ]] && src="$1"
	first_line="$(sed -n "${2}p" "$src")"
	if [[ "$first_line" != *"$4"* ]]; then
		echo "build_ref: $1:$2 does not look like '$4' (reference revision changed?)" >&2
		exit 1
	fi
	sed -n "${2},${3}p" "$src" >>"$TMP/$5"
}

# utility.hpp: Vector6d/Matrix6d, centerpoint_t, bounds_t, enums, cloudblock_t, constraint_t, CloudUtility (bbox helpers)
extract utility.hpp 84 85 "typedef Eigen::Matrix<double, 6, 1> Vector6d" util_typedefs.inc
extract utility.hpp 92 157 "struct centerpoint_t" util_types.inc
extract utility.hpp 233 558 "struct cloudblock_t" util_types.inc
extract utility.hpp 561 590 "struct constraint_t" util_types.inc
extract utility.hpp 795 886 "template <typename PointT>" util_cloudutility.inc
# cfilter.hpp: grid_t, motion compensation, random down-sampling, box filter, the ground filter, pair intersection
extract cfilter.hpp 35 43 "struct idpair_t" cfilter_body.inc
extract cfilter.hpp 45 69 "struct grid_t" cfilter_body.inc
extract cfilter.hpp 83 165 "bool voxel_downsample" cfilter_body.inc
extract cfilter.hpp 470 549 "void apply_motion_compensation" cfilter_body.inc
extract cfilter.hpp 551 602 "bool xy_normal_balanced_downsample" cfilter_body.inc
extract cfilter.hpp 606 628 "bool random_downsample_pcl" cfilter_body.inc
extract cfilter.hpp 685 712 "bool random_downsample_pcl(typename pcl::PointCloud<PointT>::Ptr &cloud_in," cfilter_body.inc
extract cfilter.hpp 713 728 "bool random_downsample(const typename pcl::PointCloud<PointT>::Ptr &cloud_in," cfilter_body.inc
extract cfilter.hpp 806 832 "bool dist_filter(typename pcl::PointCloud<PointT>::Ptr &cloud_in_out," cfilter_body.inc
extract cfilter.hpp 834 872 "bool dist_filter(typename pcl::PointCloud<PointT>::Ptr &cloud_in_out," cfilter_body.inc
extract cfilter.hpp 914 929 "bool scanner_filter" cfilter_body.inc
extract cfilter.hpp 950 981 "bool bbx_filter" cfilter_body.inc
extract cfilter.hpp 1071 1181 "bool encode_stable_points" cfilter_body.inc
extract cfilter.hpp 1243 1312 "bool non_max_suppress(typename pcl::PointCloud<PointT>::Ptr &cloud_in," cfilter_body.inc
extract cfilter.hpp 1658 2036 "bool fast_ground_filter(const typename pcl::PointCloud<PointT>::Ptr &cloud_in," cfilter_body.inc
extract cfilter.hpp 2038 2056 "bool estimate_ground_normal_by_ransac" cfilter_body.inc
extract cfilter.hpp 2058 2290 "bool classify_nground_pts" cfilter_body.inc
extract cfilter.hpp 2295 2413 "bool extract_semantic_pts" cfilter_body.inc
extract cfilter.hpp 2416 2482 "void update_parameters_self_adaptive" cfilter_body.inc
# the semantic-mask filters of extract_semantic_pts (semantic_assisted: Semantic-KITTI labels in the curvature field)
extract cfilter.hpp 2487 2504 "bool filter_with_dynamic_object_mask_pre" cfilter_body.inc
extract cfilter.hpp 2508 2609 "void filter_with_semantic_mask" cfilter_body.inc
# pca.hpp: pca_feature_t and the neighbourhood PCA (get_pc_pca_feature x2, calculate_normal_inconsistency, get_pca_feature, assign_normal)
extract pca.hpp 23 54 "struct eigenvalue_t" pca_types.inc
extract pca.hpp 207 454 "// R - K neighborhood (without already built-kd tree)" pca_body.inc
# the ground filter's normal methods: pcl::NormalEstimationOMP wrappers + check_normal (1 / 2), the per-cell plane RANSAC (3)
extract pca.hpp 66 84 "bool get_normal_pcar" pca_normals.inc
extract pca.hpp 102 119 "bool get_normal_pcak" pca_normals.inc
extract pca.hpp 462 475 "void check_normal" pca_normals.inc
extract cprocessing.hpp 67 106 "bool plane_seg_ransac" cproc_body.inc
extract cfilter.hpp 2613 2655 "bool get_cloud_pair_intersection" cfilter_body.inc
# cregistration.hpp: the driver and every helper on the path
extract cregistration.hpp 1114 1440 "int mm_lls_icp(constraint_t &registration_cons" creg_body.inc
extract cregistration.hpp 1443 1681 "bool lls_icp_3dof_ground" creg_body.inc
extract cregistration.hpp 1685 1967 "void batch_transform_feature_points" creg_body.inc
extract cregistration.hpp 1976 2275 "bool pt2pt_lls_summation" creg_body.inc
extract cregistration.hpp 2278 2386 "bool ground_3dof_lls_tran_estimation" creg_body.inc
extract cregistration.hpp 2518 2722 "bool get_multi_metrics_lls_residual" creg_body.inc
extract cregistration.hpp 2740 2764 "bool construct_trans_a" creg_body.inc
extract cregistration.hpp 2795 2836 "bool get_quat_euler_jacobi" creg_body.inc
extract cregistration.hpp 2866 2922 "bool keep_less_source_pts" creg_body.inc
# local map manager: class declaration, update_local_map, dynamic removal, PCA refresh of the linear features
extract "$REF/include/pgo/map_manager.h" 19 52 "class MapManager" map_decl.inc
extract "$REF/src/map_manager.cpp" 18 140 "bool MapManager::update_local_map" map_body.inc
extract "$REF/src/map_manager.cpp" 149 218 "bool MapManager::map_based_dynamic_close_removal" map_body.inc
extract "$REF/src/map_manager.cpp" 222 256 "bool MapManager::map_scan_feature_pts_distance_removal" map_body.inc
extract "$REF/src/map_manager.cpp" 258 292 "bool MapManager::update_cloud_vectors" map_body.inc

mkdir -p "$HERE/_ref"
# same flags as the reference's Release build (CMakeLists.txt:43: -O3, no -march); no OpenMP: the only pragmas on the path
# are the 3-wide sections (results identical) and the racy parallel-for of apply_motion_compensation
g++ -O3 -ffp-contract=off -std=c++17 -fPIC -shared -w -I"$TMP" -I"$HERE" -I"$HERE/../include" \
	"$HERE/ref_driver.cpp" -o "$HERE/_ref/libmulls_ref.so"
echo "build_ref: built $HERE/_ref/libmulls_ref.so from $REF"
# end-to-end drop-in check of include/cregistration_hip.hpp against the same reference lines (needs libmulls_hip.so)
HIPLIB="$HERE/../mulls_amd/libmulls_hip.so"
if [ -f "$HIPLIB" ]; then
	g++ -O2 -ffp-contract=off -std=c++17 -w -I"$TMP" -I"$HERE" -I"$HERE/../include" "$HERE/adapter_check.cpp" \
		-o "$HERE/_ref/adapter_check" "$HIPLIB" -Wl,-rpath,'$ORIGIN/../../mulls_amd' -Wl,-rpath,/opt/rocm/lib
	echo "build_ref: built $HERE/_ref/adapter_check"
fi
