// ref_driver.cpp — class shells + C entry point around the reference's own MULLS-ICP source lines.
//
// TEST INFRASTRUCTURE (see oracle/build_ref.sh).  The *.inc files named below do not exist in the repository: they are
// cut out of /root/reference/include/common/{utility,cfilter,cregistration}.hpp at build time.  Everything in this file
// is glue: the two class shells the member functions live in, and marshalling between the C ABI structs of
// include/mulls_hip.h and lo::constraint_t.
#include <chrono>

#include "ref_shim/shim.hpp"

#include "mulls_hip.h"

#define max_(a, b) (((a) > (b)) ? (a) : (b))
#define min_(a, b) (((a) < (b)) ? (a) : (b))

using namespace std;

typedef pcl::PointXYZINormal Point_T;
typedef pcl::PointCloud<Point_T>::Ptr pcTPtr;
typedef pcl::PointCloud<Point_T> pcT;
typedef pcl::search::KdTree<Point_T>::Ptr pcTreePtr;
typedef pcl::search::KdTree<Point_T> pcTree;

#include "util_typedefs.inc" // Vector6d, Matrix6d

namespace lo
{
#include "util_types.inc" // centerpoint_t, bounds_t, DataType, ConstraintType, cloudblock_t, constraint_t

#include "util_cloudutility.inc" // template <typename PointT> class CloudUtility { public: ... bbox helpers
};

// pca.hpp: pca_feature_t, the member functions of PrincipleComponentAnalysis that do the neighbourhood PCA, and the two wrappers of
// pcl::NormalEstimationOMP with check_normal behind them (the ground filter's normal methods 1 / 2)
#include "pca_types.inc"
template <typename PointT>
class PrincipleComponentAnalysis
{
  public:
#include "pca_normals.inc" // get_normal_pcar (pca.hpp:66-84), get_normal_pcak (:102-119), check_normal (:462-475)
#include "pca_body.inc"
};

// cprocessing.hpp: the class shell around plane_seg_ransac (:67-106), the ground filter's normal method 3
template <typename PointT>
class CProceesing : public CloudUtility<PointT>
{
  public:
#include "cproc_body.inc"
};

template <typename PointT>
class CFilter : public CloudUtility<PointT>
{
  public:
	// (the semantic-mask filters of extract_semantic_pts — filter_with_dynamic_object_mask_pre cfilter.hpp:2487-2504, filter_with_semantic_mask :2508-2609 — are
	// among the cut lines since round 5)
#include "cfilter_body.inc" // (with estimate_ground_normal_by_ransac, cfilter.hpp:2038-2056)
};

template <typename PointT>
class CRegistration : public CloudUtility<PointT>
{
  public:
#include "creg_body.inc"
};

#include "map_decl.inc" // class MapManager { ... };
#include "map_body.inc" // MapManager::update_local_map, ::map_based_dynamic_close_removal, ::map_scan_feature_pts_distance_removal, ::update_cloud_vectors
} // namespace lo

namespace
{
void fill_cloud(const mulls_cloud &c, pcTPtr &out)
{
	out->points.resize(c.n);
	const unsigned char *p = (const unsigned char *)c.pts;
	for (uint32_t i = 0; i < c.n; i++)
		std::memcpy(&out->points[i], p + (size_t)i * c.stride, sizeof(Point_T));
}
} // namespace

static void fill_constraint(const mulls_pair *pair, lo::constraint_t &con);

// CFilter::fast_ground_filter, the reference's own lines (cfilter.hpp:1658-2036), same contract as mulls_ground_filter.  The
// fixed-number branch ends in pcl::RandomSample (time-seeded upstream): not comparable, refused here.
extern "C" int mulls_ref_ground_filter(const void *pts, uint32_t n, uint32_t stride, const mulls_ground_params *P, void *ground, uint32_t cap_ground,
										   void *ground_down, uint32_t cap_ground_down, void *unground, uint32_t cap_unground, uint32_t n_out[3])
{
	if (P->fixed_num_downsampling)
		return MULLS_E_UNSUPPORTED;
	pcTPtr in(new pcT), g(new pcT), gd(new pcT), u(new pcT), curb(new pcT);
	mulls_cloud c = {pts, n, stride};
	fill_cloud(c, in);
	lo::CFilter<Point_T> cf;
	cf.fast_ground_filter(in, g, gd, u, curb, P->min_grid_pt_num, P->grid_resolution, P->max_height_difference, P->neighbor_height_diff, P->max_ground_height,
						  P->ground_random_down_rate, P->ground_random_down_down_rate, P->nonground_random_down_rate, P->reliable_neighbor_grid_num_thre,
						  P->estimate_ground_normal_method, P->normal_estimation_radius, P->distance_weight_downsampling_method, P->standard_distance, false,
						  P->down_ground_fixed_num, false, P->intensity_thre, P->apply_grid_wise_outlier_filter != 0, P->outlier_std_scale);
	auto put = [](const pcTPtr &cl, void *dst, uint32_t cap) {
		const size_t k = std::min<size_t>(cl->points.size(), cap);
		for (size_t i = 0; i < k; i++)
			std::memcpy((unsigned char *)dst + i * sizeof(Point_T), &cl->points[i], sizeof(Point_T));
	};
	put(g, ground, cap_ground);
	put(gd, ground_down, cap_ground_down);
	put(u, unground, cap_unground);
	n_out[0] = (uint32_t)g->points.size();
	n_out[1] = (uint32_t)gd->points.size();
	n_out[2] = (uint32_t)u->points.size();
	return 0;
}

// CFilter::apply_motion_compensation(pc_in_out, Tran, s_ambigous_thre), the reference's own lines (cfilter.hpp:470-491), in place on 48-byte records
extern "C" int mulls_ref_motion_compensate(void *pts, uint32_t n, uint32_t stride, const double Tran[16], float s_ambigous_thre)
{
	pcTPtr pc(new pcT);
	mulls_cloud c = {pts, n, stride};
	fill_cloud(c, pc);
	Eigen::Matrix4d T;
	std::memcpy(T.data(), Tran, sizeof(double) * 16);
	lo::CFilter<Point_T> cf;
	cf.apply_motion_compensation(pc, T, s_ambigous_thre);
	for (uint32_t i = 0; i < n; i++)
		std::memcpy((unsigned char *)pts + (size_t)i * stride, &pc->points[i], sizeof(Point_T));
	return 0;
}

extern "C" int mulls_ref_icp_3dof_ground(const mulls_pair *pair, const mulls_params *P, mulls_result *R)
{
	lo::constraint_t con;
	fill_constraint(pair, con);
	Eigen::Matrix4d guess;
	std::memcpy(guess.data(), pair->init_guess, sizeof(double) * 16);
	pcl::registration::rejector_strict_flag() = P->rejector_strict != 0;
	lo::CRegistration<Point_T> creg;
	const bool ok = creg.lls_icp_3dof_ground(con, P->max_iter_num, P->dis_thre_unit, P->converge_translation, P->converge_rotation_d, P->dis_thre_min,
												P->dis_thre_update_rate, std::string(P->weight_strategy), guess, P->keep_less_source_points != 0,
												P->max_bearable_rotation_d);
	R->code = ok ? 1 : 0; // the reference casts its process code to bool: only "nonzero" is observable
	R->iters = -1;
	std::memcpy(R->T, con.Trans1_2.data(), sizeof(R->T));
	std::memcpy(R->info, con.information_matrix.data(), sizeof(R->info));
	R->sigma = con.sigma;
	R->confidence = con.confidence;
	return 0;
}

extern "C" int mulls_ref_icp_4dof_global(const mulls_pair *pair, float heading_step_d, const double station[3], int max_iter_num,
										 float dis_thre_unit, float converge_translation, float dis_thre_min, float dis_thre_update_rate,
										 mulls_result *R, int *success)
{
	lo::constraint_t con;
	fill_constraint(pair, con);
	con.block2->local_station.x = station[0];
	con.block2->local_station.y = station[1];
	con.block2->local_station.z = station[2];
	pcl::registration::rejector_strict_flag() = true; // the trials run with mulls_default_params
	lo::CRegistration<Point_T> creg;
	const bool ok = creg.mm_lls_icp_4dof_global(con, heading_step_d, max_iter_num, dis_thre_unit, converge_translation, converge_translation,
												   dis_thre_min, dis_thre_update_rate);
	*success = ok ? 1 : 0;
	R->code = ok ? 1 : 0;
	std::memcpy(R->T, con.Trans1_2.data(), sizeof(R->T));
	std::memcpy(R->info, con.information_matrix.data(), sizeof(R->info));
	R->sigma = con.sigma;
	R->confidence = con.confidence;
	return 0;
}

static void fill_constraint(const mulls_pair *pair, lo::constraint_t &con)
{
	lo::cloudblock_t &b1 = *con.block1, &b2 = *con.block2;
	fill_cloud(pair->tgt[MULLS_GROUND], b1.pc_ground);
	fill_cloud(pair->tgt[MULLS_PILLAR], b1.pc_pillar);
	fill_cloud(pair->tgt[MULLS_FACADE], b1.pc_facade);
	fill_cloud(pair->tgt[MULLS_BEAM], b1.pc_beam);
	fill_cloud(pair->tgt[MULLS_ROOF], b1.pc_roof);
	fill_cloud(pair->tgt[MULLS_VERTEX], b1.pc_vertex);
	// the ABI hands over whichever clouds clone_feature would pick; mirror them into both slots
	fill_cloud(pair->src[MULLS_GROUND], b2.pc_ground);
	fill_cloud(pair->src[MULLS_PILLAR], b2.pc_pillar);
	fill_cloud(pair->src[MULLS_FACADE], b2.pc_facade);
	fill_cloud(pair->src[MULLS_BEAM], b2.pc_beam);
	fill_cloud(pair->src[MULLS_ROOF], b2.pc_roof);
	fill_cloud(pair->src[MULLS_VERTEX], b2.pc_vertex);
	const bool has_down = pair->src_down[MULLS_GROUND].pts || pair->src_down[MULLS_PILLAR].pts || pair->src_down[MULLS_FACADE].pts;
	const mulls_cloud *down = has_down ? pair->src_down : pair->src;
	fill_cloud(down[MULLS_GROUND], b2.pc_ground_down);
	fill_cloud(down[MULLS_PILLAR], b2.pc_pillar_down);
	fill_cloud(down[MULLS_FACADE], b2.pc_facade_down);
	fill_cloud(down[MULLS_BEAM], b2.pc_beam_down);
	fill_cloud(down[MULLS_ROOF], b2.pc_roof_down);
	b1.local_bound.min_x = pair->tgt_bound[0];
	b1.local_bound.min_y = pair->tgt_bound[1];
	b1.local_bound.min_z = pair->tgt_bound[2];
	b1.local_bound.max_x = pair->tgt_bound[3];
	b1.local_bound.max_y = pair->tgt_bound[4];
	b1.local_bound.max_z = pair->tgt_bound[5];
}

extern "C" int mulls_ref_icp(const mulls_pair *pair, const mulls_params *P, mulls_result *R)
{
	lo::constraint_t con;
	fill_constraint(pair, con);
	Eigen::Matrix4d guess;
	std::memcpy(guess.data(), pair->init_guess, sizeof(double) * 16);

	pcl::registration::rejector_strict_flag() = P->rejector_strict != 0;
	lo::CRegistration<Point_T> creg;
	const int code = creg.mm_lls_icp(con, P->max_iter_num, P->dis_thre_unit, P->converge_translation, P->converge_rotation_d, P->dis_thre_min,
									 P->dis_thre_update_rate, std::string(P->used_feature_type), std::string(P->weight_strategy),
									 P->z_xy_balanced_ratio, P->pt2pt_residual_window, P->pt2pl_residual_window, P->pt2li_residual_window, guess,
									 P->apply_intersection_filter != 0, P->apply_motion_undistortion != 0, P->normal_shooting_on != 0,
									 P->normal_bearing, P->use_more_points != 0, P->keep_less_source_points != 0, P->sigma_thre,
									 P->min_neccessary_corr_ratio, P->max_bearable_rotation_d);
	std::memset(R->ncorr, 0, sizeof(R->ncorr));
	R->code = code;
	R->iters = -1; // not observable through the reference interface
	std::memcpy(R->T, con.Trans1_2.data(), sizeof(R->T));
	std::memcpy(R->info, con.information_matrix.data(), sizeof(R->info));
	R->sigma = con.sigma;
	R->confidence = con.confidence;
	R->trace_len = 0;
	return 0;
}


// MapManager::update_local_map on plain clouds (same signature as mulls_oracle_map_update).  block1->tree_* are
// prepared the way mm_lls_icp leaves them (cregistration.hpp:1209-1232): whole class clouds or their bbx_filter'ed clones.
extern "C" int mulls_ref_map_update(const mulls_cloud map_in[6], const double map_pose[16], const mulls_cloud frame_down[6],
									 const double frame_pose[16], const mulls_map_params *P, void *const map_out[6], uint32_t map_out_n[6],
									 void *const frame_out[6], uint32_t frame_out_n[6], mulls_map_report *rep)
{
	lo::cloudblock_Ptr map(new lo::cloudblock_t()), frame(new lo::cloudblock_t());
	pcTPtr *mc[6] = {&map->pc_ground, &map->pc_pillar, &map->pc_facade, &map->pc_beam, &map->pc_roof, &map->pc_vertex};
	pcTPtr *fc[6] = {&frame->pc_ground_down, &frame->pc_pillar_down, &frame->pc_facade_down, &frame->pc_beam_down, &frame->pc_roof_down,
					 &frame->pc_vertex};
	pcTreePtr *trees[6] = {&map->tree_ground, &map->tree_pillar, &map->tree_facade, &map->tree_beam, &map->tree_roof, &map->tree_vertex};
	for (int c = 0; c < 6; c++)
	{
		fill_cloud(map_in[c], *mc[c]);
		fill_cloud(frame_down[c], *fc[c]);
	}
	std::memcpy(map->pose_lo.data(), map_pose, sizeof(double) * 16);
	std::memcpy(frame->pose_lo.data(), frame_pose, sizeof(double) * 16);
	map->feature_point_num = (int)(map->pc_ground->points.size() + map->pc_facade->points.size() + map->pc_roof->points.size() +
								   map->pc_pillar->points.size() + map->pc_beam->points.size());
	bool trees_ready = P->tree_mode != 0;
	lo::CFilter<Point_T> cf;
	for (int c = 0; c < 6 && trees_ready; c++)
		if (P->tree_used[c] == '1')
		{
			pcTPtr clone(new pcT());
			clone->points = (*mc[c])->points;
			if (P->tree_mode == 2)
			{
				lo::bounds_t b;
				b.min_x = P->tree_box[0], b.min_y = P->tree_box[1], b.min_z = P->tree_box[2];
				b.max_x = P->tree_box[3], b.max_y = P->tree_box[4], b.max_z = P->tree_box[5];
				cf.bbx_filter(clone, b);
			}
			if (clone->points.size() > 0)
				(*trees[c])->setInputCloud(clone);
		}
	// an empty kd-tree makes nearestKSearch undefined upstream (the restatement defines "leave the cloud alone"): this
	// entry point refuses configurations in which a class the removal visits has no tree
	const bool removal = P->map_based_dynamic_removal_on && trees_ready;
	if (removal)
		for (int c : {1, 2, 3})
			if (P->used_feature_type[c] == '1' && !(*trees[c])->cloud && (*fc[c])->points.size() > 10)
				return -1;
	lo::MapManager mm;
	rep->dynamic_removal_ran = (removal && map->feature_point_num > P->max_num_pts / 5) ? 1 : 0;
	mm.update_local_map(map, frame, P->local_map_radius, P->max_num_pts, P->kept_vertex_num, P->last_frame_reliable_radius, removal,
						std::string(P->used_feature_type, 6), P->dynamic_removal_center_radius, P->dynamic_dist_thre_min, P->dynamic_dist_thre_max,
						P->near_dist_thre, P->recalculate_feature_on != 0);
	for (int c = 0; c < 6; c++)
	{
		map_out_n[c] = (uint32_t)(*mc[c])->points.size();
		frame_out_n[c] = (uint32_t)(*fc[c])->points.size();
		rep->n[c] = map_out_n[c];
		rep->frame_n[c] = frame_out_n[c];
		if (map_out && map_out[c])
			std::memcpy(map_out[c], (*mc[c])->points.data(), sizeof(Point_T) * map_out_n[c]);
		if (frame_out && frame_out[c])
			std::memcpy(frame_out[c], (*fc[c])->points.data(), sizeof(Point_T) * frame_out_n[c]);
	}
	rep->feature_point_num = map->feature_point_num;
	rep->local_bound[0] = map->local_bound.min_x, rep->local_bound[1] = map->local_bound.min_y, rep->local_bound[2] = map->local_bound.min_z;
	rep->local_bound[3] = map->local_bound.max_x, rep->local_bound[4] = map->local_bound.max_y, rep->local_bound[5] = map->local_bound.max_z;
	rep->bound[0] = map->bound.min_x, rep->bound[1] = map->bound.min_y, rep->bound[2] = map->bound.min_z;
	rep->bound[3] = map->bound.max_x, rep->bound[4] = map->bound.max_y, rep->bound[5] = map->bound.max_z;
	return 0;
}

// CFilter::classify_nground_pts, the reference's own lines (cfilter.hpp:2058-2290), same contract as mulls_oracle_classify_nground.
// The fixed-number down-samplings go through pcl::RandomSample (time-seeded upstream, a fixed seed in the shim): only sizes compare.
extern "C" int mulls_ref_classify_nground(const void *pts, uint32_t n, uint32_t stride, const mulls_classify_params *P, void *const out[MULLS_CL_COUNT],
										  const uint32_t cap[MULLS_CL_COUNT], uint32_t n_out[MULLS_CL_COUNT], void *in_after, uint32_t *n_in_after)
{
	mulls_cloud c;
	c.pts = pts, c.n = n, c.stride = stride;
	pcTPtr in(new pcT()), o[MULLS_CL_COUNT];
	fill_cloud(c, in);
	for (int k = 0; k < MULLS_CL_COUNT; k++)
		o[k].reset(new pcT());
	lo::CFilter<Point_T> cf;
	cf.classify_nground_pts(in, o[MULLS_CL_PILLAR], o[MULLS_CL_BEAM], o[MULLS_CL_FACADE], o[MULLS_CL_ROOF], o[MULLS_CL_PILLAR_DOWN], o[MULLS_CL_BEAM_DOWN],
							o[MULLS_CL_FACADE_DOWN], o[MULLS_CL_ROOF_DOWN], o[MULLS_CL_VERTEX], P->neighbor_searching_radius, P->neighbor_k, P->neigh_k_min,
							P->pca_down_rate, P->edge_thre, P->planar_thre, P->edge_thre_down, P->planar_thre_down, P->extract_vertex_points_method,
							P->curvature_thre, P->vertex_curvature_non_max_radius, P->linear_vertical_sin_high_thre, P->linear_vertical_sin_low_thre,
							P->planar_vertical_sin_high_thre, P->planar_vertical_sin_low_thre, P->fixed_num_downsampling != 0, P->pillar_down_fixed_num,
							P->facade_down_fixed_num, P->beam_down_fixed_num, P->roof_down_fixed_num, P->unground_down_fixed_num, P->beam_height_max,
							P->roof_height_min, P->feature_pts_ratio_guess, P->sharpen_with_nms != 0, P->use_distance_adaptive_pca != 0);
	for (int k = 0; k < MULLS_CL_COUNT; k++)
	{
		n_out[k] = (uint32_t)o[k]->points.size();
		const size_t m = std::min<size_t>(o[k]->points.size(), cap[k]);
		if (m && out[k])
			std::memcpy(out[k], o[k]->points.data(), m * sizeof(Point_T));
	}
	if (n_in_after)
		*n_in_after = (uint32_t)in->points.size();
	if (in_after && in->points.size())
		std::memcpy(in_after, in->points.data(), in->points.size() * sizeof(Point_T));
	return 0;
}

// CFilter::scanner_filter, the reference's own lines (cfilter.hpp:914-929)
extern "C" int mulls_ref_scanner_filter(const void *pts, uint32_t n, uint32_t stride, float self_radius, float ghost_radius, float z_min_thre_ghost,
										float z_min_thre_global, void *out, uint32_t cap, uint32_t *n_out)
{
	mulls_cloud c;
	c.pts = pts, c.n = n, c.stride = stride;
	pcTPtr in(new pcT());
	fill_cloud(c, in);
	lo::CFilter<Point_T> cf;
	cf.scanner_filter(in, self_radius, ghost_radius, z_min_thre_ghost, z_min_thre_global);
	*n_out = (uint32_t)in->points.size();
	const size_t m = std::min<size_t>(in->points.size(), cap);
	if (m && out)
		std::memcpy(out, in->points.data(), m * sizeof(Point_T));
	return 0;
}

// CFilter::dist_filter(cloud, xy_dist_min, xy_dist_max), the reference's own lines (cfilter.hpp:806-832)
extern "C" int mulls_ref_dist_filter(const void *pts, uint32_t n, uint32_t stride, double xy_dist_min, double xy_dist_max, void *out, uint32_t cap, uint32_t *n_out)
{
	mulls_cloud c;
	c.pts = pts, c.n = n, c.stride = stride;
	pcTPtr in(new pcT());
	fill_cloud(c, in);
	lo::CFilter<Point_T> cf;
	cf.dist_filter(in, xy_dist_min, xy_dist_max);
	*n_out = (uint32_t)in->points.size();
	const size_t m = std::min<size_t>(in->points.size(), cap);
	if (m && out)
		std::memcpy(out, in->points.data(), m * sizeof(Point_T));
	return 0;
}

// CFilter::voxel_downsample, the reference's own lines (cfilter.hpp:83-160)
extern "C" int mulls_ref_voxel_downsample(const void *pts, uint32_t n, uint32_t stride, float voxel_size, void *out, uint32_t cap, uint32_t *n_out)
{
	mulls_cloud c;
	c.pts = pts, c.n = n, c.stride = stride;
	pcTPtr in(new pcT()), down(new pcT());
	fill_cloud(c, in);
	lo::CFilter<Point_T> cf;
	cf.voxel_downsample(in, down, voxel_size);
	*n_out = (uint32_t)down->points.size();
	const size_t m = std::min<size_t>(down->points.size(), cap);
	if (m && out)
		std::memcpy(out, down->points.data(), m * sizeof(Point_T));
	return 0;
}

// The frame front end as test/mulls_slam.cpp:359-365 runs it, the reference's own lines: CFilter::dist_filter on pc_raw (when asked for), then
// the member CFilter::extract_semantic_pts (cfilter.hpp:2295-2413) on a cloudblock_t — the composition the stage-by-stage oracle chain
// (pyoracle.extract_features) and mulls_extract_features restate.  approx_scanner_height / underground_thre are recovered from the two
// heights the ABI carries (z_min = -h - 4, z_min_min = -h + u).  out[] in enum mulls_extract_cloud's order.
extern "C" int mulls_ref_extract_semantic_pts(const void *scan, uint32_t n, uint32_t stride, const mulls_extract_params *X, void *const out[MULLS_EX_COUNT],
											  const uint32_t cap[MULLS_EX_COUNT], uint32_t n_out[MULLS_EX_COUNT], int32_t rates_after[2])
{
	mulls_cloud c;
	c.pts = scan, c.n = n, c.stride = stride;
	lo::cloudblock_Ptr blk(new lo::cloudblock_t());
	fill_cloud(c, blk->pc_raw);
	lo::CFilter<Point_T> cf;
	if (X->apply_dist_filter)
		cf.dist_filter(blk->pc_raw, X->min_dist_used, X->max_dist_used);
	const mulls_ground_params &G = X->ground;
	const mulls_classify_params &K = X->classify;
	int gdr = G.ground_random_down_rate, ndr = G.nonground_random_down_rate;
	const float approx_scanner_height = -(X->z_min + 4.0f), underground_thre = X->z_min_min + approx_scanner_height;
	cf.extract_semantic_pts(blk, X->vf_downsample_resolution, G.grid_resolution, G.max_height_difference, G.neighbor_height_diff, G.max_ground_height, gdr, ndr,
							K.neighbor_searching_radius, K.neighbor_k, K.edge_thre, K.planar_thre, K.curvature_thre, K.edge_thre_down, K.planar_thre_down,
							K.use_distance_adaptive_pca != 0, G.distance_weight_downsampling_method, G.standard_distance, G.estimate_ground_normal_method, G.normal_estimation_radius, false,
							X->apply_scanner_filter != 0, false, K.extract_vertex_points_method, G.min_grid_pt_num, G.reliable_neighbor_grid_num_thre,
							G.ground_random_down_down_rate, K.neigh_k_min, K.pca_down_rate, G.intensity_thre, K.linear_vertical_sin_high_thre,
							K.linear_vertical_sin_low_thre, K.planar_vertical_sin_high_thre, K.planar_vertical_sin_low_thre, K.sharpen_with_nms != 0,
							G.fixed_num_downsampling != 0, G.down_ground_fixed_num, K.pillar_down_fixed_num, K.facade_down_fixed_num, K.beam_down_fixed_num,
							K.roof_down_fixed_num, K.unground_down_fixed_num, K.beam_height_max, K.roof_height_min, approx_scanner_height, underground_thre,
							K.feature_pts_ratio_guess, false, false, 0.0f, 0.0f);
	rates_after[0] = gdr, rates_after[1] = ndr;
	pcTPtr all[MULLS_EX_COUNT] = {blk->pc_raw,	blk->pc_ground, blk->pc_ground_down, blk->pc_unground,	  blk->pc_pillar,	  blk->pc_beam,	  blk->pc_facade,
								  blk->pc_roof, blk->pc_pillar_down, blk->pc_beam_down,	blk->pc_facade_down, blk->pc_roof_down, blk->pc_vertex, blk->pc_down};
	for (int k = 0; k < MULLS_EX_COUNT; k++)
	{
		n_out[k] = (uint32_t)all[k]->points.size();
		const size_t m = std::min<size_t>(all[k]->points.size(), cap[k]);
		if (m && out[k])
			std::memcpy(out[k], all[k]->points.data(), m * sizeof(Point_T));
	}
	return 0;
}
