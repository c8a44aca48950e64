// cregistration_hip.hpp — drop-in bridge from MULLS' CRegistration<PointT>::mm_lls_icp() to libmulls_hip.so.
//
// The reference has no plugin/FFI layer: mm_lls_icp is a public member of the header-only class template
// lo::CRegistration<PointT> (include/common/cregistration.hpp:1114-1440), called positionally from
// test/mulls_reg.cpp:194 and test/mulls_slam.cpp:477,560,642,679.  This header keeps that exact signature (argument
// order, types, defaults) as a free function template, lo::hip::mm_lls_icp<PointT>(), so that the one-line binding a
// maintainer adds at cregistration.hpp:1125
//
//     #ifdef MULLS_USE_HIP
//         return lo::hip::mm_lls_icp<PointT>(registration_cons, max_iter_num, dis_thre_unit, converge_translation,
//                 converge_rotation_d, dis_thre_min, dis_thre_update_rate, used_feature_type, weight_strategy,
//                 z_xy_balanced_ratio, pt2pt_residual_window, pt2pl_residual_window, pt2li_residual_window,
//                 initial_guess, apply_intersection_filter, apply_motion_undistortion_while_registration,
//                 normal_shooting_on, normal_bearing, use_more_points, keep_less_source_points, sigma_thre,
//                 min_neccessary_corr_ratio, max_bearable_rotation_d);
//     #endif
//
// leaves every call site, cloudblock_t and constraint_t (include/common/utility.hpp:233-590) untouched.
// See INTEGRATION.md for the build flags and for what is and is not reproduced (kd-tree side effect, logging).
//
// Requirements on the including translation unit: MULLS' utility.hpp (lo::constraint_t, lo::cloudblock_t, Matrix6d),
// PCL point types and Eigen must already be visible — exactly what cregistration.hpp includes before line 1114.
#ifndef MULLS_CREGISTRATION_HIP_HPP
#define MULLS_CREGISTRATION_HIP_HPP

#include <cstring>
#include <map>
#include <stdexcept>
#include <string>

#include "mulls_hip.h"

namespace lo
{
namespace hip
{

// One library context per host thread (the reference path is single-threaded and stateless, SURVEY §8b "Threading").
inline mulls_ctx *thread_context(int device = 0)
{
	struct Holder
	{
		mulls_ctx *ctx = nullptr;
		~Holder()
		{
			if (ctx)
				mulls_destroy(ctx);
		}
	};
	static thread_local Holder holder;
	if (!holder.ctx)
	{
		const int rc = mulls_create(device, &holder.ctx);
		if (rc != MULLS_OK)
			throw std::runtime_error("mulls_create failed (" + std::to_string(rc) + "): no usable gfx950 device; there is no CPU fallback");
		// constraint_t carries no per-class cloud sizes: stage only the clouds the registration reads (mulls_result.nsrc0 / ntgt0 of the others read 0)
		(void)mulls_set_option(holder.ctx, MULLS_OPT_LEAN_STAGING, 1.0);
	}
	return holder.ctx;
}

template <typename CloudPtr>
inline mulls_cloud borrow(const CloudPtr &cloud)
{
	typedef typename std::remove_reference<decltype(cloud->points[0])>::type PointT;
	static_assert(sizeof(PointT) >= MULLS_POINT_BYTES, "expects pcl::PointXYZINormal-compatible records (48 bytes)");
	mulls_cloud c;
	c.pts = cloud->points.empty() ? nullptr : static_cast<const void *>(cloud->points.data());
	c.n = static_cast<uint32_t>(cloud->points.size());
	c.stride = static_cast<uint32_t>(sizeof(PointT));
	return c;
}

// When set (default), block1->tree_{ground,pillar,beam,facade,roof,vertex} are rebuilt on the cropped target clouds
// like cregistration.hpp:1209-1232 does, because MapManager::map_based_dynamic_close_removal (src/map_manager.cpp:187-243)
// queries them on the next frame.  The registration itself never uses a CPU kd-tree.  Benchmarks switch it off.
// The switch only matters while the process keeps NO device-resident local map: as soon as lo::hip::update_local_map() has
// created a mirror, the trees' one consumer runs on the device against that mirror and no registration builds them — neither
// the scan-to-map one (its block1 has the mirror) nor the scan-to-scan one against the previous frame's block.
inline bool &build_cpu_trees()
{
	static bool on = true;
	return on;
}

// ---- device-resident local map -------------------------------------------------------------------------
// A cloudblock_t that acts as the scan-to-map target can be given a mirror in HBM: lo::hip::update_local_map() (below)
// creates it on first use and keeps it up to date; lo::hip::mm_lls_icp() then takes block1's six class clouds straight
// from the device (no upload of the 20 k - 1 M point target per frame) and remembers which part of them the registration
// indexed, which is what MapManager::map_based_dynamic_close_removal looks at on the next frame (block1->tree_*).
struct MapMirror
{
	mulls_map *map = nullptr;
	int tree_mode = 0; // 0: no registration against this map yet; 1: whole clouds; 2: cropped to tree_box
	char tree_used[8] = {'0', '0', '0', '0', '0', '0', 0, 0};
	double tree_box[6] = {0, 0, 0, 0, 0, 0};
};
inline std::map<const cloudblock_t *, MapMirror> &map_mirrors()
{
	struct Holder
	{
		std::map<const cloudblock_t *, MapMirror> m;
		~Holder() // after thread_context()'s holder was constructed, so destroyed before it
		{
			for (auto &kv : m)
				if (kv.second.map)
					mulls_map_destroy(thread_context(), kv.second.map);
		}
	};
	thread_context();
	static thread_local Holder holder;
	return holder.m;
}
// When set (default), update_local_map copies the updated class clouds back into local_map->pc_* so that every other
// consumer of the block (viewer, sub-map cloning, loop closure) sees what the reference would have left there.
inline bool &sync_host_map()
{
	static bool on = true;
	return on;
}
// forget the mirror of a block whose host clouds were changed behind the bridge's back (it is rebuilt from them on next use)
inline void invalidate_map_mirror(const cloudblock_t *block)
{
	auto &m = map_mirrors();
	auto it = m.find(block);
	if (it != m.end())
	{
		mulls_map_destroy(thread_context(), it->second.map);
		m.erase(it);
	}
}
inline MapMirror &attach_local_map(const cloudblock_Ptr &block)
{
	auto &m = map_mirrors();
	auto it = m.find(block.get());
	if (it != m.end())
		return it->second;
	mulls_ctx *ctx = thread_context();
	MapMirror mir;
	int rc = mulls_map_create(ctx, &mir.map);
	if (rc == MULLS_OK)
	{
		const mulls_cloud clouds[6] = {borrow(block->pc_ground), borrow(block->pc_pillar), borrow(block->pc_facade),
									   borrow(block->pc_beam),	 borrow(block->pc_roof),   borrow(block->pc_vertex)};
		rc = mulls_map_set(ctx, mir.map, clouds, block->pose_lo.data());
	}
	if (rc != MULLS_OK)
		throw std::runtime_error(std::string("local map mirror failed (") + std::to_string(rc) + "): " + mulls_last_error(ctx));
	return m.emplace(block.get(), mir).first->second;
}

template <typename PointT>
int mm_lls_icp(constraint_t &registration_cons, // cblock_1 (target point cloud), cblock_2 (source point cloud)
			   int max_iter_num = 20, float dis_thre_unit = 1.5, float converge_translation = 0.002, float converge_rotation_d = 0.01,
			   float dis_thre_min = 0.4, float dis_thre_update_rate = 1.1, std::string used_feature_type = "111110",
			   std::string weight_strategy = "1101", float z_xy_balanced_ratio = 1.0, float pt2pt_residual_window = 0.1,
			   float pt2pl_residual_window = 0.1, float pt2li_residual_window = 0.1,
			   Eigen::Matrix4d initial_guess = Eigen::Matrix4d::Identity(), bool apply_intersection_filter = true,
			   bool apply_motion_undistortion_while_registration = false, bool normal_shooting_on = false, float normal_bearing = 45.0,
			   bool use_more_points = false, bool keep_less_source_points = false, float sigma_thre = 0.5,
			   float min_neccessary_corr_ratio = 0.03, float max_bearable_rotation_d = 45.0)
{
	mulls_ctx *ctx = thread_context();
	cloudblock_t &b1 = *registration_cons.block1; // target
	cloudblock_t &b2 = *registration_cons.block2; // source

	mulls_pair pair;
	std::memset(&pair, 0, sizeof(pair));
	// class index == used_feature_type character index (cregistration.hpp:1196-1201)
	pair.tgt[MULLS_GROUND] = borrow(b1.pc_ground);
	pair.tgt[MULLS_PILLAR] = borrow(b1.pc_pillar);
	pair.tgt[MULLS_FACADE] = borrow(b1.pc_facade);
	pair.tgt[MULLS_BEAM] = borrow(b1.pc_beam);
	pair.tgt[MULLS_ROOF] = borrow(b1.pc_roof);
	pair.tgt[MULLS_VERTEX] = borrow(b1.pc_vertex);
	MapMirror *mirror = nullptr;
	{
		auto &mm = map_mirrors();
		auto it = mm.find(&b1);
		if (it != mm.end())
		{
			mirror = &it->second; // block1 is a device-resident local map: its clouds are already in HBM
			// keep_less_source_points thins the target's ground / facade clouds before their kd-trees are built (cregistration.hpp:1190-1193,
			// time-seeded): the trees the next update_local_map's dynamic removal searches would hold that random subset, which the mirror's
			// tree state (crop box only) cannot describe — refused rather than silently searching a different point set
			if (keep_less_source_points)
				throw std::runtime_error("lo::hip::mm_lls_icp: keep_less_source_points with a device-resident block1 (local map mirror) is not supported; "
										 "call lo::hip::invalidate_map_mirror(block1) first or pass keep_less_source_points = false");
			for (int c = 0; c < 6; c++)
				if (mulls_map_cloud(ctx, mirror->map, c, &pair.tgt[c]) != MULLS_OK)
					throw std::runtime_error(std::string("mulls_map_cloud: ") + mulls_last_error(ctx));
		}
	}
	// clone_feature(..., !use_more_points): down-sampled features unless use_more_points; vertex always pc_vertex (utility.hpp:524-550)
	pair.src[MULLS_GROUND] = borrow(use_more_points ? b2.pc_ground : b2.pc_ground_down);
	pair.src[MULLS_PILLAR] = borrow(use_more_points ? b2.pc_pillar : b2.pc_pillar_down);
	pair.src[MULLS_FACADE] = borrow(use_more_points ? b2.pc_facade : b2.pc_facade_down);
	pair.src[MULLS_BEAM] = borrow(use_more_points ? b2.pc_beam : b2.pc_beam_down);
	pair.src[MULLS_ROOF] = borrow(use_more_points ? b2.pc_roof : b2.pc_roof_down);
	pair.src[MULLS_VERTEX] = borrow(b2.pc_vertex);
	// batch_apply_motion_compensation always starts from the _down clouds (cregistration.hpp:1251-1253)
	pair.src_down[MULLS_GROUND] = borrow(b2.pc_ground_down);
	pair.src_down[MULLS_PILLAR] = borrow(b2.pc_pillar_down);
	pair.src_down[MULLS_FACADE] = borrow(b2.pc_facade_down);
	pair.src_down[MULLS_BEAM] = borrow(b2.pc_beam_down);
	pair.src_down[MULLS_ROOF] = borrow(b2.pc_roof_down);
	pair.src_down[MULLS_VERTEX] = borrow(b2.pc_vertex);
	pair.tgt_bound[0] = b1.local_bound.min_x;
	pair.tgt_bound[1] = b1.local_bound.min_y;
	pair.tgt_bound[2] = b1.local_bound.min_z;
	pair.tgt_bound[3] = b1.local_bound.max_x;
	pair.tgt_bound[4] = b1.local_bound.max_y;
	pair.tgt_bound[5] = b1.local_bound.max_z;
	std::memcpy(pair.init_guess, initial_guess.data(), sizeof(pair.init_guess)); // Eigen::Matrix4d is column-major

	mulls_params P;
	mulls_default_params(&P);
	P.max_iter_num = max_iter_num;
	P.dis_thre_unit = dis_thre_unit;
	P.converge_translation = converge_translation;
	P.converge_rotation_d = converge_rotation_d;
	P.dis_thre_min = dis_thre_min;
	P.dis_thre_update_rate = dis_thre_update_rate;
	std::memset(P.used_feature_type, 0, sizeof(P.used_feature_type));
	std::memset(P.weight_strategy, 0, sizeof(P.weight_strategy));
	std::strncpy(P.used_feature_type, used_feature_type.c_str(), sizeof(P.used_feature_type) - 1);
	std::strncpy(P.weight_strategy, weight_strategy.c_str(), sizeof(P.weight_strategy) - 1);
	P.z_xy_balanced_ratio = z_xy_balanced_ratio;
	P.pt2pt_residual_window = pt2pt_residual_window;
	P.pt2pl_residual_window = pt2pl_residual_window;
	P.pt2li_residual_window = pt2li_residual_window;
	P.apply_intersection_filter = apply_intersection_filter;
	P.apply_motion_undistortion = apply_motion_undistortion_while_registration;
	P.normal_shooting_on = normal_shooting_on;
	P.use_more_points = use_more_points;
	P.normal_bearing = normal_bearing;
	P.keep_less_source_points = keep_less_source_points;
	P.faithful = 1;
	P.sigma_thre = sigma_thre;
	P.min_neccessary_corr_ratio = min_neccessary_corr_ratio;
	P.max_bearable_rotation_d = max_bearable_rotation_d;

	mulls_result R;
	std::memset(&R, 0, sizeof(R));
	const int rc = mulls_icp(ctx, &pair, &P, &R);
	if (rc != MULLS_OK) // infrastructure failure (no device, HIP error): the reference has no channel for it
		throw std::runtime_error(std::string("mulls_icp failed (") + std::to_string(rc) + "): " + mulls_last_error(ctx));

	// constraint_t outputs (cregistration.hpp:1405-1420)
	std::memcpy(registration_cons.Trans1_2.data(), R.T, sizeof(R.T));
	std::memcpy(registration_cons.information_matrix.data(), R.info, sizeof(R.info));
	registration_cons.sigma = R.sigma;
	registration_cons.confidence = R.confidence;

	if (mirror && max_iter_num > 0)
	{
		// what block1->tree_* hold from here on (cregistration.hpp:1209-1232), for the next update_local_map
		mirror->tree_mode = R.cropped ? 2 : 1;
		std::memcpy(mirror->tree_box, R.crop_box, sizeof(mirror->tree_box));
		for (int c = 0; c < 6; c++)
			mirror->tree_used[c] = (used_feature_type[c] == '1' && R.ntgt0[c] > 0) ? '1' : '0';
	}
	if (build_cpu_trees() && !mirror && map_mirrors().empty())
	{
		// kd-tree side effect of cregistration.hpp:1209-1232: trees over the intersection-filtered target clouds
		typedef typename pcl::PointCloud<PointT> Cloud;
		struct Item
		{
			int cls;
			typename Cloud::Ptr src;
			pcTreePtr tree;
		};
		const Item items[6] = {{MULLS_GROUND, b1.pc_ground, b1.tree_ground}, {MULLS_PILLAR, b1.pc_pillar, b1.tree_pillar},
							   {MULLS_FACADE, b1.pc_facade, b1.tree_facade}, {MULLS_BEAM, b1.pc_beam, b1.tree_beam},
							   {MULLS_ROOF, b1.pc_roof, b1.tree_roof},		 {MULLS_VERTEX, b1.pc_vertex, b1.tree_vertex}};
		for (const Item &it : items)
		{
			if (used_feature_type[it.cls] != '1')
				continue;
			typename Cloud::Ptr cropped(new Cloud);
			if (R.cropped)
			{
				for (const PointT &p : it.src->points) // CFilter::bbx_filter, strict inequalities (cfilter.hpp:959-961)
					if (p.x > R.crop_box[0] && p.x < R.crop_box[3] && p.y > R.crop_box[1] && p.y < R.crop_box[4] && p.z > R.crop_box[2] &&
						p.z < R.crop_box[5])
						cropped->points.push_back(p);
			}
			else
				*cropped = *it.src;
			if (cropped->points.size() > 0)
				it.tree->setInputCloud(cropped);
		}
	}
	return R.code;
}

// lls_icp_3dof_ground (cregistration.hpp:1443-1445), verbatim signature.  Like the reference it returns the process code cast
// to bool (true for 1, -1 and -2 alike) and only writes registration_cons.Trans1_2.
template <typename PointT>
bool lls_icp_3dof_ground(constraint_t &registration_cons, int max_iter_num = 20, float dis_thre_unit = 1.5, float converge_translation = 0.002,
						 float converge_rotation_d = 0.01, float dis_thre_min = 0.4, float dis_thre_update_rate = 1.1,
						 std::string weight_strategy = "1111", Eigen::Matrix4d initial_guess = Eigen::Matrix4d::Identity(),
						 bool keep_less_source_points = false, float max_bearable_rotation_d = 10.0)
{
	mulls_ctx *ctx = thread_context();
	mulls_pair pair;
	std::memset(&pair, 0, sizeof(pair));
	pair.tgt[MULLS_GROUND] = borrow(registration_cons.block1->pc_ground);
	pair.src[MULLS_GROUND] = borrow(registration_cons.block2->pc_ground_down);
	std::memcpy(pair.init_guess, initial_guess.data(), sizeof(pair.init_guess));
	mulls_params P;
	mulls_default_params(&P);
	P.max_iter_num = max_iter_num;
	P.dis_thre_unit = dis_thre_unit;
	P.converge_translation = converge_translation;
	P.converge_rotation_d = converge_rotation_d;
	P.dis_thre_min = dis_thre_min;
	P.dis_thre_update_rate = dis_thre_update_rate;
	std::memset(P.weight_strategy, 0, sizeof(P.weight_strategy));
	std::strncpy(P.weight_strategy, weight_strategy.c_str(), sizeof(P.weight_strategy) - 1);
	P.keep_less_source_points = keep_less_source_points;
	P.max_bearable_rotation_d = max_bearable_rotation_d;
	mulls_result R;
	std::memset(&R, 0, sizeof(R));
	const int rc = mulls_icp_3dof_ground(ctx, &pair, &P, &R);
	if (rc != MULLS_OK)
		throw std::runtime_error(std::string("mulls_icp_3dof_ground failed (") + std::to_string(rc) + "): " + mulls_last_error(ctx));
	std::memcpy(registration_cons.Trans1_2.data(), R.T, sizeof(R.T));
	return R.code != 0;
}

// mm_lls_icp_4dof_global (cregistration.hpp:1584-1586), verbatim signature: all heading trials run as one batch.
template <typename PointT>
bool mm_lls_icp_4dof_global(constraint_t &registration_con, float heading_step_d, int max_iter_num = 20, float dis_thre_unit = 1.5,
							float converge_translation = 0.005, float converge_rotation_d = 0.05, float dis_thre_min = 0.5,
							float dis_thre_update_rate = 1.05, float max_bearable_rotation_d = 15.0)
{
	mulls_ctx *ctx = thread_context();
	cloudblock_t &b1 = *registration_con.block1;
	cloudblock_t &b2 = *registration_con.block2;
	mulls_pair pair;
	std::memset(&pair, 0, sizeof(pair));
	pair.tgt[MULLS_GROUND] = borrow(b1.pc_ground);
	pair.tgt[MULLS_PILLAR] = borrow(b1.pc_pillar);
	pair.tgt[MULLS_FACADE] = borrow(b1.pc_facade);
	pair.tgt[MULLS_BEAM] = borrow(b1.pc_beam);
	pair.tgt[MULLS_ROOF] = borrow(b1.pc_roof);
	pair.tgt[MULLS_VERTEX] = borrow(b1.pc_vertex);
	pair.src[MULLS_GROUND] = borrow(b2.pc_ground_down);
	pair.src[MULLS_PILLAR] = borrow(b2.pc_pillar_down);
	pair.src[MULLS_FACADE] = borrow(b2.pc_facade_down);
	pair.src[MULLS_BEAM] = borrow(b2.pc_beam_down);
	pair.src[MULLS_ROOF] = borrow(b2.pc_roof_down);
	pair.src[MULLS_VERTEX] = borrow(b2.pc_vertex);
	pair.tgt_bound[0] = b1.local_bound.min_x;
	pair.tgt_bound[1] = b1.local_bound.min_y;
	pair.tgt_bound[2] = b1.local_bound.min_z;
	pair.tgt_bound[3] = b1.local_bound.max_x;
	pair.tgt_bound[4] = b1.local_bound.max_y;
	pair.tgt_bound[5] = b1.local_bound.max_z;
	const double station[3] = {b2.local_station.x, b2.local_station.y, b2.local_station.z};
	mulls_result R;
	std::memset(&R, 0, sizeof(R));
	int success = 0;
	float best_heading = 0.0f;
	const int rc = mulls_icp_4dof_global(ctx, &pair, heading_step_d, station, max_iter_num, dis_thre_unit, converge_translation, converge_rotation_d,
										 dis_thre_min, dis_thre_update_rate, max_bearable_rotation_d, &R, &success, &best_heading);
	if (rc != MULLS_OK)
		throw std::runtime_error(std::string("mulls_icp_4dof_global failed (") + std::to_string(rc) + "): " + mulls_last_error(ctx));
	if (success == 1) // the reference only touches registration_con when a succeeded trial also took the best score (:1645-1657)
	{
		std::memcpy(registration_con.Trans1_2.data(), R.T, sizeof(R.T));
		std::memcpy(registration_con.information_matrix.data(), R.info, sizeof(R.info));
		registration_con.sigma = R.sigma;
		registration_con.confidence = R.confidence;
	}
	return success != 0;
}

// MapManager::update_local_map (include/pgo/map_manager.h:21-31, src/map_manager.cpp:18-140), verbatim signature.  The binding
// a maintainer adds is one early return at the top of the member function:
//     #ifdef MULLS_USE_HIP
//         return lo::hip::update_local_map(local_map, last_target_cblock, local_map_radius, max_num_pts, ...);
//     #endif
// Differences: pcl::RandomSample is seeded with time(NULL) upstream, here with `map_rng_seed()`; with recalculate_feature_on the
// refreshed directions agree with pcl::PCA's up to Eigen's float eigen-solver accuracy and the direction's sign.
inline uint64_t &map_rng_seed()
{
	static uint64_t seed = 0;
	return seed;
}
inline bool update_local_map(cloudblock_Ptr local_map, cloudblock_Ptr last_target_cblock, float local_map_radius = 80, int max_num_pts = 20000,
							 int kept_vertex_num = 800, float last_frame_reliable_radius = 60, bool map_based_dynamic_removal_on = false,
							 std::string used_feature_type = "111110", float dynamic_removal_center_radius = 30.0,
							 float dynamic_dist_thre_min = 0.3, float dynamic_dist_thre_max = 3.0, float near_dist_thre = 0.03,
							 bool recalculate_feature_on = false)
{
	mulls_ctx *ctx = thread_context();
	MapMirror &mir = attach_local_map(local_map);
	cloudblock_t &f = *last_target_cblock;
	const mulls_cloud frame[6] = {borrow(f.pc_ground_down), borrow(f.pc_pillar_down), borrow(f.pc_facade_down),
								  borrow(f.pc_beam_down),	borrow(f.pc_roof_down),	  borrow(f.pc_vertex)};
	mulls_map_params P;
	mulls_map_default_params(&P);
	P.local_map_radius = local_map_radius;
	P.max_num_pts = max_num_pts;
	P.kept_vertex_num = kept_vertex_num;
	P.last_frame_reliable_radius = last_frame_reliable_radius;
	P.map_based_dynamic_removal_on = map_based_dynamic_removal_on;
	std::memset(P.used_feature_type, 0, sizeof(P.used_feature_type));
	std::strncpy(P.used_feature_type, used_feature_type.c_str(), sizeof(P.used_feature_type) - 1);
	P.dynamic_removal_center_radius = dynamic_removal_center_radius;
	P.dynamic_dist_thre_min = dynamic_dist_thre_min;
	P.dynamic_dist_thre_max = dynamic_dist_thre_max;
	P.near_dist_thre = near_dist_thre;
	P.recalculate_feature_on = recalculate_feature_on;
	P.rng_seed = map_rng_seed()++;
	P.tree_mode = mir.tree_mode;
	std::memcpy(P.tree_used, mir.tree_used, sizeof(P.tree_used));
	std::memcpy(P.tree_box, mir.tree_box, sizeof(P.tree_box));
	mulls_map_report rep;
	const int rc = mulls_map_update(ctx, mir.map, frame, f.pose_lo.data(), &P, &rep);
	if (rc != MULLS_OK)
		throw std::runtime_error(std::string("mulls_map_update failed (") + std::to_string(rc) + "): " + mulls_last_error(ctx));

	typedef typename std::remove_reference<decltype(*f.pc_ground_down)>::type Cloud;
	auto fetch = [&](int (*fn)(mulls_ctx *, const mulls_map *, int, void *, uint32_t, uint32_t *), int cls, Cloud &dst, uint32_t n) {
		dst.points.resize(n);
		uint32_t got = 0;
		if (fn(ctx, mir.map, cls, n ? static_cast<void *>(dst.points.data()) : nullptr, n, &got) != MULLS_OK || got != n)
			throw std::runtime_error(std::string("local map download: ") + mulls_last_error(ctx));
	};
	static_assert(sizeof(f.pc_ground_down->points[0]) == MULLS_POINT_BYTES, "download expects 48-byte point records");
	// last_target_cblock->pc_*_down were moved to the map frame and filtered in place by the reference (:32, :37-47)
	Cloud *fd[5] = {f.pc_ground_down.get(), f.pc_pillar_down.get(), f.pc_facade_down.get(), f.pc_beam_down.get(), f.pc_roof_down.get()};
	for (int c = 0; c < 5; c++)
		fetch(mulls_map_frame_download, c, *fd[c], rep.frame_n[c]);
	cloudblock_t &m = *local_map;
	if (sync_host_map())
	{
		Cloud *mc[6] = {m.pc_ground.get(), m.pc_pillar.get(), m.pc_facade.get(), m.pc_beam.get(), m.pc_roof.get(), m.pc_vertex.get()};
		for (int c = 0; c < 6; c++)
			fetch(mulls_map_download, c, *mc[c], rep.n[c]);
	}
	m.pose_lo = f.pose_lo; // :58-59
	m.pose_gt = f.pose_gt;
	m.local_bound.min_x = rep.local_bound[0], m.local_bound.min_y = rep.local_bound[1], m.local_bound.min_z = rep.local_bound[2];
	m.local_bound.max_x = rep.local_bound[3], m.local_bound.max_y = rep.local_bound[4], m.local_bound.max_z = rep.local_bound[5];
	m.bound.min_x = rep.bound[0], m.bound.min_y = rep.bound[1], m.bound.min_z = rep.bound[2];
	m.bound.max_x = rep.bound[3], m.bound.max_y = rep.bound[4], m.bound.max_z = rep.bound[5];
	m.feature_point_num = rep.feature_point_num; // :128-130
	m.free_tree();								 // :132-133
	m.free_raw_cloud();
	mir.tree_mode = 0; // the kd-trees are gone until the next registration against this map
	return true;
}

// ------------------------------------------------------------------------------------------------------------------
// Feature extraction: CFilter<PointT>::fast_ground_filter (include/common/cfilter.hpp:1658-2036) and
// CFilter<PointT>::classify_nground_pts (:2058-2290), verbatim signatures.  The binding a maintainer adds is one early return at the top of
// each member function:
//     #ifdef MULLS_USE_HIP
//         return lo::hip::fast_ground_filter<PointT>(cloud_in, cloud_ground, cloud_ground_down, cloud_unground, cloud_curb, min_grid_pt_num, ...);
//     #endif
// Differences: what PCL computes inside estimate_ground_normal_method 1 - 3 (NormalEstimationOMP, SACSegmentation) is the library's restatement of
// it (include/mulls_hip.h: mulls_ground_params), fast_ground_filter's cloud_in
// keeps its contents (upstream writes normals and data[3] into it; the clouds handed out carry them), the fixed-number selections are seeded with
// `feature_rng_seed()` (upstream: pcl::RandomSample seeded with time(NULL)), and what pcl::PCA / Eigen compute inside classify_nground_pts is
// the library's restatement.
inline uint64_t &feature_rng_seed()
{
	static uint64_t seed = 0;
	return seed;
}
template <typename PointT, typename CloudPtr>
inline void take_cloud(CloudPtr &dst, const std::vector<unsigned char> &raw, uint32_t n, bool append)
{
	static_assert(sizeof(PointT) == MULLS_POINT_BYTES, "48-byte point records expected");
	const size_t before = append ? dst->points.size() : 0;
	dst->points.resize(before + n);
	if (n)
		std::memcpy(static_cast<void *>(&dst->points[before]), raw.data(), (size_t)n * MULLS_POINT_BYTES);
}
template <typename PointT>
inline bool fast_ground_filter(const typename pcl::PointCloud<PointT>::Ptr &cloud_in, typename pcl::PointCloud<PointT>::Ptr &cloud_ground,
							   typename pcl::PointCloud<PointT>::Ptr &cloud_ground_down, typename pcl::PointCloud<PointT>::Ptr &cloud_unground,
							   typename pcl::PointCloud<PointT>::Ptr &cloud_curb, int min_grid_pt_num, float grid_resolution, float max_height_difference,
							   float neighbor_height_diff, float max_ground_height, int ground_random_down_rate, int ground_random_down_down_rate,
							   int nonground_random_down_rate, int reliable_neighbor_grid_num_thre, int estimate_ground_normal_method,
							   float normal_estimation_radius, int distance_weight_downsampling_method, float standard_distance,
							   bool fixed_num_downsampling = false, int down_ground_fixed_num = 1000, bool detect_curb_or_not = false,
							   float intensity_thre = FLT_MAX, bool apply_grid_wise_outlier_filter = false, float outlier_std_scale = 3.0)
{
	(void)cloud_curb, (void)detect_curb_or_not; // curb detection is commented out upstream (:1990-2010)
	mulls_ctx *ctx = thread_context();
	mulls_ground_params P;
	mulls_ground_default_params(&P);
	P.min_grid_pt_num = min_grid_pt_num;
	P.grid_resolution = grid_resolution;
	P.max_height_difference = max_height_difference;
	P.neighbor_height_diff = neighbor_height_diff;
	P.max_ground_height = max_ground_height;
	P.ground_random_down_rate = ground_random_down_rate;
	P.ground_random_down_down_rate = ground_random_down_down_rate;
	P.nonground_random_down_rate = nonground_random_down_rate;
	P.reliable_neighbor_grid_num_thre = reliable_neighbor_grid_num_thre;
	P.estimate_ground_normal_method = estimate_ground_normal_method;
	P.normal_estimation_radius = normal_estimation_radius;
	P.distance_weight_downsampling_method = distance_weight_downsampling_method;
	P.standard_distance = standard_distance;
	P.fixed_num_downsampling = fixed_num_downsampling;
	P.down_ground_fixed_num = down_ground_fixed_num;
	P.intensity_thre = intensity_thre;
	P.apply_grid_wise_outlier_filter = apply_grid_wise_outlier_filter;
	P.outlier_std_scale = outlier_std_scale;
	P.rng_seed = feature_rng_seed()++;
	const mulls_cloud in = borrow(cloud_in);
	std::vector<unsigned char> g((size_t)in.n * MULLS_POINT_BYTES), gd((size_t)in.n * MULLS_POINT_BYTES), u((size_t)in.n * MULLS_POINT_BYTES);
	uint32_t n_out[3] = {0, 0, 0};
	const int rc = mulls_ground_filter(ctx, in.pts, in.n, in.stride, &P, g.data(), in.n, gd.data(), in.n, u.data(), in.n, n_out);
	if (rc != MULLS_OK)
		throw std::runtime_error(std::string("mulls_ground_filter failed (") + std::to_string(rc) + "): " + mulls_last_error(ctx));
	take_cloud<PointT>(cloud_ground, g, n_out[0], true); // upstream pushes back into the clouds it is given
	take_cloud<PointT>(cloud_ground_down, gd, n_out[1], true);
	take_cloud<PointT>(cloud_unground, u, n_out[2], true);
	return 1;
}
template <typename PointT>
inline bool classify_nground_pts(typename pcl::PointCloud<PointT>::Ptr &cloud_in, typename pcl::PointCloud<PointT>::Ptr &cloud_pillar,
								 typename pcl::PointCloud<PointT>::Ptr &cloud_beam, typename pcl::PointCloud<PointT>::Ptr &cloud_facade,
								 typename pcl::PointCloud<PointT>::Ptr &cloud_roof, typename pcl::PointCloud<PointT>::Ptr &cloud_pillar_down,
								 typename pcl::PointCloud<PointT>::Ptr &cloud_beam_down, typename pcl::PointCloud<PointT>::Ptr &cloud_facade_down,
								 typename pcl::PointCloud<PointT>::Ptr &cloud_roof_down, typename pcl::PointCloud<PointT>::Ptr &cloud_vertex,
								 float neighbor_searching_radius, int neighbor_k, int neigh_k_min, int pca_down_rate, float edge_thre, float planar_thre,
								 float edge_thre_down, float planar_thre_down, int extract_vertex_points_method, float curvature_thre,
								 float vertex_curvature_non_max_radius, float linear_vertical_sin_high_thre, float linear_vertical_sin_low_thre,
								 float planar_vertical_sin_high_thre, float planar_vertical_sin_low_thre, bool fixed_num_downsampling = false,
								 int pillar_down_fixed_num = 200, int facade_down_fixed_num = 800, int beam_down_fixed_num = 200, int roof_down_fixed_num = 100,
								 int unground_down_fixed_num = 20000, float beam_height_max = FLT_MAX, float roof_height_min = -FLT_MAX,
								 float feature_pts_ratio_guess = 0.3, bool sharpen_with_nms = true, bool use_distance_adaptive_pca = false)
{
	mulls_ctx *ctx = thread_context();
	mulls_classify_params P;
	mulls_classify_default_params(&P);
	P.neighbor_searching_radius = neighbor_searching_radius;
	P.neighbor_k = neighbor_k;
	P.neigh_k_min = neigh_k_min;
	P.pca_down_rate = pca_down_rate;
	P.edge_thre = edge_thre, P.planar_thre = planar_thre, P.edge_thre_down = edge_thre_down, P.planar_thre_down = planar_thre_down;
	P.extract_vertex_points_method = extract_vertex_points_method;
	P.curvature_thre = curvature_thre;
	P.vertex_curvature_non_max_radius = vertex_curvature_non_max_radius;
	P.linear_vertical_sin_high_thre = linear_vertical_sin_high_thre, P.linear_vertical_sin_low_thre = linear_vertical_sin_low_thre;
	P.planar_vertical_sin_high_thre = planar_vertical_sin_high_thre, P.planar_vertical_sin_low_thre = planar_vertical_sin_low_thre;
	P.fixed_num_downsampling = fixed_num_downsampling;
	P.sharpen_with_nms = sharpen_with_nms;
	P.use_distance_adaptive_pca = use_distance_adaptive_pca;
	P.pillar_down_fixed_num = pillar_down_fixed_num, P.facade_down_fixed_num = facade_down_fixed_num, P.beam_down_fixed_num = beam_down_fixed_num;
	P.roof_down_fixed_num = roof_down_fixed_num, P.unground_down_fixed_num = unground_down_fixed_num;
	P.beam_height_max = beam_height_max, P.roof_height_min = roof_height_min, P.feature_pts_ratio_guess = feature_pts_ratio_guess;
	P.rng_seed = feature_rng_seed()++;
	const mulls_cloud in = borrow(cloud_in);
	std::vector<unsigned char> raw[MULLS_CL_COUNT];
	void *out[MULLS_CL_COUNT];
	uint32_t cap[MULLS_CL_COUNT], n_out[MULLS_CL_COUNT];
	for (int k = 0; k < MULLS_CL_COUNT; k++)
	{
		raw[k].resize((size_t)in.n * MULLS_POINT_BYTES);
		out[k] = raw[k].data();
		cap[k] = in.n;
	}
	std::vector<unsigned char> after((size_t)in.n * MULLS_POINT_BYTES);
	uint32_t n_after = 0;
	const int rc = mulls_classify_nground(ctx, in.pts, in.n, in.stride, &P, out, cap, n_out, after.data(), &n_after);
	if (rc != MULLS_OK)
		throw std::runtime_error(std::string("mulls_classify_nground failed (") + std::to_string(rc) + "): " + mulls_last_error(ctx));
	take_cloud<PointT>(cloud_in, after, n_after, false); // upstream works on cloud_in itself: thinned first (fixed numbers), normals written into it
	typename pcl::PointCloud<PointT>::Ptr *dst[MULLS_CL_COUNT] = {&cloud_pillar,	  &cloud_beam,		  &cloud_facade,	  &cloud_roof,	&cloud_pillar_down,
																	  &cloud_beam_down, &cloud_facade_down, &cloud_roof_down, &cloud_vertex};
	for (int k = 0; k < MULLS_CL_COUNT; k++)
		take_cloud<PointT>(*dst[k], raw[k], n_out[k], true);
	return 1;
}

// CFilter<PointT>::apply_motion_compensation (include/common/cfilter.hpp:470-491), verbatim signature: the frame's points moved by their time-stamp
// fraction of Tran, in place (mulls_motion_compensate: one upload, one kernel, one download).  test/mulls_slam.cpp:705 binds it with one early return.
template <typename PointT>
inline void apply_motion_compensation(typename pcl::PointCloud<PointT>::Ptr pc_in_out, Eigen::Matrix4d &Tran, float s_ambigous_thre = 0.000)
{
	static_assert(sizeof(PointT) == MULLS_POINT_BYTES, "in-place compensation expects 48-byte pcl::PointXYZINormal records");
	if (pc_in_out->points.empty())
		return;
	mulls_ctx *ctx = thread_context();
	const int rc = mulls_motion_compensate(ctx, pc_in_out->points.data(), static_cast<uint32_t>(pc_in_out->points.size()), MULLS_POINT_BYTES, Tran.data(), s_ambigous_thre);
	if (rc != MULLS_OK)
		throw std::runtime_error(std::string("mulls_motion_compensate failed (") + std::to_string(rc) + "): " + mulls_last_error(ctx));
}
// CFilter<PointT>::batch_apply_motion_compensation, the in-place overload (cfilter.hpp:519-531), verbatim signature (the argument NAMES are upstream's:
// test/mulls_slam.cpp:706-710 passes facade third and beam fourth, which makes no difference — every cloud gets the same treatment)
template <typename PointT>
inline void batch_apply_motion_compensation(typename pcl::PointCloud<PointT>::Ptr pc_ground, typename pcl::PointCloud<PointT>::Ptr pc_pillar,
											typename pcl::PointCloud<PointT>::Ptr pc_beam, typename pcl::PointCloud<PointT>::Ptr pc_facade,
											typename pcl::PointCloud<PointT>::Ptr pc_roof, typename pcl::PointCloud<PointT>::Ptr pc_vertex, Eigen::Matrix4d &Tran,
											bool undistort_keypoints_or_not = false)
{
	apply_motion_compensation<PointT>(pc_ground, Tran);
	apply_motion_compensation<PointT>(pc_pillar, Tran);
	apply_motion_compensation<PointT>(pc_beam, Tran);
	apply_motion_compensation<PointT>(pc_facade, Tran);
	apply_motion_compensation<PointT>(pc_roof, Tran);
	if (undistort_keypoints_or_not)
		apply_motion_compensation<PointT>(pc_vertex, Tran);
}

// CFilter<PointT>::voxel_downsample (include/common/cfilter.hpp:83-160), verbatim signature
template <typename PointT>
inline bool voxel_downsample(const typename pcl::PointCloud<PointT>::Ptr &cloud_in, typename pcl::PointCloud<PointT>::Ptr &cloud_out, float voxel_size)
{
	if (voxel_size < 0.001)
	{
		cloud_out = cloud_in; // :90-97
		return false;
	}
	mulls_ctx *ctx = thread_context();
	const mulls_cloud in = borrow(cloud_in);
	std::vector<unsigned char> raw((size_t)in.n * MULLS_POINT_BYTES);
	uint32_t n_out = 0;
	const int rc = mulls_voxel_downsample(ctx, in.pts, in.n, in.stride, voxel_size, raw.data(), in.n, &n_out);
	if (rc != MULLS_OK)
		throw std::runtime_error(std::string("mulls_voxel_downsample failed (") + std::to_string(rc) + "): " + mulls_last_error(ctx));
	take_cloud<PointT>(cloud_out, raw, n_out, true); // `cloud_out->push_back(...)`: appended (:147)
	return 1;
}

// CFilter<PointT>::extract_semantic_pts (include/common/cfilter.hpp:2295-2413), verbatim signature: the whole chain in one device call
// (mulls_extract_features).  The binding is one early return at the top of the member function:
//     #ifdef MULLS_USE_HIP
//         return lo::hip::extract_semantic_pts<PointT>(in_block, vf_downsample_resolution, gf_grid_resolution, ...);
//     #endif
// semantic_assisted (deprecated upstream, off in every shipped configuration; round 5): the pre-filter filter_with_dynamic_object_mask_pre (cfilter.hpp:2487-2504 —
// Semantic-KITTI labels in the curvature field: moving objects, labels >= 250, and outliers, label 1, leave pc_raw) runs here on the host, in place on pc_raw as
// upstream does, INSTEAD of the scanner filter (upstream's `else if`, :2331-2345); the post-filter filter_with_semantic_mask(in_block) is called with its default mask
// "000000" (:2396), with which it touches nothing.  apply_roi_filtering is dead code upstream ("#if 0") and ignored here too.  use_adpative_parameters runs upstream's update (:2416-2444) on the host; where upstream would divide by zero (no facade and
// no pillar point left) this throws instead.
// One side effect is not reproduced: upstream's ground filter writes (0,0,1) normals and data[3] heights into the points of pc_down (= pc_raw)
// it classifies; here pc_raw / pc_down / pc_sketch keep the scan's records (the clouds handed out carry those values).
template <typename PointT>
inline bool extract_semantic_pts(cloudblock_Ptr in_block, float vf_downsample_resolution, float gf_grid_resolution, float gf_max_grid_height_diff,
								 float gf_neighbor_height_diff, float gf_max_ground_height, int &gf_down_rate_ground, int &gf_downsample_rate_nonground,
								 float pca_neighbor_radius, int pca_neighbor_k, float edge_thre, float planar_thre, float curvature_thre, float edge_thre_down,
								 float planar_thre_down, bool use_distance_adaptive_pca = false, int distance_inverse_sampling_method = 0,
								 float standard_distance = 15.0, int estimate_ground_normal_method = 3, float normal_estimation_radius = 2.0,
								 bool use_adpative_parameters = false, bool apply_scanner_filter = false, bool extract_curb_or_not = false,
								 int extract_vertex_points_method = 2, int gf_grid_pt_num_thre = 8, int gf_reliable_neighbor_grid_thre = 0,
								 int gf_down_down_rate_ground = 2, int pca_neighbor_k_min = 8, int pca_down_rate = 1, float intensity_thre = FLT_MAX,
								 float linear_vertical_sin_high_thre = 0.94, float linear_vertical_sin_low_thre = 0.17, float planar_vertical_sin_high_thre = 0.98,
								 float planar_vertical_sin_low_thre = 0.34, bool sharpen_with_nms_on = true, bool fixed_num_downsampling = false,
								 int ground_down_fixed_num = 500, int pillar_down_fixed_num = 200, int facade_down_fixed_num = 800, int beam_down_fixed_num = 200,
								 int roof_down_fixed_num = 200, int unground_down_fixed_num = 20000, float beam_height_max = FLT_MAX, float roof_height_min = 0.0,
								 float approx_scanner_height = 2.0, float underground_thre = -7.0, float feature_pts_ratio_guess = 0.3, bool semantic_assisted = false,
								 bool apply_roi_filtering = false, float roi_min_y = 0.0, float roi_max_y = 0.0)
{
	(void)extract_curb_or_not, (void)apply_roi_filtering, (void)roi_min_y, (void)roi_max_y;
	if (semantic_assisted)
	{
		// filter_with_dynamic_object_mask_pre(in_block->pc_raw), cfilter.hpp:2487-2504: the float comparisons as written there
		typename pcl::PointCloud<PointT>::Ptr kept(new pcl::PointCloud<PointT>());
		for (size_t i = 0; i < in_block->pc_raw->points.size(); i++)
			if (in_block->pc_raw->points[i].curvature < 250 && in_block->pc_raw->points[i].curvature != 1)
				kept->points.push_back(in_block->pc_raw->points[i]);
		kept->points.swap(in_block->pc_raw->points);
	}
	const bool scanner_stage = apply_scanner_filter && !semantic_assisted; // upstream: `if (semantic_assisted) ... else if (apply_scanner_filter) scanner_filter(...)`
	mulls_ctx *ctx = thread_context();
	mulls_extract_params X;
	mulls_extract_default_params(&X);
	mulls_ground_params &G = X.ground;
	G.min_grid_pt_num = gf_grid_pt_num_thre;
	G.grid_resolution = gf_grid_resolution;
	G.max_height_difference = gf_max_grid_height_diff;
	G.neighbor_height_diff = gf_neighbor_height_diff;
	G.max_ground_height = gf_max_ground_height;
	G.ground_random_down_rate = gf_down_rate_ground;
	G.ground_random_down_down_rate = gf_down_down_rate_ground;
	G.nonground_random_down_rate = gf_downsample_rate_nonground;
	G.reliable_neighbor_grid_num_thre = gf_reliable_neighbor_grid_thre;
	G.estimate_ground_normal_method = estimate_ground_normal_method;
	G.normal_estimation_radius = normal_estimation_radius;
	G.distance_weight_downsampling_method = distance_inverse_sampling_method;
	G.standard_distance = standard_distance;
	G.fixed_num_downsampling = fixed_num_downsampling;
	G.down_ground_fixed_num = ground_down_fixed_num;
	G.intensity_thre = intensity_thre;
	G.apply_grid_wise_outlier_filter = apply_scanner_filter; // :2361
	G.outlier_std_scale = 3.0f;
	G.rng_seed = feature_rng_seed()++;
	mulls_classify_params &K = X.classify;
	K.neighbor_searching_radius = pca_neighbor_radius;
	K.neighbor_k = pca_neighbor_k;
	K.neigh_k_min = pca_neighbor_k_min;
	K.pca_down_rate = pca_down_rate;
	K.edge_thre = edge_thre, K.planar_thre = planar_thre, K.edge_thre_down = edge_thre_down, K.planar_thre_down = planar_thre_down;
	K.extract_vertex_points_method = extract_vertex_points_method;
	K.curvature_thre = curvature_thre;
	K.vertex_curvature_non_max_radius = 1.5 * pca_neighbor_radius; // :2365
	K.linear_vertical_sin_high_thre = linear_vertical_sin_high_thre, K.linear_vertical_sin_low_thre = linear_vertical_sin_low_thre;
	K.planar_vertical_sin_high_thre = planar_vertical_sin_high_thre, K.planar_vertical_sin_low_thre = planar_vertical_sin_low_thre;
	K.fixed_num_downsampling = fixed_num_downsampling;
	K.sharpen_with_nms = sharpen_with_nms_on;
	K.use_distance_adaptive_pca = use_distance_adaptive_pca;
	K.pillar_down_fixed_num = pillar_down_fixed_num, K.facade_down_fixed_num = facade_down_fixed_num, K.beam_down_fixed_num = beam_down_fixed_num;
	K.roof_down_fixed_num = roof_down_fixed_num, K.unground_down_fixed_num = unground_down_fixed_num;
	K.beam_height_max = beam_height_max, K.roof_height_min = roof_height_min, K.feature_pts_ratio_guess = feature_pts_ratio_guess;
	K.rng_seed = feature_rng_seed()++;
	X.apply_scanner_filter = scanner_stage;
	X.self_ring_radius = 1.75f; // :2340-2343
	X.ghost_radius = 20.0f;
	X.z_min = -approx_scanner_height - 4.0;
	X.z_min_min = -approx_scanner_height + underground_thre;
	X.vf_downsample_resolution = vf_downsample_resolution;
	const bool voxels = !(vf_downsample_resolution < 0.001);

	cloudblock_t &b = *in_block;
	const mulls_cloud in = borrow(b.pc_raw);
	std::vector<unsigned char> raw[MULLS_EX_COUNT];
	void *out[MULLS_EX_COUNT];
	uint32_t cap[MULLS_EX_COUNT], n_out[MULLS_EX_COUNT];
	for (int k = 0; k < MULLS_EX_COUNT; k++)
	{
		const bool want = k == MULLS_EX_RAW ? scanner_stage : k == MULLS_EX_DOWN ? voxels : true; // without the scanner filter pc_raw stays what it is, without voxels pc_down is pc_raw
		raw[k].resize(want ? (size_t)in.n * MULLS_POINT_BYTES : 0);
		out[k] = want ? raw[k].data() : nullptr;
		cap[k] = want ? in.n : 0;
	}
	const int rc = mulls_extract_features(ctx, in.pts, in.n, in.stride, &X, out, cap, n_out);
	if (rc != MULLS_OK)
		throw std::runtime_error(std::string("mulls_extract_features failed (") + std::to_string(rc) + "): " + mulls_last_error(ctx));
	if (scanner_stage)
		take_cloud<PointT>(b.pc_raw, raw[MULLS_EX_RAW], n_out[MULLS_EX_RAW], false); // scanner_filter works on pc_raw itself
	if (voxels)
		take_cloud<PointT>(b.pc_down, raw[MULLS_EX_DOWN], n_out[MULLS_EX_DOWN], true); // `cloud_out->push_back(...)` (:147)
	else
		b.pc_down = b.pc_raw; // voxel_downsample below 0.001 m: `cloud_out = cloud_in` (:92)
	{
		// random_downsample(pc_down, pc_sketch, size / 1024 + 1) (:2351, :713-728)
		const int ratio = (int)(b.pc_down->points.size() / 1024 + 1);
		if (ratio > 1)
		{
			b.pc_sketch->points.clear();
			for (size_t i = 0; i < b.pc_down->points.size(); i++)
				if ((int)i % ratio == 0)
					b.pc_sketch->points.push_back(b.pc_down->points[i]);
		}
	}
	take_cloud<PointT>(b.pc_ground, raw[MULLS_EX_GROUND], n_out[MULLS_EX_GROUND], true);
	take_cloud<PointT>(b.pc_ground_down, raw[MULLS_EX_GROUND_DOWN], n_out[MULLS_EX_GROUND_DOWN], true);
	take_cloud<PointT>(b.pc_unground, raw[MULLS_EX_UNGROUND], n_out[MULLS_EX_UNGROUND], false); // the filter appends to an empty cloud, the classification works on it in place
	typename pcl::PointCloud<PointT>::Ptr *dst[9] = {&b.pc_pillar, &b.pc_beam, &b.pc_facade, &b.pc_roof, &b.pc_pillar_down, &b.pc_beam_down, &b.pc_facade_down, &b.pc_roof_down, &b.pc_vertex};
	for (int k = 0; k < 9; k++)
		take_cloud<PointT>(*dst[k], raw[MULLS_EX_PILLAR + k], n_out[MULLS_EX_PILLAR + k], true);
	b.down_feature_point_num = b.pc_ground_down->points.size() + b.pc_pillar_down->points.size() + b.pc_beam_down->points.size() + b.pc_facade_down->points.size() +
							   b.pc_roof_down->points.size() + b.pc_vertex->points.size(); // :2397-2398
	if (use_adpative_parameters)
	{
		// update_parameters_self_adaptive (:2416-2444): the one live rule lowers the non-ground down-sampling rate when few facade / pillar points came out
		const int non_ground_num_min_expected = 200;
		const int non_ground_feature_num = (int)b.pc_facade_down->points.size() + (int)b.pc_pillar_down->points.size();
		if (non_ground_feature_num < non_ground_num_min_expected)
		{
			if (non_ground_feature_num == 0)
				throw std::runtime_error("lo::hip::extract_semantic_pts: adaptive parameters with no facade / pillar point (upstream divides by zero here)");
			gf_downsample_rate_nonground = std::max(1, gf_downsample_rate_nonground - non_ground_num_min_expected / non_ground_feature_num);
		}
	}
	return true;
}

} // namespace hip
} // namespace lo

#endif // MULLS_CREGISTRATION_HIP_HPP
