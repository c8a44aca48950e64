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

template <typename PointT>
class CFilter : public CloudUtility<PointT>
{
  public:
#include "cfilter_body.inc"
};

template <typename PointT>
class CRegistration : public CloudUtility<PointT>
{
  public:
#include "creg_body.inc"
};
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

extern "C" int mulls_ref_icp_3dof_ground(const mulls_pair *pair, const mulls_params *P, mulls_result *R)
{
	lo::constraint_t con;
	fill_constraint(pair, con);
	Eigen::Matrix4d guess;
	std::memcpy(guess.data(), pair->init_guess, sizeof(double) * 16);
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
