// adapter_check.cpp — end-to-end drop-in check of include/cregistration_hip.hpp (TEST INFRASTRUCTURE, built by
// oracle/build_ref.sh next to libmulls_ref.so).
//
// One process, one lo::constraint_t: the reference's own lo::CRegistration<Point_T>::mm_lls_icp (its source lines,
// compiled against oracle/ref_shim) and the adapter lo::hip::mm_lls_icp<Point_T> (-> libmulls_hip.so -> MI355X) are
// called with the positional arguments of test/mulls_slam.cpp:642-648 / test/mulls_reg.cpp:194-195, and their
// constraint_t outputs are printed side by side as JSON for tests/test_gpu_adapter.py to compare.
//
// usage: adapter_check <dump file written by the test> <kitti|reg|variants|map>
#include <chrono>
#include <cstdio>

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

#include "util_typedefs.inc"

namespace lo
{
#include "util_types.inc"
#include "util_cloudutility.inc"
};
// pca.hpp / cprocessing.hpp / cfilter.hpp: the same class shells as oracle/ref_driver.cpp
#include "pca_types.inc"
template <typename PointT>
class PrincipleComponentAnalysis
{
  public:
#include "pca_normals.inc"
#include "pca_body.inc"
};
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
#include "cfilter_body.inc"
};
template <typename PointT>
class CRegistration : public CloudUtility<PointT>
{
  public:
#include "creg_body.inc"
};
#include "map_decl.inc"
#include "map_body.inc"
} // namespace lo

#include "cregistration_hip.hpp" // the adapter under test: sees exactly the types a MULLS translation unit has at this point

static bool read_cloud(FILE *f, pcTPtr &c)
{
	uint32_t n = 0;
	if (fread(&n, 4, 1, f) != 1)
		return false;
	c->points.resize(n);
	return n == 0 || fread(c->points.data(), sizeof(Point_T), n, f) == n;
}

static void print_result(const char *who, int code, const lo::constraint_t &con, size_t tree_pts)
{
	printf("{\"who\": \"%s\", \"code\": %d, \"sigma\": %.9g, \"confidence\": %.9g, \"tree_points\": %zu, \"T\": [", who, code, con.sigma,
		   con.confidence, tree_pts);
	for (int i = 0; i < 16; i++)
		printf("%s%.17g", i ? ", " : "", con.Trans1_2.data()[i]);
	printf("], \"info\": [");
	for (int i = 0; i < 36; i++)
		printf("%s%.17g", i ? ", " : "", con.information_matrix.data()[i]);
	printf("]}\n");
}

int main(int argc, char **argv)
{
	if (argc < 3)
		return 2;
	FILE *f = fopen(argv[1], "rb");
	if (!f)
		return 2;
	lo::constraint_t con_ref, con_hip;
	lo::constraint_t *cons[2] = {&con_ref, &con_hip};
	double guess_raw[16], bound[6];
	{
		lo::cloudblock_t &b1 = *con_ref.block1, &b2 = *con_ref.block2;
		pcTPtr *tg[6] = {&b1.pc_ground, &b1.pc_pillar, &b1.pc_facade, &b1.pc_beam, &b1.pc_roof, &b1.pc_vertex};
		pcTPtr *sd[6] = {&b2.pc_ground_down, &b2.pc_pillar_down, &b2.pc_facade_down, &b2.pc_beam_down, &b2.pc_roof_down, &b2.pc_vertex};
		for (int c = 0; c < 6; c++)
			if (!read_cloud(f, *tg[c]))
				return 3;
		for (int c = 0; c < 6; c++)
			if (!read_cloud(f, *sd[c]))
				return 3;
		if (fread(guess_raw, 8, 16, f) != 16 || fread(bound, 8, 6, f) != 6)
			return 3;
		fclose(f);
		b1.local_bound.min_x = bound[0], b1.local_bound.min_y = bound[1], b1.local_bound.min_z = bound[2];
		b1.local_bound.max_x = bound[3], b1.local_bound.max_y = bound[4], b1.local_bound.max_z = bound[5];
	}
	// identical second constraint (deep copies of the clouds)
	{
		lo::cloudblock_t &a1 = *con_ref.block1, &a2 = *con_ref.block2, &b1 = *con_hip.block1, &b2 = *con_hip.block2;
		*b1.pc_ground = *a1.pc_ground, *b1.pc_pillar = *a1.pc_pillar, *b1.pc_facade = *a1.pc_facade, *b1.pc_beam = *a1.pc_beam;
		*b1.pc_roof = *a1.pc_roof, *b1.pc_vertex = *a1.pc_vertex;
		*b2.pc_ground_down = *a2.pc_ground_down, *b2.pc_pillar_down = *a2.pc_pillar_down, *b2.pc_facade_down = *a2.pc_facade_down;
		*b2.pc_beam_down = *a2.pc_beam_down, *b2.pc_roof_down = *a2.pc_roof_down, *b2.pc_vertex = *a2.pc_vertex;
		b1.local_bound = a1.local_bound;
	}
	Eigen::Matrix4d initial_guess_tran;
	std::memcpy(initial_guess_tran.data(), guess_raw, sizeof(guess_raw));

	if (std::string(argv[2]) == "features")
	{
		// extract_semantic_pts' two stages on a raw scan (handed over in block1's ground slot): the reference's CFilter members vs the
		// bridge functions of the same names, same arguments (run_mulls_reg.sh's flags, ground normal method 0)
		pcTPtr scan = con_ref.block1->pc_ground;
		for (int w = 0; w < 2; w++)
		{
			pcTPtr in(new pcT());
			*in = *scan;
			pcTPtr g(new pcT()), gd(new pcT()), u(new pcT()), curb(new pcT()), o[9];
			for (int k = 0; k < 9; k++)
				o[k].reset(new pcT());
			if (w == 0)
			{
				lo::CFilter<Point_T> cf;
				cf.fast_ground_filter(in, g, gd, u, curb, 8, 2.0f, 0.25f, 1.2f, 2.0f, 10, 2, 3, 0, 0, 2.0f, 0, 15.0f, false, 500, false, FLT_MAX, false);
				cf.classify_nground_pts(u, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8], 1.0f, 50, 8, 1, 0.65f, 0.65f, 0.75f, 0.75f, 2, 0.10f, 1.5f, 0.94f,
										0.17f, 0.98f, 0.34f, false, 200, 800, 200, 200, 20000, FLT_MAX, 0.0f, 0.3f, true, false);
			}
			else
			{
				lo::hip::fast_ground_filter<Point_T>(in, g, gd, u, curb, 8, 2.0f, 0.25f, 1.2f, 2.0f, 10, 2, 3, 0, 0, 2.0f, 0, 15.0f, false, 500, false, FLT_MAX, false);
				lo::hip::classify_nground_pts<Point_T>(u, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8], 1.0f, 50, 8, 1, 0.65f, 0.65f, 0.75f, 0.75f, 2, 0.10f,
													   1.5f, 0.94f, 0.17f, 0.98f, 0.34f, false, 200, 800, 200, 200, 20000, FLT_MAX, 0.0f, 0.3f, true, false);
			}
			pcTPtr all[12] = {g, gd, u, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8]};
			printf("{\"who\": \"%s\", \"sizes\": [", w == 0 ? "reference" : "hip");
			for (int k = 0; k < 12; k++)
				printf("%s%zu", k ? ", " : "", all[k]->points.size());
			printf("], \"sums\": [");
			for (int k = 0; k < 12; k++)
			{
				// the classes' 40 meaningful bytes per record (the bridge copies whole records, the reference's push_back too: all 48 compare)
				unsigned long long h = 1469598103934665603ull;
				const unsigned char *b = reinterpret_cast<const unsigned char *>(all[k]->points.data());
				for (size_t i = 0; i < all[k]->points.size() * sizeof(Point_T); i++)
					h = (h ^ b[i]) * 1099511628211ull;
				printf("%s\"%016llx\"", k ? ", " : "", h);
			}
			printf("]}\n");
		}
		// the same chain through its one entry point, extract_semantic_pts, with the scanner filter on: the reference member vs the bridge, on
		// cloudblocks; a second time with a voxel grid ahead of the ground filter, a high non-ground down-sampling rate and the adaptive parameter
		// update (which then lowers that rate: fewer than 200 facade + pillar points come out)
		// ... and a third pair of runs with semantic_assisted (round 5): the label pre-filter in place of the scanner filter
		for (int w = 0; w < 6; w++)
		{
			lo::cloudblock_Ptr blk(new lo::cloudblock_t());
			*blk->pc_raw = *scan;
			const bool second = w >= 2 && w < 4, semantic = w >= 4;
			int gdr = 10, ndr = second ? 20 : 3;
			const float vox = second ? 0.08f : 0.0f, thre_down = 0.75f;
			if (w % 2 == 0)
			{
				lo::CFilter<Point_T> cf;
				cf.extract_semantic_pts(blk, vox, 2.0f, 0.25f, 1.2f, 2.0f, gdr, ndr, 1.0f, 50, 0.65f, 0.65f, 0.10f, thre_down, thre_down, false, 0, 15.0f, 0, 2.0f, second, true, false, 2, 8, 0, 2,
										8, 1, FLT_MAX, 0.94f, 0.17f, 0.98f, 0.34f, true, false, 500, 200, 800, 200, 200, 20000, FLT_MAX, 0.0f, 2.0f, -7.0f, 0.3f, semantic, false, 0.0f, 0.0f);
			}
			else
				lo::hip::extract_semantic_pts<Point_T>(blk, vox, 2.0f, 0.25f, 1.2f, 2.0f, gdr, ndr, 1.0f, 50, 0.65f, 0.65f, 0.10f, thre_down, thre_down, false, 0, 15.0f, 0, 2.0f, second, true, false, 2,
													   8, 0, 2, 8, 1, FLT_MAX, 0.94f, 0.17f, 0.98f, 0.34f, true, false, 500, 200, 800, 200, 200, 20000, FLT_MAX, 0.0f, 2.0f, -7.0f, 0.3f,
													   semantic, false, 0.0f, 0.0f);
			pcTPtr all[15] = {blk->pc_raw,	  blk->pc_down,	 blk->pc_sketch, blk->pc_ground,	  blk->pc_ground_down, blk->pc_unground,	blk->pc_pillar,	   blk->pc_beam,
							  blk->pc_facade, blk->pc_roof, blk->pc_pillar_down, blk->pc_beam_down, blk->pc_facade_down, blk->pc_roof_down, blk->pc_vertex};
			static const char *who[6] = {"reference_block", "hip_block", "reference_block_voxels", "hip_block_voxels", "reference_block_semantic", "hip_block_semantic"};
			printf("{\"who\": \"%s\", \"down_feature_point_num\": %d, \"rates\": [%d, %d], \"sizes\": [", who[w], blk->down_feature_point_num, gdr, ndr);
			for (int k = 0; k < 15; k++)
				printf("%s%zu", k ? ", " : "", all[k]->points.size());
			printf("], \"sums\": [");
			for (int k = 0; k < 15; k++)
			{
				// pc_raw / pc_down / pc_sketch (k < 3): position, intensity, curvature only — upstream's ground filter writes (0,0,1) normals and
				// data[3] heights into the cloud it is given (pc_down = pc_raw), the bridge leaves the scan's points as they are
				unsigned long long h = 1469598103934665603ull;
				const unsigned char *b = reinterpret_cast<const unsigned char *>(all[k]->points.data());
				for (size_t i = 0; i < all[k]->points.size() * sizeof(Point_T); i++)
					if (k >= 3 || i % sizeof(Point_T) < 12 || (i % sizeof(Point_T) >= 32 && i % sizeof(Point_T) < 40))
						h = (h ^ b[i]) * 1099511628211ull;
				printf("%s\"%016llx\"", k ? ", " : "", h);
			}
			printf("]}\n");
		}
		return 0;
	}
	if (std::string(argv[2]) == "motion")
	{
		// test/mulls_slam.cpp:703-712: the frame's clouds (block2 here: the *_down clouds and the key points) moved by their time-stamp fraction of the
		// registration result — the reference's CFilter members vs the bridge functions of the same names, same arguments
		for (int w = 0; w < 2; w++)
		{
			lo::cloudblock_t &b = *cons[w]->block2;
			Eigen::Matrix4d T = initial_guess_tran;
			if (w == 0)
			{
				lo::CFilter<Point_T> cf;
				cf.apply_motion_compensation(b.pc_ground_down, T, 0.1f);
				cf.batch_apply_motion_compensation(b.pc_ground_down, b.pc_pillar_down, b.pc_facade_down, b.pc_beam_down, b.pc_roof_down, b.pc_vertex, T);
			}
			else
			{
				lo::hip::apply_motion_compensation<Point_T>(b.pc_ground_down, T, 0.1f);
				lo::hip::batch_apply_motion_compensation<Point_T>(b.pc_ground_down, b.pc_pillar_down, b.pc_facade_down, b.pc_beam_down, b.pc_roof_down, b.pc_vertex, T);
			}
			pcTPtr all[6] = {b.pc_ground_down, b.pc_pillar_down, b.pc_facade_down, b.pc_beam_down, b.pc_roof_down, b.pc_vertex};
			printf("{\"who\": \"%s\", \"sizes\": [", w == 0 ? "reference" : "hip");
			for (int k = 0; k < 6; k++)
				printf("%s%zu", k ? ", " : "", all[k]->points.size());
			printf("], \"xyz\": [");
			bool first = true;
			for (int k = 0; k < 6; k++)
				for (size_t i = 0; i < all[k]->points.size(); i++)
				{
					const Point_T &p = all[k]->points[i];
					unsigned int u[3];
					std::memcpy(u, &p.x, 12);
					printf("%s%u, %u, %u", first ? "" : ", ", u[0], u[1], u[2]);
					first = false;
				}
			printf("]}\n");
		}
		return 0;
	}
	if (std::string(argv[2]) == "map")
	{
		// scan-to-map step of test/mulls_slam.cpp: mm_lls_icp against the local map (block1), then update_local_map with the
		// registered frame (block2) and map-based dynamic removal on — the reference through the kd-trees its mm_lls_icp left
		// on block1, the bridge through the device-resident mirror of block1
		lo::CRegistration<Point_T> cr;
		lo::MapManager mm;
		auto fnv = [](const pcTPtr &c) {
			unsigned long long h = 1469598103934665603ull;
			const unsigned char *b = (const unsigned char *)c->points.data();
			for (size_t i = 0; i < c->points.size(); i++)
				for (int k = 0; k < 40; k++) // x..curvature, not the trailing padding
					h = (h ^ b[i * sizeof(Point_T) + k]) * 1099511628211ull;
			return h;
		};
		lo::hip::attach_local_map(con_hip.block1);
		for (int w = 0; w < 2; w++)
		{
			lo::constraint_t &con = *cons[w];
			int code;
			if (w == 0)
				code = cr.mm_lls_icp(con, 20, 2.0, 0.002, 0.01, 0.4, 1.1, "111000", "1101", 1.0, 0.1, 0.1, 0.1, initial_guess_tran);
			else
				code = lo::hip::mm_lls_icp<Point_T>(con, 20, 2.0, 0.002, 0.01, 0.4, 1.1, "111000", "1101", 1.0, 0.1, 0.1, 0.1, initial_guess_tran);
			con.block1->pose_lo.setIdentity();
			// both runs continue from the reference's pose (the two Trans1_2 agree to ~1e-12, not to the bit: float rounding of the
			// transformed map could then differ in a last place and hide a real mismatch behind a tolerance)
			con.block2->pose_lo = con.block1->pose_lo * con_ref.Trans1_2;
			con.block1->feature_point_num = (int)(con.block1->pc_ground->points.size() + con.block1->pc_facade->points.size() +
												  con.block1->pc_roof->points.size() + con.block1->pc_pillar->points.size() +
												  con.block1->pc_beam->points.size());
			const int max_pts = 4 * con.block1->feature_point_num;
			if (w == 0)
				mm.update_local_map(con.block1, con.block2, 60.0f, max_pts, 100000, 60, true, "111000", 25.0f, 0.25f, 1.0f, 0.05f, false);
			else
				lo::hip::update_local_map(con.block1, con.block2, 60.0f, max_pts, 100000, 60, true, "111000", 25.0f, 0.25f, 1.0f, 0.05f, false);
			lo::cloudblock_t &m = *con.block1, &fr = *con.block2;
			printf("{\"who\": \"%s\", \"code\": %d, \"n\": [%zu, %zu, %zu, %zu, %zu, %zu], \"frame_n\": [%zu, %zu, %zu, %zu, %zu], ",
				   w == 0 ? "reference_map" : "hip_map", code, m.pc_ground->points.size(), m.pc_pillar->points.size(), m.pc_facade->points.size(),
				   m.pc_beam->points.size(), m.pc_roof->points.size(), m.pc_vertex->points.size(), fr.pc_ground_down->points.size(),
				   fr.pc_pillar_down->points.size(), fr.pc_facade_down->points.size(), fr.pc_beam_down->points.size(),
				   fr.pc_roof_down->points.size());
			printf("\"hash\": [\"%llx\", \"%llx\", \"%llx\", \"%llx\", \"%llx\", \"%llx\"], \"frame_hash\": [\"%llx\", \"%llx\", \"%llx\"], ",
				   fnv(m.pc_ground), fnv(m.pc_pillar), fnv(m.pc_facade), fnv(m.pc_beam), fnv(m.pc_roof), fnv(m.pc_vertex), fnv(fr.pc_ground_down),
				   fnv(fr.pc_pillar_down), fnv(fr.pc_facade_down));
			printf("\"feature_point_num\": %d, \"local_bound\": [%.17g, %.17g, %.17g, %.17g, %.17g, %.17g], \"bound\": [%.17g, %.17g, %.17g, %.17g, %.17g, %.17g]}\n",
				   m.feature_point_num, m.local_bound.min_x, m.local_bound.min_y, m.local_bound.min_z, m.local_bound.max_x, m.local_bound.max_y,
				   m.local_bound.max_z, m.bound.min_x, m.bound.min_y, m.bound.min_z, m.bound.max_x, m.bound.max_y, m.bound.max_z);
		}
		return 0;
	}
	if (std::string(argv[2]) == "variants")
	{
		// lls_icp_3dof_ground and mm_lls_icp_4dof_global: reference members vs the bridge functions of the same names
		lo::CRegistration<Point_T> cr;
		const bool a3 = cr.lls_icp_3dof_ground(con_ref, 20, 1.5, 0.002, 0.01, 0.4, 1.1, "1111", initial_guess_tran);
		const bool b3 = lo::hip::lls_icp_3dof_ground<Point_T>(con_hip, 20, 1.5, 0.002, 0.01, 0.4, 1.1, "1111", initial_guess_tran);
		print_result("reference_3dof", a3 ? 1 : 0, con_ref, 0);
		print_result("hip_3dof", b3 ? 1 : 0, con_hip, 0);
		con_ref.block2->local_station.x = con_hip.block2->local_station.x = 0.5;
		const bool a4 = cr.mm_lls_icp_4dof_global(con_ref, 60.0f, 8, 2.0);
		const bool b4 = lo::hip::mm_lls_icp_4dof_global<Point_T>(con_hip, 60.0f, 8, 2.0);
		print_result("reference_4dof", a4 ? 1 : 0, con_ref, 0);
		print_result("hip_4dof", b4 ? 1 : 0, con_hip, 0);
		return 0;
	}
	const bool kitti = std::string(argv[2]) == "kitti";
	lo::CRegistration<Point_T> cReg;
	int code[2];
	for (int w = 0; w < 2; w++)
	{
		lo::constraint_t &reg_con = *cons[w];
		if (kitti)
		{
			// test/mulls_slam.cpp:642-648 with script/config/lo_gflag_list_kitti_urban.txt
			const int max_iter = 20;
			const float thre = 1.4f + 1.0f, conv_t = 0.0005f, conv_r = 0.001f, thre_min = 0.5f, win = 0.05f, bearing = 20.0f, sigma_thre = 0.35f;
			if (w == 0)
				code[w] = cReg.mm_lls_icp(reg_con, max_iter, thre, conv_t, conv_r, thre_min, 1.1, "111000", "1111", 1.0, win, win, win,
										  initial_guess_tran, true, false, false, bearing, false, false, sigma_thre, 0.03, 45.0);
			else
				code[w] = lo::hip::mm_lls_icp<Point_T>(reg_con, max_iter, thre, conv_t, conv_r, thre_min, 1.1, "111000", "1111", 1.0, win, win,
													  win, initial_guess_tran, true, false, false, bearing, false, false, sigma_thre, 0.03, 45.0);
		}
		else
		{
			// test/mulls_reg.cpp:194-195 with script/run_mulls_reg.sh values; the remaining nine parameters are defaulted
			const int max_iter = 10;
			const float thre = 3.0f, conv_t = 0.001f, conv_r = 0.01f;
			if (w == 0)
				code[w] = cReg.mm_lls_icp(reg_con, max_iter, thre, conv_t, conv_r, 0.25 * thre, 1.1, "111110", "1101", 1.0, 0.1, 0.1, 0.1,
										  initial_guess_tran);
			else
				code[w] = lo::hip::mm_lls_icp<Point_T>(reg_con, max_iter, thre, conv_t, conv_r, 0.25 * thre, 1.1, "111110", "1101", 1.0, 0.1, 0.1,
													  0.1, initial_guess_tran);
		}
	}
	// kd-tree side effect on block1 (cregistration.hpp:1209-1232): count the points each implementation indexed
	size_t tp[2] = {0, 0};
	for (int w = 0; w < 2; w++)
	{
		lo::cloudblock_t &b = *cons[w]->block1;
		pcTreePtr trees[6] = {b.tree_ground, b.tree_pillar, b.tree_facade, b.tree_beam, b.tree_roof, b.tree_vertex};
		for (auto &t : trees)
			if (t->cloud)
				tp[w] += t->cloud->points.size();
	}
	print_result("reference", code[0], con_ref, tp[0]);
	print_result("hip", code[1], con_hip, tp[1]);
	return 0;
}
