// mulls_oracle.cpp — CPU ORACLE for the MULLS-ICP hot path.  TEST INFRASTRUCTURE ONLY.
//
//   * This file is the checker, never the product: only tests/, __graft_entry__.smoke() and bench.py's
//     cpu_baseline leg may load liboracle.  Nothing under mulls_amd/ links, imports or calls it.
//   * It is a dependency-free restatement (no PCL / Eigen / FLANN / glog — none exist in this image) of
//       /root/reference/include/common/cregistration.hpp:1114-1440  (mm_lls_icp driver)
//       :1685-1696 (batch_transform_feature_points)   :1701-1835 (determine_corres)
//       :1855-1866 (update_corr_dist_thre)            :1869-1967 (multi_metrics_lls_tran_estimation)
//       :1976-2275 (pt2pt / pt2pl / pt2li summations) :2518-2722 (residual pass, weight functions)
//       :2740-2764 (construct_trans_a)                :2795-2836 (get_quat_euler_jacobi)
//       :2894-2922 (intersection_filter)  + utility.hpp:817-886, cfilter.hpp:950-981, :2613-2655
//     and of the third-party behaviour the path relies on (not under /root/reference; unvendored, no lockfile):
//       PCL 1.7-1.10 (apt libpcl-dev, Dockerfile:1-8): CorrespondenceEstimation::determineCorrespondences,
//       CorrespondenceRejectorDistance, transformPointCloudWithNormals<PointT,double>, pcl::Correspondence's
//       distance/weight union; FLANN 1.8/1.9 KDTreeSingleIndex + L2_Simple<float>; Eigen 3.3.7 PartialPivLU
//       inverse, AngleAxisd(Matrix3d).
//   * Pinning.  The reference ships no tests, golden vectors or recorded outputs for this path, and
//     PCL / Eigen / FLANN are not installable here, so the upstream binary cannot be run.  The restatement is pinned
//     instead against oracle/_ref/libmulls_ref.so: the reference's OWN function bodies for the path (cut out of
//     /root/reference at build time by oracle/build_ref.sh) compiled against oracle/ref_shim/shim.hpp, a stand-in for
//     the few Eigen/PCL/boost/glog calls they make.  tests/test_ref_pin.py requires bit-for-bit agreement of every
//     output the reference interface exposes.  That pins all MULLS-authored arithmetic and control flow (quirks
//     included); the third-party behaviour itself (PCL correspondence estimation / rejector / transform, FLANN distance,
//     Eigen LU and AngleAxis) stays a restatement from documentation — "parity partially pinned".  See DESIGN.md §Oracle.
//
// Build: see oracle/Makefile (g++ -O3 -ffp-contract=off -fopenmp).  FMA contraction must stay off: the reference
// is built -O3 without -march (CMakeLists.txt:43), i.e. plain SSE2 mul/add.
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/mulls_hip.h"
#include "pcl_restated.h"

namespace
{

struct Pt // pcl::PointXYZINormal, 48 B
{
	float x, y, z, d3;
	float nx, ny, nz, n3;
	float intensity, curvature, p2, p3;
};
static_assert(sizeof(Pt) == 48, "PointXYZINormal layout");
typedef std::vector<Pt> Cloud;

struct Corr // pcl::Correspondence: {int index_query; int index_match; union{float distance; float weight;};}
{
	int q, m;
	float dw;
};
typedef std::vector<Corr> Corrs;

// ---------------------------------------------------------------------------------------------------------
// small fixed-size linear algebra (column-major like Eigen)
struct M4
{
	double a[16];
	double &operator()(int r, int c) { return a[r + 4 * c]; }
	double operator()(int r, int c) const { return a[r + 4 * c]; }
};
struct M6
{
	double a[36];
	double &operator()(int r, int c) { return a[r + 6 * c]; }
	double operator()(int r, int c) const { return a[r + 6 * c]; }
};

M4 m4_identity()
{
	M4 m;
	for (int i = 0; i < 16; i++)
		m.a[i] = (i % 5 == 0) ? 1.0 : 0.0;
	return m;
}
M4 m4_mul(const M4 &A, const M4 &B)
{
	M4 C;
	for (int r = 0; r < 4; r++)
		for (int c = 0; c < 4; c++)
		{
			double s = 0;
			for (int k = 0; k < 4; k++)
				s += A(r, k) * B(k, c);
			C(r, c) = s;
		}
	return C;
}
// general inverse of a 4x4 (only needed for the undistortion branch; Eigen uses a cofactor formula there,
// an LU-based inverse agrees to rounding)
bool lu_inverse(const double *A, double *Ainv, int n);
M4 m4_inverse(const M4 &A)
{
	M4 R;
	lu_inverse(A.a, R.a, 4);
	return R;
}

// Eigen 3.3 PartialPivLU-style inverse: P A = L U, then solve against the identity.
bool lu_inverse(const double *A, double *Ainv, int n)
{
	double lu[36];
	int perm[6];
	for (int i = 0; i < n * n; i++)
		lu[i] = A[i];
	for (int i = 0; i < n; i++)
		perm[i] = i;
	bool ok = true;
	for (int k = 0; k < n; k++)
	{
		int piv = k;
		double best = std::fabs(lu[k + n * k]);
		for (int r = k + 1; r < n; r++)
		{
			double v = std::fabs(lu[r + n * k]);
			if (v > best)
			{
				best = v;
				piv = r;
			}
		}
		if (best == 0.0)
			ok = false; // Eigen does not stop either; inf/NaN propagate (SURVEY B-11)
		if (piv != k)
		{
			for (int c = 0; c < n; c++)
				std::swap(lu[k + n * c], lu[piv + n * c]);
			std::swap(perm[k], perm[piv]);
		}
		double d = lu[k + n * k];
		for (int r = k + 1; r < n; r++)
			lu[r + n * k] /= d;
		for (int c = k + 1; c < n; c++)
		{
			double u = lu[k + n * c];
			for (int r = k + 1; r < n; r++)
				lu[r + n * c] -= lu[r + n * k] * u;
		}
	}
	for (int c = 0; c < n; c++)
	{
		double y[6];
		for (int r = 0; r < n; r++)
			y[r] = (perm[r] == c) ? 1.0 : 0.0;
		for (int r = 0; r < n; r++) // L y' = y (unit lower)
			for (int k = 0; k < r; k++)
				y[r] -= lu[r + n * k] * y[k];
		for (int r = n - 1; r >= 0; r--) // U x = y'
		{
			for (int k = r + 1; k < n; k++)
				y[r] -= lu[r + n * k] * y[k];
			y[r] /= lu[r + n * r];
		}
		for (int r = 0; r < n; r++)
			Ainv[r + n * c] = y[r];
	}
	return ok;
}

// Eigen 3.3 inverse of a fixed 3x3 (compute_inverse_size3_helper): cofactors over the determinant, no pivoting
void inverse3_cofactor(const double m[9] /* col-major */, double out[9])
{
	auto M = [&](int r, int c) { return m[r + 3 * c]; };
	auto cof = [&](int i, int j) {
		const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
		return M(i1, j1) * M(i2, j2) - M(i1, j2) * M(i2, j1);
	};
	const double c0 = cof(0, 0), c1 = cof(1, 0), c2 = cof(2, 0);
	const double det = (c0 * M(0, 0) + c1 * M(1, 0)) + c2 * M(2, 0);
	const double invdet = 1.0 / det;
	out[0 + 3 * 0] = c0 * invdet;
	out[0 + 3 * 1] = c1 * invdet;
	out[0 + 3 * 2] = c2 * invdet;
	out[1 + 3 * 0] = cof(0, 1) * invdet;
	out[1 + 3 * 1] = cof(1, 1) * invdet;
	out[1 + 3 * 2] = cof(2, 1) * invdet;
	out[2 + 3 * 0] = cof(0, 2) * invdet;
	out[2 + 3 * 1] = cof(1, 2) * invdet;
	out[2 + 3 * 2] = cof(2, 2) * invdet;
}

// Eigen::AngleAxisd(Matrix3d).angle(): rotation matrix -> quaternion (Shepperd branches) -> 2*atan2(|v|,|w|)
double angle_of_rotation(const M4 &T)
{
	double m[3][3];
	for (int r = 0; r < 3; r++)
		for (int c = 0; c < 3; c++)
			m[r][c] = T(r, c);
	double q[4]; // x y z w
	double t = m[0][0] + m[1][1] + m[2][2];
	if (t > 0.0)
	{
		t = std::sqrt(t + 1.0);
		q[3] = 0.5 * t;
		t = 0.5 / t;
		q[0] = (m[2][1] - m[1][2]) * t;
		q[1] = (m[0][2] - m[2][0]) * t;
		q[2] = (m[1][0] - m[0][1]) * t;
	}
	else
	{
		int i = 0;
		if (m[1][1] > m[0][0])
			i = 1;
		if (m[2][2] > m[i][i])
			i = 2;
		int j = (i + 1) % 3, k = (j + 1) % 3;
		t = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
		q[i] = 0.5 * t;
		t = 0.5 / t;
		q[3] = (m[k][j] - m[j][k]) * t;
		q[j] = (m[j][i] + m[i][j]) * t;
		q[k] = (m[k][i] + m[i][k]) * t;
	}
	double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
	if (n != 0.0)
		return 2.0 * std::atan2(n, std::fabs(q[3]));
	return 0.0;
}

// ---------------------------------------------------------------------------------------------------------
// pcl::transformPointCloudWithNormals<PointT,double>(cloud, cloud, T)  (cregistration.hpp:1690-1695)
void transform_cloud(Cloud &c, const M4 &T)
{
	for (size_t i = 0; i < c.size(); i++)
	{
		Pt &p = c[i];
		double x = p.x, y = p.y, z = p.z;
		p.x = static_cast<float>(T(0, 0) * x + T(0, 1) * y + T(0, 2) * z + T(0, 3));
		p.y = static_cast<float>(T(1, 0) * x + T(1, 1) * y + T(1, 2) * z + T(1, 3));
		p.z = static_cast<float>(T(2, 0) * x + T(2, 1) * y + T(2, 2) * z + T(2, 3));
		double nx = p.nx, ny = p.ny, nz = p.nz;
		p.nx = static_cast<float>(T(0, 0) * nx + T(0, 1) * ny + T(0, 2) * nz);
		p.ny = static_cast<float>(T(1, 0) * nx + T(1, 1) * ny + T(1, 2) * nz);
		p.nz = static_cast<float>(T(2, 0) * nx + T(2, 1) * ny + T(2, 2) * nz);
	}
}

// FLANN L2_Simple<float>: result += diff*diff over x,y,z in that order, float accumulation
inline float l2_simple(const Pt &a, const Pt &b)
{
	float result = 0.0f, diff;
	diff = a.x - b.x;
	result += diff * diff;
	diff = a.y - b.y;
	result += diff * diff;
	diff = a.z - b.z;
	result += diff * diff;
	return result;
}

// Exact 1-NN index standing in for pcl::search::KdTree -> FLANN KDTreeSingleIndex (max leaf 15).  Ties on the
// float distance resolve to the lowest target index (FLANN's tie order is implementation-defined, SURVEY B-16).
class KdTree
{
  public:
	void build(const Cloud &c)
	{
		cloud_ = &c;
		idx_.resize(c.size());
		for (size_t i = 0; i < c.size(); i++)
			idx_[i] = (int)i;
		nodes_.clear();
		nodes_.reserve(c.size() / 4 + 8);
		if (!c.empty())
			build_rec(0, (int)c.size());
	}
	bool empty() const { return nodes_.empty(); }
	// returns index or -1; d2 = squared distance (float)
	int nearest(const Pt &q, float &d2) const
	{
		int best = -1;
		float bd = FLT_MAX;
		if (!nodes_.empty())
			search(0, q, best, bd);
		d2 = bd;
		return best;
	}
	// k nearest (ascending), for the normal-shooting variant
	void knearest(const Pt &q, int k, std::vector<std::pair<float, int>> &out) const
	{
		out.clear();
		if (nodes_.empty())
			return;
		ksearch(0, q, k, out);
	}

  private:
	struct Node
	{
		int lo, hi;		  // leaf: range in idx_
		int left, right;  // children or -1
		int dim;
		float split_lo, split_hi; // max of left side / min of right side along dim
	};
	static float coord(const Pt &p, int d) { return d == 0 ? p.x : (d == 1 ? p.y : p.z); }

	int build_rec(int lo, int hi)
	{
		Node n;
		n.lo = lo;
		n.hi = hi;
		n.left = n.right = -1;
		n.dim = 0;
		n.split_lo = n.split_hi = 0;
		int id = (int)nodes_.size();
		nodes_.push_back(n);
		if (hi - lo <= 15)
			return id;
		float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
		for (int i = lo; i < hi; i++)
			for (int d = 0; d < 3; d++)
			{
				float v = coord((*cloud_)[idx_[i]], d);
				mn[d] = std::min(mn[d], v);
				mx[d] = std::max(mx[d], v);
			}
		int dim = 0;
		if (mx[1] - mn[1] > mx[dim] - mn[dim])
			dim = 1;
		if (mx[2] - mn[2] > mx[dim] - mn[dim])
			dim = 2;
		int mid = (lo + hi) / 2;
		std::nth_element(idx_.begin() + lo, idx_.begin() + mid, idx_.begin() + hi, [&](int a, int b) {
			float va = coord((*cloud_)[a], dim), vb = coord((*cloud_)[b], dim);
			return va < vb || (va == vb && a < b);
		});
		float slo = -FLT_MAX, shi = FLT_MAX;
		for (int i = lo; i < mid; i++)
			slo = std::max(slo, coord((*cloud_)[idx_[i]], dim));
		for (int i = mid; i < hi; i++)
			shi = std::min(shi, coord((*cloud_)[idx_[i]], dim));
		int l = build_rec(lo, mid);
		int r = build_rec(mid, hi);
		nodes_[id].left = l;
		nodes_[id].right = r;
		nodes_[id].dim = dim;
		nodes_[id].split_lo = slo;
		nodes_[id].split_hi = shi;
		return id;
	}
	void search(int id, const Pt &q, int &best, float &bd) const
	{
		const Node &n = nodes_[id];
		if (n.left < 0)
		{
			for (int i = n.lo; i < n.hi; i++)
			{
				int t = idx_[i];
				float d = l2_simple(q, (*cloud_)[t]);
				if (d < bd || (d == bd && t < best))
				{
					bd = d;
					best = t;
				}
			}
			return;
		}
		float v = coord(q, n.dim);
		// conservative lower bounds of the distance to either side (float; a bound <= the true distance)
		float dl = v > n.split_lo ? v - n.split_lo : 0.0f;
		float dr = v < n.split_hi ? n.split_hi - v : 0.0f;
		// shrink by one ulp-ish factor so rounding in d*d can never prune a true (or tied) neighbour
		float bl = dl * dl * 0.999999f, br = dr * dr * 0.999999f;
		if (dl <= dr)
		{
			if (bl <= bd)
				search(n.left, q, best, bd);
			if (br <= bd)
				search(n.right, q, best, bd);
		}
		else
		{
			if (br <= bd)
				search(n.right, q, best, bd);
			if (bl <= bd)
				search(n.left, q, best, bd);
		}
	}
	void ksearch(int id, const Pt &q, int k, std::vector<std::pair<float, int>> &out) const
	{
		const Node &n = nodes_[id];
		if (n.left < 0)
		{
			for (int i = n.lo; i < n.hi; i++)
			{
				int t = idx_[i];
				std::pair<float, int> e(l2_simple(q, (*cloud_)[t]), t);
				if ((int)out.size() < k || e < out.back())
				{
					out.insert(std::upper_bound(out.begin(), out.end(), e), e);
					if ((int)out.size() > k)
						out.pop_back();
				}
			}
			return;
		}
		float v = coord(q, n.dim);
		float dl = v > n.split_lo ? v - n.split_lo : 0.0f;
		float dr = v < n.split_hi ? n.split_hi - v : 0.0f;
		float bl = dl * dl * 0.999999f, br = dr * dr * 0.999999f;
		int first = dl <= dr ? n.left : n.right, second = dl <= dr ? n.right : n.left;
		float bf = dl <= dr ? bl : br, bs = dl <= dr ? br : bl;
		if ((int)out.size() < k || bf <= out.back().first)
			ksearch(first, q, k, out);
		if ((int)out.size() < k || bs <= out.back().first)
			ksearch(second, q, k, out);
	}
	const Cloud *cloud_ = nullptr;
	std::vector<int> idx_;
	std::vector<Node> nodes_;
};

// Census of the third-party operators whose exact form cannot be checked against PCL / FLANN sources in this image
// (tests/test_pcl_operators.py): how often a run lands on an input where two plausible readings of the upstream code differ.
//   [0] radius tests evaluated (CorrespondenceEstimation::determineCorrespondences, `distance > max_dist_sqr` skips)   [1] ... with distance == max_dist_sqr exactly
//   [2] rejector tests evaluated (CorrespondenceRejectorDistance::getRemainingCorrespondences)   [3] ... with distance == max^2 exactly (`<` and `<=` differ)
//   [4] ... with a NaN distance (kept by `!(d > max)`, dropped by `d < max`)
//   [5] brute-force nearest-neighbour queries   [6] ... whose minimum distance is shared by two or more targets (FLANN's tie order is implementation-defined)
std::atomic<unsigned long long> g_census[8];
bool g_rejector_strict = true; // mulls_params.rejector_strict of the registration being run (set on entry; the OpenMP sections read it)

int brute_nearest(const Cloud &tgt, const Pt &q, float &d2)
{
	int ties = 0;
	int best = -1;
	float bd = FLT_MAX;
	for (size_t t = 0; t < tgt.size(); t++)
	{
		float d = l2_simple(q, tgt[t]);
		if (d < bd)
		{
			bd = d;
			best = (int)t;
			ties = 0;
		}
		else if (d == bd)
			ties++;
	}
	g_census[5]++;
	if (ties && best >= 0)
		g_census[6]++;
	d2 = bd;
	return best;
}

// ---------------------------------------------------------------------------------------------------------
// determine_corres (cregistration.hpp:1701-1835).  `src` is mutated exactly like the reference
// (permanent compaction when |src| >= 500); `orig` carries the original index of every surviving source point
// so tests can compare against an implementation that keeps flags instead of compacting.
bool determine_corres(Cloud &src, std::vector<int> &orig, const Cloud &tgt, const KdTree *tree, float dis_thre, Corrs &corr_f,
					  bool normal_shooting_on, bool normal_check, float angle_thre_degree, bool brute,
					  bool duplicate_check = true, int K_filter_distant_point = 500)
{
	const int K_min = 3;
	const float filter_dis_times = 2.5f;
	const int normal_shooting_candidate_count = 10;

	if (!((int)src.size() >= K_min && (int)tgt.size() >= K_min))
		return 0; // corr_f untouched: the previous iteration's content survives

	Corrs corr;
	const double max_distance = filter_dis_times * dis_thre; // float product widened to the double parameter
	if (normal_shooting_on)
	{
		// pcl::registration::CorrespondenceEstimationNormalShooting::determineCorrespondences (PCL 1.8-1.10):
		// among the k nearest, minimise |n_s x (p_t - p_s)|^2 (double); reject if that minimum > max_distance
		// (compared to r, not r^2 — PCL quirk); stored distance = that candidate's squared Euclidean distance.
		std::vector<std::pair<float, int>> knn;
		for (size_t s = 0; s < src.size(); s++)
		{
			tree->knearest(src[s], normal_shooting_candidate_count, knn);
			double min_dist = std::numeric_limits<double>::max();
			int min_index = 0;
			for (size_t j = 0; j < knn.size(); j++)
			{
				const Pt &t = tgt[knn[j].second];
				// PCL computes pt = src - tgt in float fields, then the cross product in double
				float ptx = src[s].x - t.x, pty = src[s].y - t.y, ptz = src[s].z - t.z;
				double Vx = ptx, Vy = pty, Vz = ptz;
				double Nx = src[s].nx, Ny = src[s].ny, Nz = src[s].nz;
				double cx = Ny * Vz - Nz * Vy, cy = Nz * Vx - Nx * Vz, cz = Nx * Vy - Ny * Vx;
				double dist = cx * cx + cy * cy + cz * cz;
				if (dist < min_dist)
				{
					min_dist = dist;
					min_index = (int)j;
				}
			}
			if (knn.empty() || min_dist > max_distance)
				continue;
			Corr c = {(int)s, knn[min_index].second, knn[min_index].first};
			corr.push_back(c);
		}
	}
	else
	{
		// pcl::registration::CorrespondenceEstimation::determineCorrespondences(corrs, double max_distance)
		const double max_dist_sqr = max_distance * max_distance;
		for (size_t s = 0; s < src.size(); s++)
		{
			float d2;
			int t = brute ? brute_nearest(tgt, src[s], d2) : tree->nearest(src[s], d2);
			if (t >= 0)
			{
				g_census[0]++;
				if ((double)d2 == max_dist_sqr)
					g_census[1]++;
			}
			if (t < 0 || (double)d2 > max_dist_sqr)
				continue;
			Corr c = {(int)s, t, d2};
			corr.push_back(c);
		}
	}

	bool compacted = false;
	if ((int)src.size() >= K_filter_distant_point) // :1755
	{
		int count = 0;
		std::vector<unsigned int> table(tgt.size(), 0);
		Cloud src_f;
		std::vector<int> orig_f;
		Corrs kept;
		for (size_t i = 0; i < corr.size(); i++)
		{
			int s_index = corr[i].q, t_index = corr[i].m;
			if (duplicate_check && table[t_index] > 0)
				continue; // erased
			table[t_index]++;
			src_f.push_back(src[s_index]);
			orig_f.push_back(orig[s_index]);
			Corr c = corr[i];
			c.q = count++;
			kept.push_back(c);
		}
		corr.swap(kept);
		src.swap(src_f);
		orig.swap(orig_f);
		compacted = true;
	}

	// CorrespondenceRejectorDistance: setMaximumDistance(float d) stores d*d as float; getCorrespondences()
	// returns WITHOUT touching the output when the input is empty.
	if (!corr.empty())
	{
		const float max_sqr = dis_thre * dis_thre;
		corr_f.clear();
		for (size_t i = 0; i < corr.size(); i++)
		{
			g_census[2]++;
			if (corr[i].dw == max_sqr)
				g_census[3]++;
			if (corr[i].dw != corr[i].dw)
				g_census[4]++;
			if (g_rejector_strict ? corr[i].dw < max_sqr : !(corr[i].dw > max_sqr)) // mulls_params.rejector_strict
				corr_f.push_back(corr[i]);
		}
	}
	else if (compacted)
	{
		// Reference behaviour here is undefined: the source cloud was just replaced by an empty one while the
		// stale corr_f still indexes the old one (out-of-bounds reads at :1813).  Defined here (and in the HIP
		// path) as "no correspondences".
		corr_f.clear();
	}

	if (normal_check) // :1798-1830
	{
		const double cos_thre = std::cos(angle_thre_degree / 180.0 * M_PI);
		Corrs kept;
		for (size_t i = 0; i < corr_f.size(); i++)
		{
			const Pt &ps = src[corr_f[i].q];
			const Pt &pt = tgt[corr_f[i].m];
			double d = (double)ps.nx * (double)pt.nx + (double)ps.ny * (double)pt.ny + (double)ps.nz * (double)pt.nz;
			float cos_intersection_angle = (float)std::fabs(d);
			if ((double)cos_intersection_angle < cos_thre)
				continue;
			kept.push_back(corr_f[i]);
		}
		corr_f.swap(kept);
	}
	return 1;
}

// ---------------------------------------------------------------------------------------------------------
// weight functions (cregistration.hpp:2686-2722)
inline float get_weight_by_dist_adaptive(float dist, int iter_num, float unit_dist = 30.0f, float b_min = 0.7f, float b_max = 1.3f,
										 float b_step = 0.05f)
{
	float t = b_min + b_step * iter_num;
	float b_current = (t < b_max) ? t : b_max;
	float temp_weight = (float)(b_current + (1.0 - b_current) * dist / unit_dist);
	temp_weight = (float)((temp_weight > 0.01) ? (double)temp_weight : 0.01);
	return temp_weight;
}
inline float get_weight_by_intensity(float intensity_1, float intensity_2, float intensity_scale = 255.0f)
{
	float intensity_diff_ratio = std::fabs(intensity_1 - intensity_2) / intensity_scale;
	float intensity_weight = (float)std::exp(-1.0 * intensity_diff_ratio);
	return intensity_weight;
}
inline float get_weight_by_residual(float res, float huber_thre, int delta = 1)
{
	return (float)((res > huber_thre) ? (double)((2 * res * huber_thre + (delta * delta - 2 * delta) * (huber_thre * huber_thre)) / res / res)
									  : (1.0));
}

struct Normal
{
	double lower[36]; // slots written with coeffRef(k) by pt2pl / pt2pt (k = row + 6*col, rows >= cols) AND the diagonal
					  // slots of pt2li (ATPA(j,j) is the same memory), so the += order on the diagonal equals the reference's
	double upper[36]; // strictly-upper slots written with ATPA(j,k), k>j by pt2li
	double b[6];
	void zero()
	{
		std::memset(lower, 0, sizeof(lower));
		std::memset(upper, 0, sizeof(upper));
		std::memset(b, 0, sizeof(b));
	}
};

// pt2pl_lls_summation (cregistration.hpp:2066-2156)
void pt2pl_sum(const Cloud &S, const Cloud &T, Corrs &corr, Normal &N, int iter_num, float weight, bool dist_w, bool resid_w, bool inten_w,
			   float window)
{
	double *A = N.lower;
	for (size_t i = 0; i < corr.size(); i++)
	{
		const Pt &s = S[corr[i].q];
		const Pt &t = T[corr[i].m];
		float px = s.x, py = s.y, pz = s.z, qx = t.x, qy = t.y, qz = t.z;
		float ntx = t.nx, nty = t.ny, ntz = t.nz;
		float pi = s.intensity, qi = t.intensity;
		float w = weight;
		float a = ntz * py - nty * pz;
		float b = ntx * pz - ntz * px;
		float c = nty * px - ntx * py;
		float d = ntx * qx + nty * qy + ntz * qz - ntx * px - nty * py - ntz * pz;
		float dist = std::sqrt(qx * qx + qy * qy + qz * qz);
		if (dist_w)
			w = w * get_weight_by_dist_adaptive(dist, iter_num);
		if (resid_w)
			w = w * get_weight_by_residual(std::fabs(d), window);
		if (inten_w)
			w = w * get_weight_by_intensity((float)(pi + 0.0001), (float)(qi + 0.0001));
		corr[i].dw = w;
		A[0] += w * ntx * ntx;
		A[1] += w * ntx * nty;
		A[2] += w * ntx * ntz;
		A[3] += w * a * ntx;
		A[4] += w * b * ntx;
		A[5] += w * c * ntx;
		A[7] += w * nty * nty;
		A[8] += w * nty * ntz;
		A[9] += w * a * nty;
		A[10] += w * b * nty;
		A[11] += w * c * nty;
		A[14] += w * ntz * ntz;
		A[15] += w * a * ntz;
		A[16] += w * b * ntz;
		A[17] += w * c * ntz;
		A[21] += w * a * a;
		A[22] += w * a * b;
		A[23] += w * a * c;
		A[28] += w * b * b;
		A[29] += w * b * c;
		A[35] += w * c * c;
		N.b[0] += w * d * ntx;
		N.b[1] += w * d * nty;
		N.b[2] += w * d * ntz;
		N.b[3] += w * d * a;
		N.b[4] += w * d * b;
		N.b[5] += w * d * c;
	}
}

// pt2li_lls_pri_direction_summation (cregistration.hpp:2160-2275)
void pt2li_sum(const Cloud &S, const Cloud &T, Corrs &corr, Normal &N, int iter_num, float weight, bool dist_w, bool resid_w, bool inten_w,
			   float window)
{
	for (size_t i = 0; i < corr.size(); i++)
	{
		const Pt &s = S[corr[i].q];
		const Pt &t = T[corr[i].m];
		float px = s.x, py = s.y, pz = s.z, qx = t.x, qy = t.y, qz = t.z;
		float vx = t.nx, vy = t.ny, vz = t.nz;
		float pi = s.intensity, qi = t.intensity;
		float dx = px - qx, dy = py - qy, dz = pz - qz;
		double Am[3][6], bv[3];
		Am[0][0] = 0;
		Am[0][1] = -vz;
		Am[0][2] = vy;
		Am[0][3] = vy * py + vz * pz;
		Am[0][4] = -vy * px;
		Am[0][5] = -vz * px;
		Am[1][0] = vz;
		Am[1][1] = 0;
		Am[1][2] = -vx;
		Am[1][3] = -vx * py;
		Am[1][4] = vz * pz + vx * px;
		Am[1][5] = -vz * py;
		Am[2][0] = -vy;
		Am[2][1] = vx;
		Am[2][2] = 0;
		Am[2][3] = -vx * pz;
		Am[2][4] = -vy * pz;
		Am[2][5] = vx * px + vy * py;
		bv[0] = -vy * dz + vz * dy;
		bv[1] = -vz * dx + vx * dz;
		bv[2] = -vx * dy + vy * dx;
		float ex = (float)std::fabs(bv[0]), ey = (float)std::fabs(bv[1]), ez = (float)std::fabs(bv[2]);
		float ed = std::sqrt(ex * ex + ey * ey + ez * ez);
		float wx = weight;
		float dist = std::sqrt(qx * qx + qy * qy + qz * qz);
		if (dist_w)
			wx *= get_weight_by_dist_adaptive(dist, iter_num);
		if (inten_w)
			wx *= get_weight_by_intensity((float)(pi + 0.0001), (float)(qi + 0.0001));
		if (resid_w)
			wx = wx * get_weight_by_residual(ed, window);
		corr[i].dw = wx;
		double sw = (double)std::sqrt(wx); // std::sqrt(float) -> float, stored in a double Wmat
		for (int r = 0; r < 3; r++)
		{
			for (int c = 0; c < 6; c++)
				Am[r][c] = sw * Am[r][c];
			bv[r] = sw * bv[r];
		}
		for (int j = 0; j < 6; j++)
			for (int k = j; k < 6; k++)
				(j == k ? N.lower : N.upper)[j + 6 * k] += (Am[0][j] * Am[0][k] + Am[1][j] * Am[1][k]) + Am[2][j] * Am[2][k];
		for (int j = 0; j < 6; j++)
			N.b[j] += (Am[0][j] * bv[0] + Am[1][j] * bv[1]) + Am[2][j] * bv[2];
	}
}

// pt2pt_lls_summation (cregistration.hpp:1976-2063).  Does NOT write corr.weight.
void pt2pt_sum(const Cloud &S, const Cloud &T, Corrs &corr, Normal &N, int iter_num, float weight, bool dist_w, bool resid_w, bool inten_w,
			   float window)
{
	double *A = N.lower;
	for (size_t i = 0; i < corr.size(); i++)
	{
		const Pt &s = S[corr[i].q];
		const Pt &t = T[corr[i].m];
		float px = s.x, py = s.y, pz = s.z, qx = t.x, qy = t.y, qz = t.z;
		float pi = s.intensity, qi = t.intensity;
		float dx = px - qx, dy = py - qy, dz = pz - qz;
		float wx = weight, wy, wz;
		float dist = std::sqrt(qx * qx + qy * qy + qz * qz);
		if (dist_w)
			wx = wx * get_weight_by_dist_adaptive(dist, iter_num);
		if (resid_w)
			wx = wx * get_weight_by_residual(std::sqrt(dx * dx + dy * dy + dz * dz), window);
		if (inten_w)
			wx = wx * get_weight_by_intensity((float)(pi + 0.0001), (float)(qi + 0.0001));
		wy = wx;
		wz = wx;
		A[0] += wx;
		A[4] += wx * pz;
		A[5] += (-wx * py);
		A[7] += wy;
		A[9] += (-wy * pz);
		A[11] += wy * px;
		A[14] += wz;
		A[15] += wz * py;
		A[16] += (-wz * px);
		A[21] += wy * pz * pz + wz * py * py;
		A[22] += (-wz * px * py);
		A[23] += (-wy * px * pz);
		A[28] += wx * pz * pz + wz * px * px;
		A[29] += (-wx * py * pz);
		A[35] += wx * py * py + wy * px * px;
		N.b[0] += (-wx * dx);
		N.b[1] += (-wy * dy);
		N.b[2] += (-wz * dz);
		N.b[3] += wy * pz * dy - wz * py * dz;
		N.b[4] += wz * px * dz - wx * pz * dx;
		N.b[5] += wx * py * dx - wy * px * dy;
	}
}

// assemble the 6x6 the reference actually inverts.  faithful: the mirror (cregistration.hpp:1924-1938) copies the
// LOWER triangle over the upper one, so pt2li's off-diagonal terms (written to the upper triangle) are discarded.
void assemble_atpa(const Normal &N, bool faithful, M6 &ATPA)
{
	for (int c = 0; c < 6; c++)
		for (int r = 0; r < 6; r++)
		{
			int k = r + 6 * c;
			if (r == c)
				ATPA.a[k] = N.lower[k];
			else if (r > c)
				ATPA.a[k] = N.lower[k] + (faithful ? 0.0 : N.upper[c + 6 * r]);
			else
				ATPA.a[k] = 0; // filled by the mirror below
		}
	for (int c = 0; c < 6; c++)
		for (int r = 0; r < c; r++)
			ATPA.a[r + 6 * c] = ATPA.a[c + 6 * r];
}

void construct_trans_a(const double x[6], M4 &T) // cregistration.hpp:2740-2764
{
	double tx = x[0], ty = x[1], tz = x[2], alpha = x[3], beta = x[4], gamma = x[5];
	for (int i = 0; i < 16; i++)
		T.a[i] = 0;
	T(0, 0) = std::cos(gamma) * std::cos(beta);
	T(0, 1) = -std::sin(gamma) * std::cos(alpha) + std::cos(gamma) * std::sin(beta) * std::sin(alpha);
	T(0, 2) = std::sin(gamma) * std::sin(alpha) + std::cos(gamma) * std::sin(beta) * std::cos(alpha);
	T(1, 0) = std::sin(gamma) * std::cos(beta);
	T(1, 1) = std::cos(gamma) * std::cos(alpha) + std::sin(gamma) * std::sin(beta) * std::sin(alpha);
	T(1, 2) = -std::cos(gamma) * std::sin(alpha) + std::sin(gamma) * std::sin(beta) * std::cos(alpha);
	T(2, 0) = -std::sin(beta);
	T(2, 1) = std::cos(beta) * std::sin(alpha);
	T(2, 2) = std::cos(beta) * std::cos(alpha);
	T(0, 3) = tx;
	T(1, 3) = ty;
	T(2, 3) = tz;
	T(3, 3) = 1.0;
}

void get_quat_euler_jacobi(const double e[3], double J[3][3]) // cregistration.hpp:2795-2819 (xyz branch); float locals
{
	float sr = (float)std::sin(0.5 * e[0]), sp = (float)std::sin(0.5 * e[1]), sy = (float)std::sin(0.5 * e[2]);
	float cr = (float)std::cos(0.5 * e[0]), cp = (float)std::cos(0.5 * e[1]), cy = (float)std::cos(0.5 * e[2]);
	J[0][0] = 0.5 * (cr * cp * cy + sr * sp * sy);
	J[0][1] = 0.5 * (-sr * sp * cy - cr * cp * sy);
	J[0][2] = 0.5 * (-sr * cp * sy - cr * sp * cy);
	J[1][0] = 0.5 * (-sr * sp * cy + cr * cp * sy);
	J[1][1] = 0.5 * (cr * cp * cy - sr * sp * sy);
	J[1][2] = 0.5 * (-cr * sp * sy + sr * cp * cy);
	J[2][0] = 0.5 * (-sr * cp * sy - cr * sp * cy);
	J[2][1] = 0.5 * (-cr * sp * sy - sr * cp * cy);
	J[2][2] = 0.5 * (cr * cp * cy + sr * sp * sy);
}

// the part of multi_metrics_lls_tran_estimation after the summations (cregistration.hpp:1951-1964)
bool solve_normal(const M6 &ATPA, const double ATPb[6], double x[6], M6 &cof)
{
	M6 inv;
	bool ok = lu_inverse(ATPA.a, inv.a, 6);
	for (int r = 0; r < 6; r++)
	{
		double s = 0;
		for (int c = 0; c < 6; c++)
			s += inv(r, c) * ATPb[c];
		x[r] = s;
	}
	double J[3][3];
	get_quat_euler_jacobi(x + 3, J);
	cof = inv;
	double Q33[3][3], Q03[3][3], Q30[3][3], tmp[3][3];
	for (int r = 0; r < 3; r++)
		for (int c = 0; c < 3; c++)
		{
			Q33[r][c] = cof(3 + r, 3 + c);
			Q03[r][c] = cof(r, 3 + c);
			Q30[r][c] = cof(3 + r, c);
		}
	// Q33 = J * Q33 * J^T
	for (int r = 0; r < 3; r++)
		for (int c = 0; c < 3; c++)
			tmp[r][c] = J[r][0] * Q33[0][c] + J[r][1] * Q33[1][c] + J[r][2] * Q33[2][c];
	for (int r = 0; r < 3; r++)
		for (int c = 0; c < 3; c++)
			cof(3 + r, 3 + c) = tmp[r][0] * J[c][0] + tmp[r][1] * J[c][1] + tmp[r][2] * J[c][2];
	// Q03 = Q03 * J^T
	for (int r = 0; r < 3; r++)
		for (int c = 0; c < 3; c++)
			cof(r, 3 + c) = Q03[r][0] * J[c][0] + Q03[r][1] * J[c][1] + Q03[r][2] * J[c][2];
	// Q30 = J * Q30
	for (int r = 0; r < 3; r++)
		for (int c = 0; c < 3; c++)
			cof(3 + r, c) = J[r][0] * Q30[0][c] + J[r][1] * Q30[1][c] + J[r][2] * Q30[2][c];
	bool finite = true;
	for (int i = 0; i < 6; i++)
		finite = finite && std::isfinite(x[i]);
	return ok && finite;
}

// residual pass (cregistration.hpp:2546-2677)
void pt2pl_residual(const Cloud &S, const Cloud &T, const Corrs &corr, const double x[6], double &VTPV, int &n)
{
	for (size_t i = 0; i < corr.size(); i++)
	{
		const Pt &s = S[corr[i].q];
		const Pt &t = T[corr[i].m];
		float px = s.x, py = s.y, pz = s.z, qx = t.x, qy = t.y, qz = t.z;
		float ntx = t.nx, nty = t.ny, ntz = t.nz;
		float a = ntz * py - nty * pz;
		float b = ntx * pz - ntz * px;
		float c = nty * px - ntx * py;
		float d = ntx * qx + nty * qy + ntz * qz - ntx * px - nty * py - ntz * pz;
		float residual = (float)(ntx * x[0] + nty * x[1] + ntz * x[2] + a * x[3] + b * x[4] + c * x[5] - d);
		VTPV += corr[i].dw * residual * residual;
		n++;
	}
}
void pt2li_residual(const Cloud &S, const Cloud &T, const Corrs &corr, const double x[6], double &VTPV, int &n)
{
	for (size_t i = 0; i < corr.size(); i++)
	{
		const Pt &s = S[corr[i].q];
		const Pt &t = T[corr[i].m];
		float px = s.x, py = s.y, pz = s.z, qx = t.x, qy = t.y, qz = t.z;
		float vx = t.nx, vy = t.ny, vz = t.nz;
		float dx = px - qx, dy = py - qy, dz = pz - qz;
		double A[3][6] = {{0, vz, -vy, -vz * pz - vy * py, vy * px, vz * px},
						  {-vz, 0, vx, vx * py, -vx * px - vz * pz, vz * py},
						  {vy, -vx, 0, vx * pz, vy * pz, -vy * py - vx * px}};
		double b[3] = {-vz * dy + vy * dz, -vx * dz + vz * dx, -vy * dx + vx * dy};
		double r[3];
		for (int k = 0; k < 3; k++)
		{
			double acc = 0;
			for (int j = 0; j < 6; j++)
				acc += A[k][j] * x[j];
			r[k] = acc - b[k];
		}
		VTPV += corr[i].dw * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
		n += 3;
	}
}
void pt2pt_residual(const Cloud &S, const Cloud &T, const Corrs &corr, const double x[6], double &VTPV, int &n, bool faithful)
{
	for (size_t i = 0; i < corr.size(); i++)
	{
		const Pt &s = S[corr[i].q];
		const Pt &t = T[corr[i].m];
		float px = s.x, py = s.y, pz = s.z, qx = t.x, qy = t.y, qz = t.z;
		float dx = px - qx, dy = py - qy, dz = pz - qz;
		double A[3][6] = {{1, 0, 0, 0, pz, -py}, {0, 1, 0, -pz, 0, px}, {0, 0, 1, py, -px, 0}};
		double b[3] = {-dx, -dy, -dz};
		double r[3];
		for (int k = 0; k < 3; k++)
		{
			double acc = 0;
			for (int j = 0; j < 6; j++)
				acc += A[k][j] * x[j];
			r[k] = acc - b[k];
		}
		// faithful: weight was never written for vertex correspondences, the union still holds d^2
		float w = corr[i].dw;
		(void)faithful; // non-faithful callers store the real weight in dw before the residual pass
		VTPV += w * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
		n += 3;
	}
}

// ---------------------------------------------------------------------------------------------------------
void load_cloud(const mulls_cloud &c, Cloud &out)
{
	out.resize(c.n);
	const uint8_t *p = (const uint8_t *)c.pts;
	for (uint32_t i = 0; i < c.n; i++)
		std::memcpy(&out[i], p + (size_t)i * c.stride, sizeof(Pt));
}

// CFilter::bbx_filter (cfilter.hpp:950-981): strict inequalities, float coordinate vs double bound
void bbx_filter(Cloud &c, const double b[6])
{
	Cloud out;
	for (size_t i = 0; i < c.size(); i++)
		if (c[i].x > b[0] && c[i].x < b[3] && c[i].y > b[1] && c[i].y < b[4] && c[i].z > b[2] && c[i].z < b[5])
			out.push_back(c[i]);
	c.swap(out);
}
void cloud_bbx(const Cloud &c, double b[6]) // utility.hpp:817-848
{
	b[0] = b[1] = b[2] = DBL_MAX;
	b[3] = b[4] = b[5] = -DBL_MAX;
	for (size_t i = 0; i < c.size(); i++)
	{
		if (b[0] > c[i].x)
			b[0] = c[i].x;
		if (b[1] > c[i].y)
			b[1] = c[i].y;
		if (b[2] > c[i].z)
			b[2] = c[i].z;
		if (b[3] < c[i].x)
			b[3] = c[i].x;
		if (b[4] < c[i].y)
			b[4] = c[i].y;
		if (b[5] < c[i].z)
			b[5] = c[i].z;
	}
}

// CFilter::random_downsample_pcl (cfilter.hpp:606-628) with the ABI's seeded selection sampling (Knuth's Algorithm S
// driven by splitmix64): unchanged if size <= keep, emptied if keep == 0, otherwise exactly `keep` points, order kept.
inline uint64_t splitmix64(uint64_t &x)
{
	x += 0x9E3779B97F4A7C15ull;
	uint64_t z = x;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
void random_downsample(Cloud &c, int keep_number, uint64_t seed, int cloud_id)
{
	if (keep_number < 0 || (long)c.size() <= (long)keep_number) // `points.size() <= keep_number` compares as size_t upstream: a negative count keeps all
		return;
	if (keep_number == 0)
	{
		c.clear();
		return;
	}
	uint64_t state = seed ^ (0x100000001B3ull * (uint64_t)(cloud_id + 1));
	Cloud out;
	size_t need = (size_t)keep_number;
	const size_t N = c.size();
	for (size_t i = 0; i < N && need > 0; i++)
	{
		const double u = (double)(splitmix64(state) >> 11) * (1.0 / 9007199254740992.0);
		if (u * (double)(N - i) < (double)need)
		{
			out.push_back(c[i]);
			need--;
		}
	}
	c.swap(out);
}

// ---------------------------------------------------------------------------------------------------------
// CFilter::fast_ground_filter (cfilter.hpp:1658-2036), every estimate_ground_normal_method (the PCL calls of 1 - 3: pcl_restated.h).  Statement by statement: the float /
// double mix of every expression is the reference's (bounds_t holds doubles, the thresholds are floats).  The per-cell
// down-sampling rates of distance_weight_downsampling_method 1 / 2 follow the loop's sequential semantics (upstream the loop
// over cells is an OpenMP parallel-for writing a shared `distance_weight`, :1829-1840).  fixed_num_downsampling uses the ABI's
// seeded selection (upstream: pcl::RandomSample seeded with time(NULL)).
namespace
{
struct GfCell // grid_t (cfilter.hpp:45-69)
{
	std::vector<int> point_id;
	float min_z = 0.f, min_z_outlier_thre = -FLT_MAX, neighbor_min_z = 0.f, dist2station = 0.001f;
	int pts_count = 0, reliable_neighbor_grid_num = 0;
};
} // namespace
int ground_filter_impl(Cloud &in, const mulls_ground_params &P, Cloud &ground, Cloud &ground_down, Cloud &unground)
{
	const int estimate_ground_normal_method = P.estimate_ground_normal_method;
	if (estimate_ground_normal_method < 0 || estimate_ground_normal_method > 3)
		return MULLS_E_INVALID;
	const int min_grid_pt_num = P.min_grid_pt_num;
	const float grid_resolution = P.grid_resolution, max_height_difference = P.max_height_difference, neighbor_height_diff = P.neighbor_height_diff,
				max_ground_height = P.max_ground_height, standard_distance = P.standard_distance, intensity_thre = P.intensity_thre;
	const int ground_random_down_rate = P.ground_random_down_rate, nonground_random_down_rate = P.nonground_random_down_rate;
	const int dw_method = P.distance_weight_downsampling_method;
	const int reliable_grid_pts_count_thre = min_grid_pt_num - 1;
	int count_checkpoint = 0;
	float sum_height = 0.001;
	float appro_mean_height;
	float underground_noise_thre = -FLT_MAX;
	float non_ground_height_thre;
	float distance_weight;
	for (size_t j = 0; j < in.size(); j++) // :1689-1697
		if (j % 100 == 0)
		{
			sum_height += in[j].z;
			count_checkpoint++;
		}
	appro_mean_height = sum_height / count_checkpoint;
	non_ground_height_thre = appro_mean_height + max_ground_height;
	double b[6];
	cloud_bbx(in, b); // get_cloud_bbx (utility.hpp:817-848)
	const double min_x = b[0], min_y = b[1], max_x = b[3], max_y = b[4];
	int row, col, num_grid;
	row = ceil((max_y - min_y) / grid_resolution);
	col = ceil((max_x - min_x) / grid_resolution);
	num_grid = row * col;
	if (num_grid < 0)
		num_grid = 0;
	std::vector<GfCell> grid((size_t)num_grid);
	for (int i = 0; i < num_grid; i++)
	{
		grid[i].min_z = FLT_MAX;
		grid[i].neighbor_min_z = FLT_MAX;
	}
	for (size_t j = 0; j < in.size(); j++) // :1727-1766
	{
		int temp_row, temp_col, temp_id;
		temp_col = floor((in[j].x - min_x) / grid_resolution);
		temp_row = floor((in[j].y - min_y) / grid_resolution);
		temp_id = temp_row * col + temp_col;
		if (temp_id >= 0 && temp_id < num_grid)
		{
			if (dw_method > 0 && !grid[temp_id].pts_count)
				grid[temp_id].dist2station = std::sqrt(in[j].x * in[j].x + in[j].y * in[j].y + in[j].z * in[j].z);
			if (in[j].z > non_ground_height_thre)
			{
				distance_weight = 1.0 * standard_distance / (grid[temp_id].dist2station + 0.0001);
				int nonground_random_down_rate_temp = nonground_random_down_rate;
				if (dw_method == 1)
					nonground_random_down_rate_temp = (int)(distance_weight * nonground_random_down_rate + 1);
				else if (dw_method == 2)
					nonground_random_down_rate_temp = (int)(distance_weight * distance_weight * nonground_random_down_rate + 1);
				if ((int)j % nonground_random_down_rate_temp == 0 || in[j].intensity > intensity_thre)
				{
					in[j].d3 = in[j].z - (appro_mean_height - 3.0);
					unground.push_back(in[j]);
				}
			}
			else if (in[j].z > underground_noise_thre)
			{
				grid[temp_id].pts_count++;
				grid[temp_id].point_id.push_back((int)j);
				if (in[j].z < grid[temp_id].min_z)
				{
					grid[temp_id].min_z = in[j].z;
					grid[temp_id].neighbor_min_z = in[j].z;
				}
			}
		}
	}
	if (P.apply_grid_wise_outlier_filter) // :1770-1790
		for (int i = 0; i < num_grid; i++)
			if (grid[i].pts_count >= min_grid_pt_num)
			{
				double sum_z = 0, sum_z2 = 0, std_z = 0, mean_z = 0;
				for (size_t j = 0; j < grid[i].point_id.size(); j++)
					sum_z += in[grid[i].point_id[j]].z;
				mean_z = sum_z / grid[i].pts_count;
				for (size_t j = 0; j < grid[i].point_id.size(); j++)
					sum_z2 += (in[grid[i].point_id[j]].z - mean_z) * (in[grid[i].point_id[j]].z - mean_z);
				std_z = std::sqrt(sum_z2 / grid[i].pts_count);
				grid[i].min_z_outlier_thre = mean_z - P.outlier_std_scale * std_z;
				grid[i].min_z = (grid[i].min_z > grid[i].min_z_outlier_thre) ? grid[i].min_z : grid[i].min_z_outlier_thre;
				grid[i].neighbor_min_z = grid[i].min_z;
			}
	for (int m = 0; m < num_grid; m++) // :1795-1812
	{
		const int temp_row = m / col, temp_col = m % col;
		if (temp_row >= 1 && temp_row <= row - 2 && temp_col >= 1 && temp_col <= col - 2)
			for (int j = -1; j <= 1; j++)
				for (int k = -1; k <= 1; k++)
				{
					grid[m].neighbor_min_z = (grid[m].neighbor_min_z < grid[m + j * col + k].min_z) ? grid[m].neighbor_min_z : grid[m + j * col + k].min_z;
					if (grid[m + j * col + k].pts_count > reliable_grid_pts_count_thre)
						grid[m].reliable_neighbor_grid_num++;
				}
	}
	std::vector<Cloud> grid_ground_pcs((size_t)num_grid), grid_unground_pcs((size_t)num_grid);
	Cloud grid_ground; // method 3: the cell's ground candidates, every one of them (:1860-1861)
	for (int i = 0; i < num_grid; i++) // :1832-1935
	{
		if (grid[i].pts_count >= min_grid_pt_num && grid[i].reliable_neighbor_grid_num >= P.reliable_neighbor_grid_num_thre)
		{
			int ground_random_down_rate_temp = ground_random_down_rate;
			int nonground_random_down_rate_temp = nonground_random_down_rate;
			distance_weight = 1.0 * standard_distance / (grid[i].dist2station + 0.0001);
			if (dw_method == 1)
			{
				ground_random_down_rate_temp = (int)(distance_weight * ground_random_down_rate + 1);
				nonground_random_down_rate_temp = (int)(distance_weight * nonground_random_down_rate + 1);
			}
			else if (dw_method == 2)
			{
				ground_random_down_rate_temp = (int)(distance_weight * distance_weight * ground_random_down_rate + 1);
				nonground_random_down_rate_temp = (int)(distance_weight * distance_weight * nonground_random_down_rate + 1);
			}
			if (grid[i].min_z - grid[i].neighbor_min_z < neighbor_height_diff)
			{
				for (int j = 0; j < (int)grid[i].point_id.size(); j++)
				{
					Pt &p = in[grid[i].point_id[j]];
					if (p.z > grid[i].min_z_outlier_thre)
					{
						if (p.z - grid[i].min_z < max_height_difference)
						{
							if (estimate_ground_normal_method == 3)
								grid_ground.push_back(p);
							else if (j % ground_random_down_rate_temp == 0)
							{
								if (estimate_ground_normal_method == 0)
								{
									p.nx = 0.0;
									p.ny = 0.0;
									p.nz = 1.0;
								}
								grid_ground_pcs[i].push_back(p);
							}
						}
						else if (j % nonground_random_down_rate_temp == 0 || p.intensity > intensity_thre)
						{
							p.d3 = p.z - grid[i].min_z;
							grid_unground_pcs[i].push_back(p);
						}
					}
				}
			}
			else
			{
				for (int j = 0; j < (int)grid[i].point_id.size(); j++)
				{
					Pt &p = in[grid[i].point_id[j]];
					if (p.z > grid[i].min_z_outlier_thre && (j % nonground_random_down_rate_temp == 0 || p.intensity > intensity_thre))
					{
						p.d3 = p.z - grid[i].neighbor_min_z;
						grid_unground_pcs[i].push_back(p);
					}
				}
			}
			if (estimate_ground_normal_method == 3 && (int)grid_ground.size() >= min_grid_pt_num) // :1909-1932
			{
				// estimate_ground_normal_by_ransac(grid_ground, 0.3 * max_height_difference, 20, ...) (:2038-2056) -> CProceesing::plane_seg_ransac
				// (cprocessing.hpp:67-106): grid_ground becomes the plane's inliers, the normal the plane's first three coefficients
				const float dist_thre = 0.3 * max_height_difference;
				std::vector<restated::P4> pts(grid_ground.size());
				for (size_t k = 0; k < grid_ground.size(); k++)
					pts[k] = restated::P4{grid_ground[k].x, grid_ground[k].y, grid_ground[k].z, grid_ground[k].d3};
				std::vector<int> inliers;
				float coeff[4];
				if (restated::plane_ransac(pts, dist_thre, 20, inliers, coeff)) // (no model: upstream reads an empty coefficient vector — undefined; here: no ground points)
				{
					const float normal_x = coeff[0], normal_y = coeff[1], normal_z = coeff[2];
					for (int j = 0; j < (int)inliers.size(); j++)
						if (j % ground_random_down_rate_temp == 0 && std::abs(normal_z) > 0.8)
						{
							Pt q = grid_ground[inliers[j]];
							q.nx = normal_x;
							q.ny = normal_y;
							q.nz = normal_z;
							grid_ground_pcs[i].push_back(q);
						}
				}
			}
			grid_ground.clear();
		}
	}
	for (int i = 0; i < num_grid; i++) // :1938-1942
	{
		ground.insert(ground.end(), grid_ground_pcs[i].begin(), grid_ground_pcs[i].end());
		unground.insert(unground.end(), grid_unground_pcs[i].begin(), grid_unground_pcs[i].end());
	}
	if (estimate_ground_normal_method == 1 || estimate_ground_normal_method == 2) // :1943-1954: pca_estimator.get_normal_pcar / _pcak + check_normal (pca.hpp:66-119, :462-475)
	{
		std::vector<float> nrm;
		restated::normal_estimation(ground, estimate_ground_normal_method == 1 ? (double)P.normal_estimation_radius : 0.0, 2 * min_grid_pt_num, nrm);
		for (size_t i = 0; i < ground.size(); i++)
		{
			const bool finite = std::isfinite(nrm[3 * i]) && std::isfinite(nrm[3 * i + 1]) && std::isfinite(nrm[3 * i + 2]);
			ground[i].nx = finite ? nrm[3 * i] : (float)0.577;
			ground[i].ny = finite ? nrm[3 * i + 1] : (float)0.577;
			ground[i].nz = finite ? nrm[3 * i + 2] : (float)0.577;
		}
	}
	if (!P.fixed_num_downsampling) // :1955-1968
	{
		for (int i = 0; i < (int)ground.size(); i++)
			if (i % P.ground_random_down_down_rate == 0)
				ground_down.push_back(ground[i]);
	}
	else
	{
		ground_down = ground; // random_downsample_pcl(cloud_ground, cloud_ground_down, down_ground_fixed_num)
		random_downsample(ground_down, P.down_ground_fixed_num, P.rng_seed, 12);
	}
	return 0;
}

struct Quat
{
	double w, x, y, z;
};
Quat quat_from_matrix(const M4 &T)
{
	double m[3][3];
	for (int r = 0; r < 3; r++)
		for (int c = 0; c < 3; c++)
			m[r][c] = T(r, c);
	double q[4];
	double t = m[0][0] + m[1][1] + m[2][2];
	if (t > 0.0)
	{
		t = std::sqrt(t + 1.0);
		q[3] = 0.5 * t;
		t = 0.5 / t;
		q[0] = (m[2][1] - m[1][2]) * t;
		q[1] = (m[0][2] - m[2][0]) * t;
		q[2] = (m[1][0] - m[0][1]) * t;
	}
	else
	{
		int i = 0;
		if (m[1][1] > m[0][0])
			i = 1;
		if (m[2][2] > m[i][i])
			i = 2;
		int j = (i + 1) % 3, k = (j + 1) % 3;
		t = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
		q[i] = 0.5 * t;
		t = 0.5 / t;
		q[3] = (m[k][j] - m[j][k]) * t;
		q[j] = (m[j][i] + m[i][j]) * t;
		q[k] = (m[k][i] + m[i][k]) * t;
	}
	Quat r = {q[3], q[0], q[1], q[2]};
	return r;
}
// Eigen 3.3 QuaternionBase::slerp(t, other) from the identity quaternion
Quat slerp_from_identity(double t, const Quat &o)
{
	const double one = 1.0 - DBL_EPSILON;
	double d = o.w; // dot(identity, other)
	double absD = std::fabs(d);
	double scale0, scale1;
	if (absD >= one)
	{
		scale0 = 1.0 - t;
		scale1 = t;
	}
	else
	{
		double theta = std::acos(absD);
		double sinTheta = std::sin(theta);
		scale0 = std::sin((1.0 - t) * theta) / sinTheta;
		scale1 = std::sin((t * theta)) / sinTheta;
	}
	if (d < 0)
		scale1 = -scale1;
	Quat r = {scale0 + scale1 * o.w, scale1 * o.x, scale1 * o.y, scale1 * o.z};
	return r;
}
// Eigen quaternion * vector: v + 2w (u x v) + 2 u x (u x v)  (QuaternionBase::_transformVector)
void quat_rotate(const Quat &q, const double v[3], double out[3])
{
	double ux = q.x, uy = q.y, uz = q.z;
	double uvx = 2.0 * (uy * v[2] - uz * v[1]), uvy = 2.0 * (uz * v[0] - ux * v[2]), uvz = 2.0 * (ux * v[1] - uy * v[0]);
	out[0] = v[0] + q.w * uvx + (uy * uvz - uz * uvy);
	out[1] = v[1] + q.w * uvy + (uz * uvx - ux * uvz);
	out[2] = v[2] + q.w * uvz + (ux * uvy - uy * uvx);
}
// CFilter::apply_motion_compensation(in, out, Tran, s_ambigous_thre) (cfilter.hpp:493-516; the in-place overload :470-491 is the same loop); serial restatement
void apply_motion_compensation(const Cloud &in, Cloud &out, const M4 &Tran, float s_ambigous_thre = 0.0f)
{
	out = in;
	Quat q21 = quat_from_matrix(Tran);
	for (size_t i = 0; i < in.size(); i++)
	{
		float s = in[i].curvature;
		if (s < s_ambigous_thre || s > 1.0 - s_ambigous_thre) // curvature as the timestamp (float against float; float against the double 1.0 - thre)
			continue;
		Quat dq = slerp_from_identity((double)s, q21);
		double v[3] = {in[i].x, in[i].y, in[i].z}, r[3];
		quat_rotate(dq, v, r);
		out[i].x = (float)(r[0] + (double)s * Tran(0, 3));
		out[i].y = (float)(r[1] + (double)s * Tran(1, 3));
		out[i].z = (float)(r[2] + (double)s * Tran(2, 3));
	}
}

void mulls_oracle_default_params_impl(mulls_params *p);
inline bool used(const mulls_params *p, int c) { return p->used_feature_type[c] == '1'; }
inline int metric_of(int c) { return (c == MULLS_PILLAR || c == MULLS_BEAM) ? 1 : (c == MULLS_VERTEX ? 2 : 0); }

int icp_impl(const mulls_pair *pair, const mulls_params *P, mulls_result *R, int brute, int use_omp)
{
	g_rejector_strict = P->rejector_strict != 0;
	auto tic = std::chrono::steady_clock::now();
	int process_code = 0;
	const int min_total_corr_num = 40, min_neccessary_corr_num = 20;
	float neccessary_corr_ratio = 1.0f;
	M6 cofactor, information;
	for (int i = 0; i < 36; i++)
		cofactor.a[i] = information.a[i] = (i % 7 == 0) ? 1.0 : 0.0;
	double sigma_square_post = 1.0;
	M4 TempTran = m4_identity();
	double x[6] = {0, 0, 0, 0, 0, 0};
	int singular = 0;

	float thr[6];
	for (int c = 0; c < 6; c++)
		thr[c] = P->dis_thre_unit;
	float max_bearable_translation = (float)(2.0 * P->dis_thre_unit);
	float converge_rotation = (float)(P->converge_rotation_d / 180.0 * M_PI);
	float max_bearable_rotation = (float)(P->max_bearable_rotation_d / 180.0 * M_PI);

	Cloud tc[6], sc[6];
	std::vector<int> orig[6];
	for (int c = 0; c < 6; c++)
	{
		load_cloud(pair->tgt[c], tc[c]);
		load_cloud(pair->src[c], sc[c]);
	}
	M4 guess;
	std::memcpy(guess.a, pair->init_guess, sizeof(guess.a));
	for (int c = 0; c < 6; c++)
		transform_cloud(sc[c], guess); // :1183 (all six, used or not)

	const bool undistort = P->apply_motion_undistortion != 0;
	R->cropped = 0;
	std::memset(R->crop_box, 0, sizeof(R->crop_box));
	if (P->apply_intersection_filter && !undistort) // :1186-1188, :2894-2922
	{
		double b[3][6], merged[6] = {DBL_MAX, DBL_MAX, DBL_MAX, -DBL_MAX, -DBL_MAX, -DBL_MAX};
		cloud_bbx(sc[MULLS_GROUND], b[0]);
		cloud_bbx(sc[MULLS_PILLAR], b[1]);
		cloud_bbx(sc[MULLS_FACADE], b[2]);
		for (int i = 0; i < 3; i++)
			for (int k = 0; k < 3; k++)
			{
				merged[k] = std::min(merged[k], b[i][k]);
				merged[3 + k] = std::max(merged[3 + k], b[i][3 + k]);
			}
		const float pad = 1.0f;
		double ib[6];
		for (int k = 0; k < 3; k++)
		{
			ib[k] = std::max(pair->tgt_bound[k], merged[k]) - pad;
			ib[3 + k] = std::min(pair->tgt_bound[3 + k], merged[3 + k]) + pad;
		}
		for (int c = 0; c < 6; c++)
		{
			bbx_filter(tc[c], ib);
			bbx_filter(sc[c], ib);
		}
		R->cropped = 1;
		std::memcpy(R->crop_box, ib, sizeof(ib));
	}
	if (P->keep_less_source_points && !undistort) // keep_less_source_pts, cregistration.hpp:2866-2892
	{
		// upstream thins with pcl::RandomSample seeded by time(NULL) (cfilter.hpp:620), i.e. differently on every run;
		// the ABI defines a seeded, order-preserving selection sampling instead (include/mulls_hip.h, rng_seed)
		random_downsample(tc[MULLS_GROUND], (int)(tc[MULLS_GROUND].size() / 2), P->rng_seed, 0 * 6 + MULLS_GROUND);
		random_downsample(tc[MULLS_FACADE], (int)(tc[MULLS_FACADE].size() / 2), P->rng_seed, 0 * 6 + MULLS_FACADE);
		random_downsample(sc[MULLS_GROUND], (int)(tc[MULLS_GROUND].size() / 4), P->rng_seed, 1 * 6 + MULLS_GROUND);
		random_downsample(sc[MULLS_FACADE], (int)(tc[MULLS_FACADE].size() / 2), P->rng_seed, 1 * 6 + MULLS_FACADE);
		random_downsample(sc[MULLS_PILLAR], (int)(tc[MULLS_PILLAR].size()), P->rng_seed, 1 * 6 + MULLS_PILLAR);
		random_downsample(sc[MULLS_BEAM], (int)(tc[MULLS_BEAM].size()), P->rng_seed, 1 * 6 + MULLS_BEAM);
		random_downsample(sc[MULLS_ROOF], (int)(tc[MULLS_ROOF].size()), P->rng_seed, 1 * 6 + MULLS_ROOF);
		random_downsample(sc[MULLS_VERTEX], (int)(tc[MULLS_VERTEX].size()), P->rng_seed, 1 * 6 + MULLS_VERTEX);
	}

	for (int c = 0; c < 6; c++)
	{
		orig[c].resize(sc[c].size());
		for (size_t i = 0; i < sc[c].size(); i++)
			orig[c][i] = (int)i;
		R->nsrc0[c] = (uint32_t)sc[c].size();
		R->ntgt0[c] = (uint32_t)tc[c].size();
	}

	int source_feature_points_count = 0;
	if (used(P, 1))
		source_feature_points_count += (int)sc[MULLS_PILLAR].size();
	if (used(P, 2))
		source_feature_points_count += (int)sc[MULLS_FACADE].size();
	if (used(P, 3))
		source_feature_points_count += (int)sc[MULLS_BEAM].size();

	KdTree tree[6];
	if (!brute)
	{
		// :1209-1232 — three OpenMP sections {ground, roof} {pillar, beam} {facade}, vertex afterwards
#pragma omp parallel sections num_threads(3) if (use_omp)
		{
#pragma omp section
			{
				if (used(P, 0) && !tc[0].empty())
					tree[0].build(tc[0]);
				if (used(P, 4) && !tc[4].empty())
					tree[4].build(tc[4]);
			}
#pragma omp section
			{
				if (used(P, 1) && !tc[1].empty())
					tree[1].build(tc[1]);
				if (used(P, 3) && !tc[3].empty())
					tree[3].build(tc[3]);
			}
#pragma omp section
			{
				if (used(P, 2) && !tc[2].empty())
					tree[2].build(tc[2]);
			}
		}
		if (used(P, 5) && !tc[5].empty())
			tree[5].build(tc[5]);
	}

	Corrs corr[6];
	const bool ns = P->normal_shooting_on != 0;
	const float bearing = P->normal_bearing;
	int iters = 0;
	R->trace_len = 0;

	for (int i = 0; i < P->max_iter_num; i++)
	{
		if (undistort && i == 0) // :1248-1258
		{
			M4 inv_guess = m4_inverse(guess);
			for (int c = 0; c < 5; c++) // vertex is not regenerated (cfilter.hpp:540,547)
			{
				Cloud down;
				load_cloud(pair->src_down[c].pts ? pair->src_down[c] : pair->src[c], down);
				apply_motion_compensation(down, sc[c], inv_guess);
				orig[c].resize(sc[c].size());
				for (size_t k = 0; k < sc[c].size(); k++)
					orig[c][k] = (int)k;
			}
			for (int c = 0; c < 6; c++)
				transform_cloud(sc[c], guess);
		}
		else
			for (int c = 0; c < 6; c++)
				transform_cloud(sc[c], TempTran);

		float thr_used[6];
		for (int c = 0; c < 6; c++)
			thr_used[c] = thr[c];

			// :1268-1292 — sections {ground} {pillar} {facade, beam}, then roof, vertex serially
#pragma omp parallel sections num_threads(3) if (use_omp)
		{
#pragma omp section
			{
				if (used(P, 0) && sc[0].size() > 0)
					determine_corres(sc[0], orig[0], tc[0], &tree[0], thr[0], corr[0], ns, true, bearing, brute);
			}
#pragma omp section
			{
				if (used(P, 1) && sc[1].size() > 0)
					determine_corres(sc[1], orig[1], tc[1], &tree[1], thr[1], corr[1], false, true, bearing, brute);
			}
#pragma omp section
			{
				if (used(P, 2) && sc[2].size() > 0)
					determine_corres(sc[2], orig[2], tc[2], &tree[2], thr[2], corr[2], ns, true, bearing, brute);
				if (used(P, 3) && sc[3].size() > 0)
					determine_corres(sc[3], orig[3], tc[3], &tree[3], thr[3], corr[3], false, true, bearing, brute);
			}
		}
		if (used(P, 4) && sc[4].size() > 0)
			determine_corres(sc[4], orig[4], tc[4], &tree[4], thr[4], corr[4], ns, true, bearing, brute);
		if (used(P, 5) && sc[5].size() > 0)
			determine_corres(sc[5], orig[5], tc[5], &tree[5], thr[5], corr[5], false, false, 40.0f, brute);
		iters = i + 1;

		int total_corr_num = 0;
		for (int c = 0; c < 6; c++)
			total_corr_num += (int)corr[c].size();
		int neccessary_corr_num = (int)(corr[MULLS_PILLAR].size() + corr[MULLS_BEAM].size() + corr[MULLS_FACADE].size());
		neccessary_corr_ratio = (float)(1.0 * neccessary_corr_num / source_feature_points_count);

		mulls_iter_trace *tr = nullptr;
		if (R->trace && R->trace_len < R->trace_cap)
		{
			tr = &R->trace[R->trace_len++];
			std::memset(tr, 0, sizeof(*tr));
			tr->iter = i;
			for (int c = 0; c < 6; c++)
			{
				tr->ncorr[c] = (uint32_t)corr[c].size();
				tr->nsrc[c] = (uint32_t)sc[c].size();
				tr->thr[c] = thr_used[c];
			}
		}
		for (int c = 0; c < 6; c++)
			R->ncorr[c] = (uint32_t)corr[c].size();

		if (total_corr_num < min_total_corr_num || neccessary_corr_num < min_neccessary_corr_num ||
			neccessary_corr_ratio < P->min_neccessary_corr_ratio)
		{
			process_code = -2;
			TempTran = m4_identity();
			break;
		}

		for (int c = 0; c < 6; c++) // update_corr_dist_thre :1855-1866
		{
			double v = 1.0 * thr[c] / P->dis_thre_update_rate;
			thr[c] = (float)((v > P->dis_thre_min) ? v : (double)P->dis_thre_min);
		}

		// multi_metrics_lls_tran_estimation :1869-1967
		{
			float w_ground = 1.0f, w_roof = 1.0f;
			int m1 = (int)(corr[MULLS_GROUND].size() + corr[MULLS_ROOF].size());
			int m2 = (int)corr[MULLS_FACADE].size();
			int m3 = (int)corr[MULLS_PILLAR].size();
			int m4 = (int)corr[MULLS_BEAM].size();
			if (P->weight_strategy[0] == '1')
			{
				double v = P->z_xy_balanced_ratio * (m2 + 2 * m3 - m4) / (0.0001 + 2.0 * m1);
				w_ground = (float)((0.01 > v) ? 0.01 : v);
				w_roof = w_ground;
			}
			bool resid_w = (P->weight_strategy[1] == '1' && i > 2);
			bool dist_w = P->weight_strategy[2] == '1';
			bool inten_w = P->weight_strategy[3] == '1';
			Normal N;
			N.zero();
			pt2pl_sum(sc[0], tc[0], corr[0], N, i, w_ground, dist_w, resid_w, inten_w, P->pt2pl_residual_window);
			pt2pl_sum(sc[2], tc[2], corr[2], N, i, 1.0f, dist_w, resid_w, inten_w, P->pt2pl_residual_window);
			pt2pl_sum(sc[4], tc[4], corr[4], N, i, w_roof, dist_w, resid_w, inten_w, P->pt2pl_residual_window);
			pt2li_sum(sc[1], tc[1], corr[1], N, i, 1.0f, dist_w, resid_w, inten_w, P->pt2li_residual_window);
			pt2li_sum(sc[3], tc[3], corr[3], N, i, 1.0f, dist_w, resid_w, inten_w, P->pt2li_residual_window);
			if (!P->faithful)
			{
				// intended version: keep the real weight for the vertex residual pass
				Corrs tmp = corr[5];
				pt2pt_sum(sc[5], tc[5], corr[5], N, i, 1.0f, dist_w, resid_w, inten_w, P->pt2pt_residual_window);
				for (size_t k = 0; k < corr[5].size(); k++)
				{
					const Pt &s = sc[5][corr[5][k].q];
					const Pt &t = tc[5][corr[5][k].m];
					float dx = s.x - t.x, dy = s.y - t.y, dz = s.z - t.z;
					float w = 1.0f;
					float dist = std::sqrt(t.x * t.x + t.y * t.y + t.z * t.z);
					if (dist_w)
						w = w * get_weight_by_dist_adaptive(dist, i);
					if (resid_w)
						w = w * get_weight_by_residual(std::sqrt(dx * dx + dy * dy + dz * dz), P->pt2pt_residual_window);
					if (inten_w)
						w = w * get_weight_by_intensity((float)(s.intensity + 0.0001), (float)(t.intensity + 0.0001));
					corr[5][k].dw = w;
				}
			}
			else
				pt2pt_sum(sc[5], tc[5], corr[5], N, i, 1.0f, dist_w, resid_w, inten_w, P->pt2pt_residual_window);

			M6 ATPA;
			assemble_atpa(N, P->faithful != 0, ATPA);
			if (!solve_normal(ATPA, N.b, x, cofactor))
				singular = 1;
			if (tr)
			{
				std::memcpy(tr->atpa, ATPA.a, sizeof(tr->atpa));
				std::memcpy(tr->atpb, N.b, sizeof(tr->atpb));
				std::memcpy(tr->x, x, sizeof(tr->x));
			}
		}
		construct_trans_a(x, TempTran);

		double ts_norm = std::sqrt(TempTran(0, 3) * TempTran(0, 3) + TempTran(1, 3) * TempTran(1, 3) + TempTran(2, 3) * TempTran(2, 3));
		double rs_angle = angle_of_rotation(TempTran);
		if (ts_norm > max_bearable_translation || std::fabs(rs_angle) > max_bearable_rotation)
		{
			process_code = -1;
			TempTran = m4_identity();
			break;
		}
		if (i == P->max_iter_num - 1 || (i > 2 && ts_norm < P->converge_translation && std::fabs(rs_angle) < converge_rotation))
		{
			double VTPV = 0;
			int n = 0;
			pt2pl_residual(sc[0], tc[0], corr[0], x, VTPV, n);
			pt2pl_residual(sc[2], tc[2], corr[2], x, VTPV, n);
			pt2pl_residual(sc[4], tc[4], corr[4], x, VTPV, n);
			pt2li_residual(sc[1], tc[1], corr[1], x, VTPV, n);
			pt2li_residual(sc[3], tc[3], corr[3], x, VTPV, n);
			pt2pt_residual(sc[5], tc[5], corr[5], x, VTPV, n, P->faithful != 0);
			sigma_square_post = VTPV / (n - 6);
			process_code = (std::sqrt(sigma_square_post) < (double)P->sigma_thre) ? 1 : -3;
			M6 cinv;
			lu_inverse(cofactor.a, cinv.a, 6);
			for (int k = 0; k < 36; k++)
				information.a[k] = (1.0 / sigma_square_post) * cinv.a[k];
			break;
		}
		guess = m4_mul(TempTran, guess);
	}
	guess = m4_mul(TempTran, guess); // :1403

	R->code = process_code;
	R->iters = iters;
	std::memcpy(R->T, guess.a, sizeof(R->T));
	std::memcpy(R->info, information.a, sizeof(R->info));
	R->sigma = (float)std::sqrt(sigma_square_post);
	R->confidence = neccessary_corr_ratio;
	R->singular = singular;
	R->ms_total = (float)(std::chrono::duration<double>(std::chrono::steady_clock::now() - tic).count() * 1000.0);
	return MULLS_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// lls_icp_3dof_ground (cregistration.hpp:1443-1582) with ground_3dof_lls_tran_estimation / pt2pl_ground_3dof_lls_summation
// (:2278-2386): ground class only, unknowns (roll, pitch, z).  Parameters are read from the mulls_params fields of the
// same names (max_iter_num, dis_thre_unit, converge_*, dis_thre_min, dis_thre_update_rate, weight_strategy,
// keep_less_source_points, max_bearable_rotation_d).
int icp_3dof_impl(const mulls_pair *pair, const mulls_params *P, mulls_result *R, int brute)
{
	g_rejector_strict = P->rejector_strict != 0;
	int process_code = 0;
	const int min_total_corr_num = 100, down_rate = 3;
	M4 S2T = m4_identity(), TempTran = m4_identity();
	float dis_thre_ground = P->dis_thre_unit;
	float max_bearable_translation = (float)(2.0 * P->dis_thre_unit);
	float converge_rotation = (float)(P->converge_rotation_d / 180.0 * M_PI);
	float max_bearable_rotation = (float)(P->max_bearable_rotation_d / 180.0 * M_PI);
	Cloud sc, tc;
	load_cloud(pair->src[MULLS_GROUND], sc);
	load_cloud(pair->tgt[MULLS_GROUND], tc);
	M4 guess;
	std::memcpy(guess.a, pair->init_guess, sizeof(guess.a));
	transform_cloud(sc, guess);
	if (P->keep_less_source_points)
		random_downsample(sc, (int)(tc.size() / down_rate), P->rng_seed, 1 * 6 + MULLS_GROUND);
	std::vector<int> orig(sc.size());
	for (size_t i = 0; i < sc.size(); i++)
		orig[i] = (int)i;
	std::memset(R->nsrc0, 0, sizeof(R->nsrc0));
	std::memset(R->ntgt0, 0, sizeof(R->ntgt0));
	std::memset(R->ncorr, 0, sizeof(R->ncorr));
	R->nsrc0[0] = (uint32_t)sc.size();
	R->ntgt0[0] = (uint32_t)tc.size();
	KdTree tree;
	if (!brute)
		tree.build(tc);
	Corrs corr;
	int iters = 0;
	R->trace_len = 0;
	for (int i = 0; i < P->max_iter_num; i++)
	{
		const float thr_used = dis_thre_ground;
		determine_corres(sc, orig, tc, &tree, dis_thre_ground, corr, false, true, 40.0f, brute != 0);
		iters = i + 1;
		R->ncorr[0] = (uint32_t)corr.size();
		mulls_iter_trace *tr = nullptr;
		if (R->trace && R->trace_len < R->trace_cap)
		{
			tr = &R->trace[R->trace_len++];
			std::memset(tr, 0, sizeof(*tr));
			tr->iter = i;
			tr->ncorr[0] = (uint32_t)corr.size();
			tr->nsrc[0] = (uint32_t)sc.size();
			tr->thr[0] = thr_used;
		}
		if ((int)corr.size() < min_total_corr_num)
		{
			process_code = -2;
			break;
		}
		{
			double v = 1.0 * dis_thre_ground / P->dis_thre_update_rate;
			dis_thre_ground = (float)((v > P->dis_thre_min) ? v : (double)P->dis_thre_min);
		}
		// ground_3dof_lls_tran_estimation: residual weighting has no iteration gate here, window = 0.1 (default argument)
		const bool resid_w = P->weight_strategy[1] == '1', dist_w = P->weight_strategy[2] == '1', inten_w = P->weight_strategy[3] == '1';
		double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, b3[3] = {0, 0, 0};
		for (size_t k = 0; k < corr.size(); k++)
		{
			const Pt &s = sc[corr[k].q];
			const Pt &t = tc[corr[k].m];
			float px = s.x, py = s.y, pz = s.z, qx = t.x, qy = t.y, qz = t.z, ntx = t.nx, nty = t.ny, ntz = t.nz;
			float pi = s.intensity, qi = t.intensity;
			float w = 1.0f;
			float a = ntz * py - nty * pz;
			float b = ntx * pz - ntz * px;
			float d = ntx * qx + nty * qy + ntz * qz - ntx * px - nty * py - ntz * pz;
			float dist = std::sqrt(qx * qx + qy * qy + qz * qz);
			if (dist_w)
				w = w * get_weight_by_dist_adaptive(dist, i);
			if (resid_w)
				w = w * get_weight_by_residual(std::fabs(d), 0.1f);
			if (inten_w)
				w = w * get_weight_by_intensity((float)(pi + 0.0001), (float)(qi + 0.0001));
			corr[k].dw = w;
			A[0] += w * a * a;
			A[1] += w * a * b;
			A[2] += w * a * ntz;
			A[4] += w * b * b;
			A[5] += w * b * ntz;
			A[8] += w * ntz * ntz;
			b3[0] += w * d * a;
			b3[1] += w * d * b;
			b3[2] += w * d * ntz;
		}
		A[3] = A[1];
		A[6] = A[2];
		A[7] = A[5];
		double Ainv[9], x3[3];
		inverse3_cofactor(A, Ainv);
		for (int r = 0; r < 3; r++)
			x3[r] = (Ainv[r + 0] * b3[0] + Ainv[r + 3] * b3[1]) + Ainv[r + 6] * b3[2];
		double x6[6] = {0, 0, x3[2], x3[0], x3[1], 0};
		construct_trans_a(x6, TempTran);
		if (tr)
		{
			std::memcpy(tr->x, x6, sizeof(x6));
			for (int k = 0; k < 9; k++)
				tr->atpa[k] = A[k];
			std::memcpy(tr->atpb, b3, sizeof(b3));
		}
		double ts_norm = std::sqrt(TempTran(0, 3) * TempTran(0, 3) + TempTran(1, 3) * TempTran(1, 3) + TempTran(2, 3) * TempTran(2, 3));
		double rs_angle = angle_of_rotation(TempTran);
		if (ts_norm > max_bearable_translation || std::fabs(rs_angle) > max_bearable_rotation)
		{
			process_code = -1;
			TempTran = m4_identity();
			break;
		}
		if (i == P->max_iter_num - 1 || (i > 2 && ts_norm < P->converge_translation && std::fabs(rs_angle) < converge_rotation))
		{
			process_code = 1;
			break;
		}
		transform_cloud(sc, TempTran);
		S2T = m4_mul(TempTran, S2T);
	}
	S2T = m4_mul(TempTran, S2T);
	S2T = m4_mul(S2T, guess);
	R->code = process_code;
	R->iters = iters;
	std::memcpy(R->T, S2T.a, sizeof(R->T));
	for (int k = 0; k < 36; k++)
		R->info[k] = (k % 7 == 0) ? 1.0 : 0.0; // constraint_t::information_matrix is not touched by this variant
	R->sigma = FLT_MAX;							   // nor are sigma (constraint_t default) and confidence
	R->confidence = 0.0f;
	R->singular = 0;
	R->cropped = 0;
	return MULLS_OK;
}

// mm_lls_icp_4dof_global (cregistration.hpp:1584-1681): heading sweep about the source station, each trial a full
// mm_lls_icp with weights "1001" and classes "111110"; keeps the trial maximising confidence / sigma.
int icp_4dof_impl(const mulls_pair *pair, float heading_step_d, const double station[3], int max_iter_num, float dis_thre_unit,
				  float converge_translation, float dis_thre_min, float dis_thre_update_rate, mulls_result *R, int *success, float *best_heading,
				  int brute, int use_omp)
{
	float current_best_score = 0, current_best_heading_d = 0;
	float heading_d = 0.0f;
	bool successful_reg = false;
	mulls_result best;
	std::memset(&best, 0, sizeof(best));
	for (int k = 0; k < 4; k++)
		best.T[5 * k] = 1.0; // registration_con.Trans1_2 keeps its constructor value (identity) when no trial succeeds
	for (int k = 0; k < 6; k++)
		best.info[7 * k] = 1.0;
	best.sigma = FLT_MAX;
	int count = 0;
	while (heading_d < 360.0)
	{
		float heading_rad = (float)(heading_d * M_PI / 180.0);
		M4 rot = m4_identity(), g2s = m4_identity(), s2g = m4_identity();
		// `cos(heading_rad)` with a float argument under `using namespace std` resolves to std::cos(float)
		rot(0, 0) = std::cos(heading_rad);
		rot(0, 1) = std::sin(heading_rad);
		rot(1, 0) = -std::sin(heading_rad);
		rot(1, 1) = std::cos(heading_rad);
		g2s(0, 3) = -station[0];
		g2s(1, 3) = -station[1];
		g2s(2, 3) = -station[2];
		s2g(0, 3) = station[0];
		s2g(1, 3) = station[1];
		s2g(2, 3) = station[2];
		M4 guess = m4_mul(m4_mul(s2g, rot), g2s);
		mulls_pair trial = *pair;
		std::memcpy(trial.init_guess, guess.a, sizeof(guess.a));
		mulls_params P;
		mulls_oracle_default_params_impl(&P);
		P.max_iter_num = max_iter_num;
		P.dis_thre_unit = dis_thre_unit;
		P.converge_translation = converge_translation;
		P.converge_rotation_d = converge_translation; // the reference passes converge_translation twice (:1640-1642)
		P.dis_thre_min = dis_thre_min;
		P.dis_thre_update_rate = dis_thre_update_rate;
		std::strcpy(P.used_feature_type, "111110");
		std::strcpy(P.weight_strategy, "1001");
		mulls_result r;
		std::memset(&r, 0, sizeof(r));
		icp_impl(&trial, &P, &r, brute, use_omp);
		if (r.code > 0)
		{
			float cur_score = r.confidence / r.sigma;
			if (cur_score > current_best_score)
			{
				best = r;
				current_best_score = cur_score;
				current_best_heading_d = heading_d;
			}
			successful_reg = true;
		}
		heading_d += heading_step_d;
		count++;
	}
	mulls_iter_trace *keep_trace = R->trace;
	int keep_cap = R->trace_cap;
	*R = best;
	R->trace = keep_trace;
	R->trace_cap = keep_cap;
	R->trace_len = 0;
	R->iters = count; // number of heading trials
	if (success)
		*success = successful_reg ? 1 : 0;
	if (best_heading)
		*best_heading = current_best_heading_d;
	return MULLS_OK;
}

} // namespace

extern "C"
{

	void mulls_oracle_default_params(mulls_params *p) { mulls_oracle_default_params_impl(p); }
	// the first n draws of the sample sequence of the ground filter's plane RANSAC (pcl_restated.h: plane_ransac)
	void mulls_oracle_sac_draws(uint32_t *out, int n)
	{
		std::mt19937 eng(12345u);
		for (int i = 0; i < n; i++)
			out[i] = (uint32_t)(eng() >> 1);
	}
	int mulls_oracle_icp_3dof_ground(const mulls_pair *pair, const mulls_params *params, mulls_result *result, int nn_mode)
	{
		if (!pair || !params || !result)
			return MULLS_E_INVALID;
		return icp_3dof_impl(pair, params, result, nn_mode);
	}
	int mulls_oracle_icp_4dof_global(const mulls_pair *pair, float heading_step_d, const double station[3], int max_iter_num, float dis_thre_unit,
									 float converge_translation, float dis_thre_min, float dis_thre_update_rate, mulls_result *result, int *success,
									 float *best_heading_d, int nn_mode, int use_omp)
	{
		if (!pair || !result || !(heading_step_d > 0.0f))
			return MULLS_E_INVALID;
		return icp_4dof_impl(pair, heading_step_d, station, max_iter_num, dis_thre_unit, converge_translation, dis_thre_min, dis_thre_update_rate,
							 result, success, best_heading_d, nn_mode, use_omp);
	}
}

namespace
{
	void mulls_oracle_default_params_impl(mulls_params *p)
	{
		std::memset(p, 0, sizeof(*p));
		p->max_iter_num = 20;
		p->dis_thre_unit = 1.5f;
		p->converge_translation = 0.002f;
		p->converge_rotation_d = 0.01f;
		p->dis_thre_min = 0.4f;
		p->dis_thre_update_rate = 1.1f;
		std::strcpy(p->used_feature_type, "111110");
		std::strcpy(p->weight_strategy, "1101");
		p->z_xy_balanced_ratio = 1.0f;
		p->pt2pt_residual_window = 0.1f;
		p->pt2pl_residual_window = 0.1f;
		p->pt2li_residual_window = 0.1f;
		p->apply_intersection_filter = 1;
		p->normal_bearing = 45.0f;
		p->faithful = 1;
		p->rejector_strict = 1;
		p->sigma_thre = 0.5f;
		p->min_neccessary_corr_ratio = 0.03f;
		p->max_bearable_rotation_d = 45.0f;
	}
} // namespace

extern "C"
{

	// fast_ground_filter: same contract as mulls_ground_filter in include/mulls_hip.h
	int mulls_oracle_ground_filter(const void *pts, uint32_t n, uint32_t stride, const mulls_ground_params *params, void *ground, uint32_t cap_ground,
								   void *ground_down, uint32_t cap_ground_down, void *unground, uint32_t cap_unground, uint32_t n_out[3])
	{
		Cloud in(n), g, gd, u;
		for (uint32_t i = 0; i < n; i++)
			std::memcpy(&in[i], (const unsigned char *)pts + (size_t)i * stride, sizeof(Pt));
		const int rc = ground_filter_impl(in, *params, g, gd, u);
		if (rc != 0)
			return rc;
		auto put = [](const Cloud &c, void *dst, uint32_t cap) {
			const size_t k = std::min<size_t>(c.size(), cap);
			if (k)
				std::memcpy(dst, c.data(), k * sizeof(Pt));
		};
		put(g, ground, cap_ground);
		put(gd, ground_down, cap_ground_down);
		put(u, unground, cap_unground);
		n_out[0] = (uint32_t)g.size();
		n_out[1] = (uint32_t)gd.size();
		n_out[2] = (uint32_t)u.size();
		return 0;
	}

	// operator census (see g_census): copies the eight counters, optionally resetting them
	void mulls_oracle_census(unsigned long long out[8], int reset)
	{
		for (int k = 0; k < 8; k++)
		{
			out[k] = g_census[k].load();
			if (reset)
				g_census[k] = 0;
		}
		// [7]: plane_ransac's stopping test (pcl_restated.h): evaluations (bits 0-31), those within 1e-9 of the boundary (32-47), disagreements of the two forms (48-63)
		out[7] = (restated::ransac_census(2) & 0xffffffffull) | ((restated::ransac_census(0) & 0xffffull) << 32) | ((restated::ransac_census(1) & 0xffffull) << 48);
		if (reset)
			restated::ransac_census(0) = restated::ransac_census(1) = restated::ransac_census(2) = 0;
	}

	// nn_mode: 0 = kd-tree (the PCL/FLANN cost model), 1 = brute force (cross-check).  use_omp: 1 = the reference's
	// 3-wide OpenMP sections, 0 = serial.
	int mulls_oracle_icp(const mulls_pair *pair, const mulls_params *params, mulls_result *result, int nn_mode, int use_omp)
	{
		if (!pair || !params || !result)
			return MULLS_E_INVALID;
		return icp_impl(pair, params, result, nn_mode, use_omp);
	}

	int mulls_oracle_transform(void *pts, uint32_t n, uint32_t stride, const double T[16])
	{
		Cloud c;
		mulls_cloud mc = {pts, n, stride};
		load_cloud(mc, c);
		M4 m;
		std::memcpy(m.a, T, sizeof(m.a));
		transform_cloud(c, m);
		for (uint32_t i = 0; i < n; i++)
			std::memcpy((uint8_t *)pts + (size_t)i * stride, &c[i], sizeof(Pt));
		return 0;
	}

	// same contract as mulls_motion_compensate (include/mulls_hip.h): CFilter::apply_motion_compensation(pc_in_out, Tran, s_ambigous_thre), cfilter.hpp:470-491
	int mulls_oracle_motion_compensate(void *pts, uint32_t n, uint32_t stride, const double Tran[16], float s_ambigous_thre)
	{
		Cloud c, o;
		mulls_cloud mc = {pts, n, stride};
		load_cloud(mc, c);
		M4 m;
		std::memcpy(m.a, Tran, sizeof(m.a));
		apply_motion_compensation(c, o, m, s_ambigous_thre);
		for (uint32_t i = 0; i < n; i++) // the reference assigns x, y, z of the point it was given: every other byte of the record stays (data[3], the padding)
			std::memcpy((uint8_t *)pts + (size_t)i * stride, &o[i].x, 3 * sizeof(float));
		return 0;
	}

	// same contract as mulls_stage_correspond (include/mulls_hip.h); outputs indexed by ORIGINAL source index
	int mulls_oracle_correspond(const mulls_cloud *src, const mulls_cloud *tgt, float dis_thre, int normal_check, float angle_thre_degree,
								int32_t *match, float *d2, uint8_t *flags, int nn_mode)
	{
		Cloud S, T;
		load_cloud(*src, S);
		load_cloud(*tgt, T);
		std::vector<int> orig(S.size());
		for (size_t i = 0; i < S.size(); i++)
			orig[i] = (int)i;
		KdTree tree;
		if (!nn_mode)
			tree.build(T);
		// raw NN for every source point (match/d2 outputs)
		const double maxd = (double)(2.5f * dis_thre);
		for (size_t s = 0; s < S.size(); s++)
		{
			float dd;
			int t = (T.size() < 3 || S.size() < 3) ? -1 : (nn_mode ? brute_nearest(T, S[s], dd) : tree.nearest(S[s], dd));
			if (t >= 0 && (double)dd > maxd * maxd)
				t = -1;
			match[s] = t;
			d2[s] = t >= 0 ? dd : 0.0f;
			flags[s] = 0;
		}
		Corrs cf;
		size_t n0 = S.size();
		g_rejector_strict = true; // the stage entry point has no parameter block: the default form (PCL's `<`)
		determine_corres(S, orig, T, &tree, dis_thre, cf, false, normal_check != 0, angle_thre_degree, nn_mode != 0);
		(void)n0;
		for (size_t k = 0; k < orig.size(); k++) // identity when no compaction happened
			flags[orig[k]] |= 1;
		for (size_t k = 0; k < cf.size(); k++)
			flags[orig[cf[k].q]] |= 2;
		return 0;
	}

	int mulls_oracle_accumulate(int metric, const mulls_cloud *src, const mulls_cloud *tgt, const int32_t *corr_src, const int32_t *corr_tgt,
								const float *corr_d2, uint32_t ncorr, int iter_num, float class_weight, int dist_w, int resid_w, int inten_w,
								float window, double *out27, float *weight_out)
	{
		Cloud S, T;
		load_cloud(*src, S);
		load_cloud(*tgt, T);
		Corrs c(ncorr);
		for (uint32_t i = 0; i < ncorr; i++)
		{
			c[i].q = corr_src[i];
			c[i].m = corr_tgt[i];
			c[i].dw = corr_d2 ? corr_d2[i] : 0.0f;
		}
		Normal N;
		N.zero();
		if (metric == 0)
			pt2pl_sum(S, T, c, N, iter_num, class_weight, dist_w, resid_w, inten_w, window);
		else if (metric == 1)
			pt2li_sum(S, T, c, N, iter_num, class_weight, dist_w, resid_w, inten_w, window);
		else
			pt2pt_sum(S, T, c, N, iter_num, class_weight, dist_w, resid_w, inten_w, window);
		int k = 0;
		for (int r = 0; r < 6; r++) // row-major-upper enumeration (r <= cc)
			for (int cc = r; cc < 6; cc++)
				out27[k++] = (metric == 1 && r != cc) ? N.upper[r + 6 * cc] : N.lower[cc + 6 * r];
		for (int j = 0; j < 6; j++)
			out27[21 + j] = N.b[j];
		if (weight_out)
			for (uint32_t i = 0; i < ncorr; i++)
				weight_out[i] = c[i].dw;
		return 0;
	}

	// kd-tree vs brute-force self check helper: nearest neighbour of every src point, no radius
	int mulls_oracle_nn(const mulls_cloud *src, const mulls_cloud *tgt, int32_t *match, float *d2, int nn_mode)
	{
		Cloud S, T;
		load_cloud(*src, S);
		load_cloud(*tgt, T);
		KdTree tree;
		if (!nn_mode)
			tree.build(T);
		for (size_t s = 0; s < S.size(); s++)
			match[s] = nn_mode ? brute_nearest(T, S[s], d2[s]) : tree.nearest(S[s], d2[s]);
		return 0;
	}

	// host pieces exposed for unit tests
	void mulls_oracle_construct_trans(const double x[6], double T[16])
	{
		M4 m;
		construct_trans_a(x, m);
		std::memcpy(T, m.a, sizeof(m.a));
	}
	int mulls_oracle_solve(const double atpa[36], const double atpb[6], double x[6], double cof[36])
	{
		M6 A, C;
		std::memcpy(A.a, atpa, sizeof(A.a));
		bool ok = solve_normal(A, atpb, x, C);
		std::memcpy(cof, C.a, sizeof(C.a));
		return ok ? 0 : 1;
	}
	double mulls_oracle_rotation_angle(const double T[16])
	{
		M4 m;
		std::memcpy(m.a, T, sizeof(m.a));
		return angle_of_rotation(m);
	}
}

// ---------------------------------------------------------------------------------------------------------
// MapManager::update_local_map (src/map_manager.cpp:18-140) and map_based_dynamic_close_removal (:149-256),
// restated on plain clouds.  Class order as everywhere: ground, pillar, facade, beam, roof, vertex.
namespace
{
// CFilter::dist_filter(cloud, xy_dis_thre, keep_inside = true) (cfilter.hpp:834-871): float products summed in float,
// widened to double; z compared against +-DBL_MAX (drops NaN / inf heights)
void dist_filter_inside(Cloud &c, double xy_dis_thre)
{
	Cloud out;
	for (size_t i = 0; i < c.size(); i++)
	{
		const double dis_square = c[i].x * c[i].x + c[i].y * c[i].y;
		if (dis_square < xy_dis_thre * xy_dis_thre && c[i].z < DBL_MAX && c[i].z > -DBL_MAX)
			out.push_back(c[i]);
	}
	c.swap(out);
}

// pcl::transformPointCloud<PointT,double> (positions only)
void transform_positions(Cloud &c, const M4 &T)
{
	for (size_t i = 0; i < c.size(); i++)
	{
		Pt &p = c[i];
		const double x = p.x, y = p.y, z = p.z;
		p.x = static_cast<float>(T(0, 0) * x + T(0, 1) * y + T(0, 2) * z + T(0, 3));
		p.y = static_cast<float>(T(1, 0) * x + T(1, 1) * y + T(1, 2) * z + T(1, 3));
		p.z = static_cast<float>(T(2, 0) * x + T(2, 1) * y + T(2, 2) * z + T(2, 3));
	}
}

// MapManager::map_scan_feature_pts_distance_removal (map_manager.cpp:233-268); `tree` holds what mm_lls_icp indexed
void distance_removal(Cloud &pts, const Cloud &tree_pts, float center_radius, float dmin, float dmax, float near)
{
	if (pts.size() <= 10 || tree_pts.empty())
		return; // the reference would query an empty kd-tree here (undefined); defined as "leave the cloud alone"
	Cloud out;
	for (size_t i = 0; i < pts.size(); i++)
	{
		if (pts[i].x * pts[i].x + pts[i].y * pts[i].y > center_radius * center_radius)
			out.push_back(pts[i]);
		else
		{
			float d2;
			brute_nearest(tree_pts, pts[i], d2);
			if ((d2 > near * near && d2 < dmin * dmin) || d2 > dmax * dmax)
				out.push_back(pts[i]);
		}
	}
	pts.swap(out);
}
// ---------------------------------------------------------------------------------------------------------
// PrincipleComponentAnalysis (pca.hpp): get_pc_pca_feature in its two overloads (:209-290 without, :292-352 with a kd-tree
// argument), get_pca_feature (:392-437), assign_normal (:440-456).  What they call of PCL / FLANN / Eigen is in pcl_restated.h.
struct PcaFeature // pca_feature_t (pca.hpp:18-47): eigenvalues and the ratios derived from them are doubles, the directions floats
{
	double lamada1 = 0, lamada2 = 0, lamada3 = 0, curvature = 0, linear_2 = 0, planar_2 = 0, spherical_2 = 0;
	float principal[3] = {0, 0, 0}, normal[3] = {0, 0, 0};
	int pt_num = 0;
	std::vector<int> neighbor_indices;
	std::vector<char> close_to_query_point;
};
bool pca_feature(const Cloud &c, std::vector<int> &idx, PcaFeature &f) // get_pca_feature (:392-437)
{
	const int pt_num = (int)idx.size();
	if (pt_num <= 3)
		return false;
	float eval[3], evec[3][3];
	restated::pca(c, idx, eval, evec);
	for (int k = 0; k < 3; k++)
	{
		f.principal[k] = evec[k][0];
		f.normal[k] = evec[k][2];
	}
	restated::normalize3(f.principal);
	restated::normalize3(f.normal);
	f.lamada1 = eval[0];
	f.lamada2 = eval[1];
	f.lamada3 = eval[2];
	if ((f.lamada1 + f.lamada2 + f.lamada3) == 0)
		f.curvature = 0;
	else
		f.curvature = f.lamada3 / (f.lamada1 + f.lamada2 + f.lamada3);
	f.linear_2 = ((f.lamada1) - (f.lamada2)) / (f.lamada1);
	f.planar_2 = ((f.lamada2) - (f.lamada3)) / (f.lamada1);
	f.spherical_2 = (f.lamada3) / (f.lamada1);
	idx.swap(f.neighbor_indices);
	return true;
}
void assign_normal(Pt &pt, const PcaFeature &f, bool is_plane_feature = true) // :440-456
{
	const float *d = is_plane_feature ? f.normal : f.principal;
	pt.nx = d[0];
	pt.ny = d[1];
	pt.nz = d[2];
	pt.n3 = (float)(is_plane_feature ? f.planar_2 : f.linear_2);
}
// with_tree: the overload taking the kd-tree (:292-352; distance-adaptive radius from the 3-D range, square-root law) — the other one
// (:209-290) adapts linearly to the planar range and is "deprecated" there
void get_pc_pca_feature(Cloud &cloud, std::vector<PcaFeature> &features, bool with_tree, float radius, int nearest_k, int min_k, int pca_down_rate,
						bool distance_adaptive_on, float unit_dist)
{
	features.assign(cloud.size(), PcaFeature());
	const Cloud positions = cloud; // the loop writes normals into the cloud it searches: only x, y, z are ever read back
	restated::RadiusIndex<Pt> tree;
	tree.build(positions, radius);
	if (pca_down_rate < 1)
		pca_down_rate = 1; // upstream: i += 0 never ends
	std::vector<int> search_indices;
	std::vector<float> squared_distances;
	for (size_t i = 0; i < cloud.size(); i += (size_t)pca_down_rate)
	{
		float neighborhood_r = radius;
		if (distance_adaptive_on)
		{
			if (with_tree)
			{
				double dist = std::sqrt(cloud[i].x * cloud[i].x + cloud[i].y * cloud[i].y + cloud[i].z * cloud[i].z);
				if (dist > unit_dist)
					neighborhood_r = std::sqrt(dist / unit_dist) * radius;
			}
			else
			{
				double dist = std::sqrt(cloud[i].x * cloud[i].x + cloud[i].y * cloud[i].y);
				const double scaled = dist / unit_dist * radius;
				neighborhood_r = (radius > scaled) ? radius : scaled; // max_(radius, dist / unit_dist * radius)
			}
		}
		tree.search(positions[i], neighborhood_r, (unsigned)nearest_k, search_indices, squared_distances);
		PcaFeature &f = features[i];
		f.pt_num = (int)search_indices.size();
		f.close_to_query_point.resize(search_indices.size());
		for (size_t j = 0; j < search_indices.size(); j++)
			f.close_to_query_point[j] = squared_distances[j] < 0.64 * radius * radius;
		pca_feature(positions, search_indices, f);
		if (f.pt_num > min_k)
			assign_normal(cloud[i], f);
	}
}
// ---------------------------------------------------------------------------------------------------------
// CFilter::classify_nground_pts (cfilter.hpp:2058-2290) and what it calls: encode_stable_points (:1071-1181), non_max_suppress
// (:1243-1311), xy_normal_balanced_downsample (:551-602), random_downsample_pcl (:606-628).
std::atomic<unsigned long long> g_nms_ties{0}; // pairs of neighbours in non_max_suppress's visiting order with equal normal[3]
// classify_nground_pts' decisions against Eigen's accuracy: [0] points with a decision (enough neighbours), [1] / [2] those of them where an eigenvalue ratio or an eigenvector
// component the chain compares lies within 1e-5 / 1e-4 of the threshold it is compared with (upstream's SelfAdjointEigenSolver<Matrix3f> is good to ~1e-6 of the largest
// eigenvalue; pcl_restated.h decomposes in double) — how many labels COULD differ from a run with real PCL
std::atomic<unsigned long long> g_cls_census[3];

// non_max_suppress(cloud_in, cloud_out, nms_radius).  The visiting order is std::sort's by normal[3] descending: among equal keys — a few
// hundred per scan, points of one small cluster share their neighbourhood — it is whatever this toolchain's introsort leaves, upstream as
// here (the permutation depends on the keys and the element count only, not on what is attached to a key)
bool non_max_suppress(Cloud &cloud_in, Cloud &cloud_out, float nms_radius)
{
	const int pt_count_before = (int)cloud_in.size();
	if (pt_count_before < 10)
		return false;
	std::sort(cloud_in.begin(), cloud_in.end(), [](const Pt &a, const Pt &b) { return a.n3 > b.n3; });
	for (int i = 1; i < pt_count_before; i++)
		if (cloud_in[i].n3 == cloud_in[i - 1].n3)
			g_nms_ties++;
	std::vector<char> visited(pt_count_before, 0);
	restated::RadiusIndex<Pt> tree;
	tree.build(cloud_in, nms_radius);
	std::vector<int> search_indices;
	std::vector<float> distances;
	for (int id = 0; id < pt_count_before; id++) // *unVisitedPtId.begin(): the lowest id not yet erased
	{
		if (visited[id])
			continue;
		cloud_out.push_back(cloud_in[id]);
		visited[id] = 1;
		tree.search(cloud_in[id], nms_radius, 0, search_indices, distances);
		for (size_t i = 0; i < search_indices.size(); i++)
			visited[search_indices[i]] = 1;
	}
	return true;
}
// xy_normal_balanced_downsample(cloud_in_out, keep_number_per_sector, sector_num)
bool xy_normal_balanced_downsample(Cloud &cloud, int keep_number_per_sector, int sector_num, uint64_t seed, int cloud_id)
{
	if (keep_number_per_sector < 0 || (long)cloud.size() <= (long)keep_number_per_sector) // size_t comparison upstream (:554)
		return false;
	std::vector<Cloud> sectors(sector_num);
	const double angle_per_sector = 360.0 / sector_num;
	for (size_t i = 0; i < cloud.size(); i++)
	{
		double ang = std::atan2(cloud[i].ny, cloud[i].nx); // the float overload
		if (ang < 0)
			ang += 2 * M_PI;
		ang *= (180.0 / M_PI);
		int sector_id = (int)(ang / angle_per_sector);
		if (sector_id >= sector_num) // -tiny + 2 pi rounds to 2 pi: upstream writes past the last sector; the ABI puts the point into the last one
			sector_id = sector_num - 1;
		if (sector_id < 0) // NaN normals: (int)NaN is INT_MIN on x86
			sector_id = 0;
		sectors[sector_id].push_back(cloud[i]);
	}
	Cloud temp;
	for (int j = 0; j < sector_num; j++)
	{
		random_downsample(sectors[j], keep_number_per_sector, seed, cloud_id + j);
		temp.insert(temp.end(), sectors[j].begin(), sectors[j].end());
	}
	temp.swap(cloud);
	return true;
}
// encode_stable_points(cloud_in, cloud_out, features, index_with_feature, min_curvature, min_feature_point_num_neighborhood, min_point_num_neighborhood)
void encode_stable_points(const Cloud &cloud_in, Cloud &cloud_out, const std::vector<PcaFeature> &features, const std::vector<int> &index_with_feature,
						  float min_curvature, int min_feature_point_num_neighborhood, int min_point_num_neighborhood)
{
	for (size_t i = 0; i < features.size(); ++i)
	{
		if (features[i].pt_num > min_point_num_neighborhood && features[i].curvature > min_curvature)
		{
			float accu_intensity = 0.0;
			Pt pt = cloud_in[i];
			pt.n3 = features[i].curvature;
			int cnt[5] = {0, 0, 0, 0, 0}, close_cnt[5] = {0, 0, 0, 0, 0}, far_cnt[5] = {0, 0, 0, 0, 0};
			const int neighbor_total_count = (int)features[i].neighbor_indices.size();
			for (int j = 0; j < neighbor_total_count; j++)
			{
				const int nb = features[i].neighbor_indices[j];
				const int lab = index_with_feature[nb];
				if (lab >= 1 && lab <= 4)
				{
					cnt[lab]++;
					if (features[i].close_to_query_point[j])
						close_cnt[lab]++;
					else
						far_cnt[lab]++;
				}
				accu_intensity += cloud_in[nb].intensity;
			}
			if (cnt[1] + cnt[2] + cnt[3] + cnt[4] < min_feature_point_num_neighborhood)
				continue;
			if (neighbor_total_count == 0) // only with neigh_k_min < 3 and a non-positive minimum: upstream divides by zero below
				continue;
			for (int l = 1; l <= 4; l++)
			{
				cnt[l] = 100 * cnt[l] / neighbor_total_count;
				close_cnt[l] = 100 * close_cnt[l] / neighbor_total_count;
				far_cnt[l] = 100 * far_cnt[l] / neighbor_total_count;
			}
			const int descriptor = cnt[1] * 1000000 + cnt[2] * 10000 + cnt[3] * 100 + cnt[4];
			const int descriptor_1 = close_cnt[1] * 1000000 + close_cnt[2] * 10000 + close_cnt[3] * 100 + close_cnt[4];
			const int descriptor_2 = far_cnt[1] * 1000000 + far_cnt[2] * 10000 + far_cnt[3] * 100 + far_cnt[4];
			pt.curvature = descriptor;
			pt.nx = descriptor_1;
			pt.ny = descriptor_2;
			pt.intensity = accu_intensity / neighbor_total_count;
			cloud_out.push_back(pt);
		}
	}
}
// clouds: enum mulls_classify_cloud; cloud_in is modified as upstream modifies it (normals of the queried points)
int classify_nground_impl(Cloud &cloud_in, const mulls_classify_params &P, Cloud out[MULLS_CL_COUNT])
{
	Cloud &cloud_pillar = out[MULLS_CL_PILLAR], &cloud_beam = out[MULLS_CL_BEAM], &cloud_facade = out[MULLS_CL_FACADE], &cloud_roof = out[MULLS_CL_ROOF];
	Cloud &cloud_pillar_down = out[MULLS_CL_PILLAR_DOWN], &cloud_beam_down = out[MULLS_CL_BEAM_DOWN], &cloud_facade_down = out[MULLS_CL_FACADE_DOWN],
		  &cloud_roof_down = out[MULLS_CL_ROOF_DOWN], &cloud_vertex = out[MULLS_CL_VERTEX];
	const float neighbor_searching_radius = P.neighbor_searching_radius;
	const int neighbor_k = P.neighbor_k, neigh_k_min = P.neigh_k_min, pca_down_rate = P.pca_down_rate;
	if (pca_down_rate < 1 || neighbor_k < 1 || neighbor_k > 64 || !(neighbor_searching_radius > 0))
		return MULLS_E_INVALID;
	const float edge_thre = P.edge_thre, planar_thre = P.planar_thre, edge_thre_down = P.edge_thre_down, planar_thre_down = P.planar_thre_down;
	int extract_vertex_points_method = P.extract_vertex_points_method;
	const float curvature_thre = P.curvature_thre;
	const float linear_vertical_sin_high_thre = P.linear_vertical_sin_high_thre, linear_vertical_sin_low_thre = P.linear_vertical_sin_low_thre;
	const float planar_vertical_sin_high_thre = P.planar_vertical_sin_high_thre, planar_vertical_sin_low_thre = P.planar_vertical_sin_low_thre;
	const bool fixed_num_downsampling = P.fixed_num_downsampling, sharpen_with_nms = P.sharpen_with_nms;
	const float beam_height_max = P.beam_height_max, roof_height_min = P.roof_height_min, feature_pts_ratio_guess = P.feature_pts_ratio_guess;

	if (fixed_num_downsampling)
		random_downsample(cloud_in, P.unground_down_fixed_num, P.rng_seed, 30);

	std::vector<PcaFeature> cloud_features;
	const float unit_distance = 30.0;
	get_pc_pca_feature(cloud_in, cloud_features, true, neighbor_searching_radius, neighbor_k, 1, pca_down_rate, P.use_distance_adaptive_pca, unit_distance);

	std::vector<int> index_with_feature(cloud_in.size(), 0); // 0 - not special points, 1 - pillar, 2 - beam, 3 - facade, 4 - roof
	for (size_t i = 0; i < cloud_in.size(); i++)
	{
		const PcaFeature &f = cloud_features[i];
		if (f.pt_num > neigh_k_min)
		{
			{
				const double pz = std::abs((double)f.principal[2]), nz = std::abs((double)f.normal[2]);
				const double gaps[] = {std::abs(f.linear_2 - edge_thre),	 std::abs(f.linear_2 - edge_thre_down),	   std::abs(f.planar_2 - planar_thre),
									   std::abs(f.planar_2 - planar_thre_down), std::abs(pz - linear_vertical_sin_high_thre), std::abs(pz - linear_vertical_sin_low_thre),
									   std::abs(nz - planar_vertical_sin_high_thre), std::abs(nz - planar_vertical_sin_low_thre)};
				double g = 1e300;
				for (double v : gaps)
					g = std::min(g, v);
				g_cls_census[0]++;
				if (g < 1e-5)
					g_cls_census[1]++;
				if (g < 1e-4)
					g_cls_census[2]++;
			}
			if (f.linear_2 > edge_thre)
			{
				if (std::abs(f.principal[2]) > linear_vertical_sin_high_thre)
				{
					assign_normal(cloud_in[i], f, false);
					cloud_pillar.push_back(cloud_in[i]);
					index_with_feature[i] = 1;
				}
				else if (std::abs(f.principal[2]) < linear_vertical_sin_low_thre && cloud_in[i].z < beam_height_max)
				{
					assign_normal(cloud_in[i], f, false);
					cloud_beam.push_back(cloud_in[i]);
					index_with_feature[i] = 2;
				}
				if (!sharpen_with_nms && f.linear_2 > edge_thre_down)
				{
					if (std::abs(f.principal[2]) > linear_vertical_sin_high_thre)
						cloud_pillar_down.push_back(cloud_in[i]);
					else if (std::abs(f.principal[2]) < linear_vertical_sin_low_thre && cloud_in[i].z < beam_height_max)
						cloud_beam_down.push_back(cloud_in[i]);
				}
			}
			else if (f.planar_2 > planar_thre)
			{
				if (std::abs(f.normal[2]) > planar_vertical_sin_high_thre && cloud_in[i].z > roof_height_min)
				{
					assign_normal(cloud_in[i], f, true);
					cloud_roof.push_back(cloud_in[i]);
					index_with_feature[i] = 4;
				}
				else if (std::abs(f.normal[2]) < planar_vertical_sin_low_thre)
				{
					assign_normal(cloud_in[i], f, true);
					cloud_facade.push_back(cloud_in[i]);
					index_with_feature[i] = 3;
				}
				if (!sharpen_with_nms && f.planar_2 > planar_thre_down)
				{
					if (std::abs(f.normal[2]) > planar_vertical_sin_high_thre && cloud_in[i].z > roof_height_min)
						cloud_roof_down.push_back(cloud_in[i]);
					else if (std::abs(f.normal[2]) < planar_vertical_sin_low_thre)
						cloud_facade_down.push_back(cloud_in[i]);
				}
			}
		}
	}

	if (curvature_thre < 1e-8)
		extract_vertex_points_method = 0;
	if (extract_vertex_points_method == 2) // high-curvature points among the neighbourhood of geometric feature points; the labels grow while the loop runs
	{
		const float vertex_feature_ratio_thre = feature_pts_ratio_guess / pca_down_rate;
		for (size_t i = 0; i < cloud_in.size(); i++)
		{
			const PcaFeature &f = cloud_features[i];
			if (index_with_feature[i] == 0 && f.pt_num > neigh_k_min && f.curvature > curvature_thre)
			{
				int geo_feature_point_count = 0;
				for (size_t j = 0; j < f.neighbor_indices.size(); j++)
					if (index_with_feature[f.neighbor_indices[j]])
						geo_feature_point_count++;
				if (1.0 * geo_feature_point_count / f.pt_num > vertex_feature_ratio_thre)
				{
					assign_normal(cloud_in[i], f, false);
					cloud_in[i].n3 = 5.0 * f.curvature;
					if (std::abs(f.principal[2]) > linear_vertical_sin_high_thre)
					{
						cloud_pillar.push_back(cloud_in[i]);
						index_with_feature[i] = 1;
					}
					else if (std::abs(f.principal[2]) < linear_vertical_sin_low_thre && cloud_in[i].z < beam_height_max)
					{
						cloud_beam.push_back(cloud_in[i]);
						index_with_feature[i] = 2;
					}
				}
			}
		}
	}
	const int min_neighbor_feature_pts = (int)(feature_pts_ratio_guess / pca_down_rate * neighbor_k) - 1;
	encode_stable_points(cloud_in, cloud_vertex, cloud_features, index_with_feature, 0.3 * curvature_thre, min_neighbor_feature_pts, neigh_k_min);

	if (sharpen_with_nms)
	{
		const float nms_radius = 0.25 * neighbor_searching_radius;
		if (P.pillar_down_fixed_num > 0)
			non_max_suppress(cloud_pillar, cloud_pillar_down, nms_radius);
		if (P.facade_down_fixed_num > 0)
			non_max_suppress(cloud_facade, cloud_facade_down, nms_radius);
		if (P.beam_down_fixed_num > 0)
			non_max_suppress(cloud_beam, cloud_beam_down, nms_radius);
		if (P.roof_down_fixed_num > 0)
			non_max_suppress(cloud_roof, cloud_roof_down, nms_radius);
	}
	if (fixed_num_downsampling)
	{
		random_downsample(cloud_pillar_down, P.pillar_down_fixed_num, P.rng_seed, 31);
		const int sector_num = 4;
		xy_normal_balanced_downsample(cloud_facade_down, (int)(P.facade_down_fixed_num / sector_num), sector_num, P.rng_seed, 32);
		xy_normal_balanced_downsample(cloud_beam_down, (int)(P.beam_down_fixed_num / sector_num), sector_num, P.rng_seed, 36);
		random_downsample(cloud_roof_down, P.roof_down_fixed_num, P.rng_seed, 40);
	}
	return MULLS_OK;
}

// MapManager::update_cloud_vectors (src/map_manager.cpp:258-292), the PCA refresh of the local map's linear features
// (recalculate_feature_on, :98-118)
void update_cloud_vectors(Cloud &pts, float pca_radius, int pca_k, int k_min, float sin_low, float sin_high, float min_linearity)
{
	if (pts.size() == 0)
		return;
	std::vector<PcaFeature> pca_features;
	Cloud temp;
	get_pc_pca_feature(pts, pca_features, false, pca_radius, pca_k, k_min, 1, false, 35.0f);
	for (size_t i = 0; i < pts.size(); i++)
	{
		if (pca_features[i].pt_num >= k_min && pca_features[i].linear_2 > min_linearity)
		{
			assign_normal(pts[i], pca_features[i], false);
			if (std::abs(pca_features[i].principal[2]) > sin_high || std::abs(pca_features[i].principal[2]) < sin_low)
			{
				pts[i].curvature = pca_features[i].linear_2;
				temp.push_back(pts[i]);
			}
		}
	}
	pts.swap(temp);
}
} // namespace

extern "C"
{
	// CFilter::scanner_filter (cfilter.hpp:914-929): points on the ego vehicle and underground ghost points near the scanner are dropped
	int mulls_oracle_scanner_filter(const void *pts, uint32_t n, uint32_t stride, float self_radius, float ghost_radius, float z_min_thre_ghost,
									float z_min_thre_global, void *out, uint32_t cap, uint32_t *n_out)
	{
		uint32_t w = 0;
		for (uint32_t i = 0; i < n; i++)
		{
			Pt p;
			std::memcpy(&p, (const unsigned char *)pts + (size_t)i * stride, sizeof(Pt));
			float dis_square = p.x * p.x + p.y * p.y;
			if (dis_square > self_radius * self_radius && p.z > z_min_thre_global)
			{
				if (dis_square > ghost_radius * ghost_radius || p.z > z_min_thre_ghost)
				{
					if (w < cap)
						std::memcpy((unsigned char *)out + (size_t)w * sizeof(Pt), &p, sizeof(Pt));
					w++;
				}
			}
		}
		*n_out = w;
		return MULLS_OK;
	}
	// CFilter::dist_filter(cloud, xy_dist_min, xy_dist_max) (cfilter.hpp:806-832), the pass test/mulls_slam.cpp:359-360 / :404-405 runs on pc_raw
	// under --apply_dist_filter: the squared horizontal range is a float expression widened to double, the limits are doubles
	int mulls_oracle_dist_filter(const void *pts, uint32_t n, uint32_t stride, double xy_dist_min, double xy_dist_max, void *out, uint32_t cap, uint32_t *n_out)
	{
		uint32_t w = 0;
		for (uint32_t i = 0; i < n; i++)
		{
			Pt p;
			std::memcpy(&p, (const unsigned char *)pts + (size_t)i * stride, sizeof(Pt));
			double dis_square = p.x * p.x + p.y * p.y;
			if (dis_square < xy_dist_max * xy_dist_max && dis_square > xy_dist_min * xy_dist_min)
			{
				if (w < cap)
					std::memcpy((unsigned char *)out + (size_t)w * sizeof(Pt), &p, sizeof(Pt));
				w++;
			}
		}
		*n_out = w;
		return MULLS_OK;
	}
	// CFilter::voxel_downsample (cfilter.hpp:83-160): one point per occupied voxel, the first of its voxel in the order std::sort leaves the
	// (voxel, index) pairs in (the comparison looks at the voxel only, :42), voxels in increasing index.  Below 0.001 m the cloud is handed on as
	// it is (:90-97).  Beyond 2^21 voxels along an axis the call is refused (the reference's index arithmetic would start to depend on whether
	// ceil / floor resolve to the float or the double overload, and the 64-bit index could wrap).
	int mulls_oracle_voxel_downsample(const void *pts, uint32_t n, uint32_t stride, float voxel_size, void *out, uint32_t cap, uint32_t *n_out)
	{
		std::vector<Pt> in(n);
		for (uint32_t i = 0; i < n; i++)
			std::memcpy(&in[i], (const unsigned char *)pts + (size_t)i * stride, sizeof(Pt));
		*n_out = 0;
		if (voxel_size < 0.001)
		{
			*n_out = n;
			if (std::min(n, cap))
				std::memcpy(out, in.data(), (size_t)std::min(n, cap) * sizeof(Pt));
			return MULLS_OK;
		}
		if (n == 0)
			return MULLS_OK;
		float inverse_voxel_size = 1.0f / voxel_size;
		float min_p[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, max_p[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX}; // pcl::getMinMax3D
		for (uint32_t i = 0; i < n; i++)
		{
			const float v[3] = {in[i].x, in[i].y, in[i].z};
			if (!std::isfinite(v[0]) || !std::isfinite(v[1]) || !std::isfinite(v[2]))
				return MULLS_E_INVALID;
			for (int k = 0; k < 3; k++)
				min_p[k] = std::min(min_p[k], v[k]), max_p[k] = std::max(max_p[k], v[k]);
		}
		float gap_p[3] = {max_p[0] - min_p[0], max_p[1] - min_p[1], max_p[2] - min_p[2]};
		for (int k = 0; k < 3; k++)
			if (!(std::ceil((double)(gap_p[k] * inverse_voxel_size)) + 1 < 2097152.0))
				return MULLS_E_UNSUPPORTED;
		unsigned long long max_vy = ceil(gap_p[1] * inverse_voxel_size) + 1;
		unsigned long long max_vz = ceil(gap_p[2] * inverse_voxel_size) + 1;
		unsigned long long mul_vx = max_vy * max_vz;
		unsigned long long mul_vy = max_vz;
		unsigned long long mul_vz = 1;
		struct IdPair
		{
			unsigned long long voxel_idx;
			int idx;
			bool operator<(const IdPair &o) const { return voxel_idx < o.voxel_idx; }
		};
		std::vector<IdPair> id_pairs(n);
		for (uint32_t i = 0; i < n; i++)
		{
			unsigned long long vx = floor((in[i].x - min_p[0]) * inverse_voxel_size);
			unsigned long long vy = floor((in[i].y - min_p[1]) * inverse_voxel_size);
			unsigned long long vz = floor((in[i].z - min_p[2]) * inverse_voxel_size);
			id_pairs[i].idx = (int)i;
			id_pairs[i].voxel_idx = vx * mul_vx + vy * mul_vy + vz * mul_vz;
		}
		std::sort(id_pairs.begin(), id_pairs.end());
		uint32_t w = 0;
		size_t begin_id = 0;
		while (begin_id < id_pairs.size())
		{
			if (w < cap)
				std::memcpy((unsigned char *)out + (size_t)w * sizeof(Pt), &in[id_pairs[begin_id].idx], sizeof(Pt));
			w++;
			size_t compare_id = begin_id + 1;
			while (compare_id < id_pairs.size() && id_pairs[begin_id].voxel_idx == id_pairs[compare_id].voxel_idx)
				compare_id++;
			begin_id = compare_id;
		}
		*n_out = w;
		return MULLS_OK;
	}
	void mulls_oracle_classify_default_params(mulls_classify_params *p)
	{
		std::memset(p, 0, sizeof(*p));
		p->neighbor_searching_radius = 1.0f;
		p->neighbor_k = 50;
		p->neigh_k_min = 8;
		p->pca_down_rate = 1;
		p->edge_thre = 0.65f;
		p->planar_thre = 0.65f;
		p->edge_thre_down = 0.75f;
		p->planar_thre_down = 0.75f;
		p->extract_vertex_points_method = 2;
		p->curvature_thre = 0.10f;
		p->vertex_curvature_non_max_radius = 1.5f;
		p->linear_vertical_sin_high_thre = 0.94f;
		p->linear_vertical_sin_low_thre = 0.17f;
		p->planar_vertical_sin_high_thre = 0.98f;
		p->planar_vertical_sin_low_thre = 0.34f;
		p->sharpen_with_nms = 1;
		p->pillar_down_fixed_num = 200;
		p->facade_down_fixed_num = 800;
		p->beam_down_fixed_num = 200;
		p->roof_down_fixed_num = 200;
		p->unground_down_fixed_num = 20000;
		p->beam_height_max = FLT_MAX;
		p->roof_height_min = 0.0f;
		p->feature_pts_ratio_guess = 0.3f;
	}
	// same contract as mulls_classify_nground; additionally `in_after` (n records or NULL): cloud_in as the function leaves it
	int mulls_oracle_classify_nground(const void *pts, uint32_t n, uint32_t stride, const mulls_classify_params *P, void *const out[MULLS_CL_COUNT],
									  const uint32_t cap[MULLS_CL_COUNT], uint32_t n_out[MULLS_CL_COUNT], void *in_after, uint32_t *n_in_after)
	{
		Cloud in(n);
		for (uint32_t i = 0; i < n; i++)
			std::memcpy(&in[i], (const unsigned char *)pts + (size_t)i * stride, sizeof(Pt));
		Cloud o[MULLS_CL_COUNT];
		const int rc = classify_nground_impl(in, *P, o);
		if (rc != MULLS_OK)
			return rc;
		for (int k = 0; k < MULLS_CL_COUNT; k++)
		{
			n_out[k] = (uint32_t)o[k].size();
			const size_t m = std::min<size_t>(o[k].size(), cap[k]);
			if (m && out[k])
				std::memcpy(out[k], o[k].data(), m * sizeof(Pt));
		}
		if (n_in_after)
			*n_in_after = (uint32_t)in.size();
		if (in_after && in.size())
			std::memcpy(in_after, in.data(), in.size() * sizeof(Pt));
		return MULLS_OK;
	}
	// classify_nground_pts' threshold margins (g_cls_census): out[0..2]; reset != 0 clears them afterwards
	void mulls_oracle_classify_census(unsigned long long out[3], int reset)
	{
		for (int k = 0; k < 3; k++)
		{
			out[k] = g_cls_census[k];
			if (reset)
				g_cls_census[k] = 0;
		}
	}

	// restated::search_census (pcl_restated.h): out[0..3]; reset != 0 clears it afterwards
	void mulls_oracle_search_census(unsigned long long out[4], int reset)
	{
		for (int k = 0; k < 4; k++)
		{
			out[k] = restated::search_census(k);
			if (reset)
				restated::search_census(k) = 0;
		}
	}

	unsigned long long mulls_oracle_nms_ties(int reset)
	{
		const unsigned long long v = g_nms_ties;
		if (reset)
			g_nms_ties = 0;
		return v;
	}
	void mulls_oracle_map_default_params(mulls_map_params *p)
	{
		std::memset(p, 0, sizeof(*p));
		p->local_map_radius = 80;
		p->max_num_pts = 20000;
		p->kept_vertex_num = 800;
		p->last_frame_reliable_radius = 60;
		std::strcpy(p->used_feature_type, "111110");
		p->dynamic_removal_center_radius = 30.0f;
		p->dynamic_dist_thre_min = 0.3f;
		p->dynamic_dist_thre_max = 3.0f;
		p->near_dist_thre = 0.03f;
		std::strcpy(p->tree_used, "000000");
	}

	// map_out[c] / frame_out[c]: caller buffers of (map_in[c].n + frame_down[c].n) and frame_down[c].n 48-B records
	int mulls_oracle_map_update(const mulls_cloud map_in[6], const double map_pose[16], const mulls_cloud frame_down[6],
								const double frame_pose[16], const mulls_map_params *P, void *const map_out[6], uint32_t map_out_n[6],
								void *const frame_out[6], uint32_t frame_out_n[6], mulls_map_report *rep)
	{
		if (!map_in || !map_pose || !frame_down || !frame_pose || !P || !rep)
			return MULLS_E_INVALID;
		Cloud M[6], F[6];
		for (int c = 0; c < 6; c++)
		{
			load_cloud(map_in[c], M[c]);
			load_cloud(frame_down[c], F[c]);
		}
		M4 map_T, frame_T;
		std::memcpy(map_T.a, map_pose, sizeof(map_T.a));
		std::memcpy(frame_T.a, frame_pose, sizeof(frame_T.a));
		const M4 tran_target_map = m4_mul(m4_inverse(frame_T), map_T); // :28
		const M4 inv = m4_inverse(tran_target_map);
		for (int c = 0; c < 5; c++) // transform_feature(inv, true, false): the five *_down clouds, not the vertex cloud (:32)
			transform_cloud(F[c], inv);

		float dmax = P->dynamic_dist_thre_max;
		{
			const double lo = P->dynamic_dist_thre_min + 0.1; // max_(a, b) on (float, double) operands (:34)
			dmax = (float)(((double)dmax > lo) ? (double)dmax : lo);
		}
		int feature_point_num = 0;
		{
			static const int five[5] = {MULLS_GROUND, MULLS_FACADE, MULLS_ROOF, MULLS_PILLAR, MULLS_BEAM};
			for (int k = 0; k < 5; k++)
				feature_point_num += (int)M[five[k]].size();
		}
		rep->dynamic_removal_ran = 0;
		if (P->map_based_dynamic_removal_on && feature_point_num > P->max_num_pts / 5 && P->tree_mode != 0) // :37
		{
			rep->dynamic_removal_ran = 1;
			static const int order[3] = {MULLS_PILLAR, MULLS_BEAM, MULLS_FACADE};
			for (int k = 0; k < 3; k++)
			{
				const int c = order[k];
				if (P->used_feature_type[c] != '1' || P->tree_used[c] != '1')
					continue;
				Cloud tree_pts = M[c];
				if (P->tree_mode == 2)
					bbx_filter(tree_pts, P->tree_box);
				distance_removal(F[c], tree_pts, P->dynamic_removal_center_radius, P->dynamic_dist_thre_min, dmax, P->near_dist_thre);
			}
		}
		for (int c = 0; c < 6; c++)
		{
			frame_out_n[c] = (uint32_t)F[c].size();
			rep->frame_n[c] = frame_out_n[c];
			if (frame_out && frame_out[c])
				for (size_t i = 0; i < F[c].size(); i++)
					std::memcpy((uint8_t *)frame_out[c] + i * sizeof(Pt), &F[c][i], sizeof(Pt));
		}
		for (int c = 0; c < 6; c++) // append_feature(last_target, true, used) (:54, utility.hpp:438-469): the vertex cloud always
			if (c == MULLS_VERTEX || P->used_feature_type[c] == '1')
				M[c].insert(M[c].end(), F[c].begin(), F[c].end());
		for (int c = 0; c < 6; c++) // transform_feature(tran_target_map, false): all six undown clouds (:57)
			transform_cloud(M[c], tran_target_map);
		for (int c = 0; c < 6; c++) // :62-67
			dist_filter_inside(M[c], P->local_map_radius);
		const int cur = (int)(M[MULLS_GROUND].size() + M[MULLS_FACADE].size() + M[MULLS_ROOF].size() + M[MULLS_PILLAR].size() +
							  M[MULLS_BEAM].size());
		for (int c = 0; c < 5; c++) // :75-85 (a division by zero points makes kept = INT_MIN: nothing to thin anyway)
		{
			const int kept = cur > 0 ? (int)(1.0 * P->max_num_pts / cur * M[c].size() + 1) : 1;
			random_downsample(M[c], kept, P->rng_seed, 20 + c);
		}
		random_downsample(M[MULLS_VERTEX], P->kept_vertex_num, P->rng_seed, 20 + MULLS_VERTEX);

		Cloud raw; // merge_feature_points(pc_raw, false): ground, facade, pillar, beam, roof, vertex (utility.hpp:471-481)
		static const int merge_order[6] = {MULLS_GROUND, MULLS_FACADE, MULLS_PILLAR, MULLS_BEAM, MULLS_ROOF, MULLS_VERTEX};
		for (int k = 0; k < 6; k++)
			raw.insert(raw.end(), M[merge_order[k]].begin(), M[merge_order[k]].end());
		cloud_bbx(raw, rep->local_bound);
		transform_positions(raw, frame_T); // local_map->pose_lo = last_target_cblock->pose_lo (:58), then :92
		cloud_bbx(raw, rep->bound);

		if (P->recalculate_feature_on) // :98-118: pca_radius 1.8, pca_max_k 20, pca_min_k 6, pillars steeper than 55 deg, beams flatter than 15 deg
		{
			if (P->used_feature_type[1] == '1')
				update_cloud_vectors(M[MULLS_PILLAR], 1.8f, 20, 6, 0.0f, 0.80f, 0.65f);
			if (P->used_feature_type[3] == '1')
				update_cloud_vectors(M[MULLS_BEAM], 1.8f, 20, 6, 0.25f, 1.0f, 0.65f);
		}
		rep->feature_point_num = (int)(M[MULLS_GROUND].size() + M[MULLS_FACADE].size() + M[MULLS_ROOF].size() + M[MULLS_PILLAR].size() +
									   M[MULLS_BEAM].size());
		for (int c = 0; c < 6; c++)
		{
			map_out_n[c] = (uint32_t)M[c].size();
			rep->n[c] = map_out_n[c];
			if (map_out && map_out[c])
				for (size_t i = 0; i < M[c].size(); i++)
					std::memcpy((uint8_t *)map_out[c] + i * sizeof(Pt), &M[c][i], sizeof(Pt));
		}
		rep->ms_total = 0.0f;
		return MULLS_OK;
	}
}
