// pcl_restated.h — TEST INFRASTRUCTURE.  The PCL / FLANN / Eigen behaviours that the reference's feature-extraction lines call and
// that are not in this image, restated once and shared by the oracle (mulls_oracle.cpp) and by the stand-ins the reference's own
// lines are compiled against (ref_shim/shim.hpp) — so a comparison "oracle == reference lines" pins everything MULLS wrote
// (types, thresholds, orders, side effects) and nothing in this file.  "Parity unpinned" for what is restated here:
//
//   pcl::KdTreeFLANN<PointT>::radiusSearch(index|point, radius, k_indices, k_sqr_distances, max_nn)
//   pcl::search::KdTree<PointT>::radiusSearch(point, radius, k_indices, k_sqr_distances)          (pcl/kdtree/impl/kdtree_flann.hpp)
//       squared L2_Simple<float> distance (((dx*dx)+dy*dy)+dz*dz) < (float)(radius*radius) — FLANN's (KNN)RadiusResultSet takes a
//       candidate while `dist < worst` —, sorted ascending (SearchParams sorted = true), cut to the max_nn nearest when max_nn > 0.
//       Equal distances: by index here; FLANN's order among them is an implementation detail.
//   pcl::PCA<PointT> (pcl/common/impl/pca.hpp, initCompute)
//       mean_ = compute3DCentroid (float accumulators in the points' order, divided by the count); demeaned float coordinates;
//       covariance = 1/(n-1) * D * D^T in float; Eigen::SelfAdjointEigenSolver<Matrix3f>; eigenvalues descending, eigenvectors as
//       columns, col(2) = col(0) x col(1).  Here: the float sums in the points' order, the decomposition by cyclic Jacobi rotations
//       in double on the float matrix, results rounded to float; Eigen leaves an eigenvector's sign open, here its largest component is
//       positive.  Eigen's float solver is accurate to ~1e-6 of the largest eigenvalue: threshold comparisons downstream can differ from
//       upstream for points that close to a threshold.
//   pcl::SACSegmentation<PointT> with SACMODEL_PLANE, SAC_RANSAC, setOptimizeCoefficients(true)   (cprocessing.hpp:67-106: the ground filter's
//       normal method 3, one call per grid cell) — plane_ransac below: segmentation/impl/sac_segmentation.hpp (segment), sample_consensus/
//       impl/ransac.hpp (computeModel), sac_model.h (getSamples, drawIndexSample: a partial Fisher-Yates shuffle carried from draw to draw,
//       rnd() = boost::mt19937 seeded 12345u — SACSegmentation(random = false) — through boost::uniform_int<>(0, INT_MAX), which for Boost >=
//       1.47 is eng() / 2), impl/sac_model_plane.hpp (isSampleGood, computeModelCoefficients, countWithinDistance, selectWithinDistance,
//       optimizeModelCoefficients), common/impl/centroid.hpp (computeMeanAndCovarianceMatrix: one-pass float moments).  Float expressions in
//       Eigen's SSE evaluation order for 4-vectors ((a0 + a2) + (a1 + a3)).  Two things are NOT PCL's: the loop test `iterations < log(1 - 0.99) /
//       log(1 - w^3)` is evaluated as `(1 - w^3)^iterations > 1 - 0.99` by repeated multiplication (no libm: the same bits on the device; the two
//       differ only if the power lands within rounding of 0.01), and pcl::eigen33's closed-form float roots are replaced by the smallest
//       eigenvector of the same float covariance from cyclic Jacobi in double (jacobi3), its largest component positive (eigen33 leaves the
//       sign to a cross product; fast_ground_filter tests abs(normal_z) and the point-to-plane metric is even in the normal).
//   pcl::NormalEstimationOMP<PointT, pcl::Normal> with setRadiusSearch / setKSearch   (pca.hpp:66-119: normal methods 1 / 2) — normal_estimation
//       below: features/impl/normal_3d_omp.hpp, features/normal_3d.h (computePointNormal, flipNormalTowardsViewpoint with the view point
//       (0, 0, 0)): neighbours ascending by (distance, index), the query included; fewer than 3 -> NaN (which pca.hpp's check_normal turns
//       into 0.577); one-pass float moments in that order; the normal = the smallest eigenvector as above, turned towards the origin.
#pragma once
#include <algorithm>
#include <atomic>
#include <climits>
#include <cstdint>
#include <random>
#include <cmath>
#include <cstring>
#include <utility>
#include <vector>

namespace restated
{
template <typename P, typename Q>
inline float l2_simple(const P &a, const Q &b)
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

// Census of the places where the restated searches had to CHOOSE what FLANN leaves to its implementation (tests/test_pcl_operators.py reads it on the demo scans):
//   [0] searches   [1] searches whose max_nn cut falls inside a group of equal distances (which of them stay is FLANN's business)
//   [2] candidates exactly ON the radius (d == r^2: in or out depends on `<` against `<=`)   [3] searches that return two neighbours at the same distance
//       (their order decides the order of the float sums of the PCA that follows)
inline std::atomic<unsigned long long> &search_census(int which)
{
	static std::atomic<unsigned long long> c[4];
	return c[which];
}

// exact radius / radius-k search over a fixed point set; a uniform grid only limits which points are looked at
template <typename P>
class RadiusIndex
{
  public:
	void build(const std::vector<P> &pts, float cell_hint)
	{
		pts_ = &pts;
		n_ = pts.size();
		cell_ = cell_hint > 1e-3f ? cell_hint : 1e-3f;
		start_.clear();
		order_.clear();
		finite_ = true;
		if (!n_)
			return;
		double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
		for (size_t i = 0; i < n_; i++)
		{
			const double v[3] = {pts[i].x, pts[i].y, pts[i].z};
			for (int k = 0; k < 3; k++)
			{
				if (!std::isfinite(v[k]))
					finite_ = false;
				lo[k] = std::min(lo[k], v[k]);
				hi[k] = std::max(hi[k], v[k]);
			}
		}
		if (!finite_)
			return; // brute force below
		for (;;)
		{
			double cells = 1;
			for (int k = 0; k < 3; k++)
			{
				lo_[k] = lo[k];
				dim_[k] = (long)std::floor((hi[k] - lo[k]) / cell_) + 1;
				cells *= (double)dim_[k];
			}
			if (cells <= 8e6)
				break;
			cell_ *= 2;
		}
		const size_t nc = (size_t)dim_[0] * dim_[1] * dim_[2];
		start_.assign(nc + 1, 0);
		std::vector<size_t> key(n_);
		for (size_t i = 0; i < n_; i++)
		{
			key[i] = cell_of(pts[i].x, pts[i].y, pts[i].z);
			start_[key[i] + 1]++;
		}
		for (size_t c = 0; c < nc; c++)
			start_[c + 1] += start_[c];
		order_.resize(n_);
		std::vector<size_t> fill(start_.begin(), start_.end() - 1);
		for (size_t i = 0; i < n_; i++)
			order_[fill[key[i]]++] = (int)i;
	}
	template <typename Q>
	void search(const Q &q, double radius, unsigned max_nn, std::vector<int> &idx, std::vector<float> &d2) const
	{
		const float r2 = static_cast<float>(radius * radius);
		std::vector<std::pair<float, int>> all;
		const std::vector<P> &pts = *pts_;
		if (!finite_ || !std::isfinite((double)q.x) || !std::isfinite((double)q.y) || !std::isfinite((double)q.z) || !(radius < 1e18))
		{
			for (size_t t = 0; t < n_; t++)
			{
				const float d = l2_simple(q, pts[t]);
				if (d == r2)
					search_census(2)++;
				if (d < r2)
					all.push_back(std::make_pair(d, (int)t));
			}
		}
		else if (n_)
		{
			long c0[3], c1[3];
			const double v[3] = {q.x, q.y, q.z};
			const double reach = radius * (1.0 + 1e-6) + 1e-6;
			for (int k = 0; k < 3; k++)
			{
				c0[k] = std::max(0L, (long)std::floor((v[k] - reach - lo_[k]) / cell_));
				c1[k] = std::min(dim_[k] - 1, (long)std::floor((v[k] + reach - lo_[k]) / cell_));
			}
			for (long cx = c0[0]; cx <= c1[0]; cx++)
				for (long cy = c0[1]; cy <= c1[1]; cy++)
					for (long cz = c0[2]; cz <= c1[2]; cz++)
					{
						const size_t c = ((size_t)cx * dim_[1] + cy) * dim_[2] + cz;
						for (size_t s = start_[c]; s < start_[c + 1]; s++)
						{
							const int t = order_[s];
							const float d = l2_simple(q, pts[t]);
							if (d == r2)
								search_census(2)++;
							if (d < r2)
								all.push_back(std::make_pair(d, t));
						}
					}
		}
		std::sort(all.begin(), all.end()); // (distance, index)
		search_census(0)++;
		if (max_nn > 0 && all.size() > (size_t)max_nn)
		{
			if (all[max_nn - 1].first == all[max_nn].first)
				search_census(1)++;
			all.resize(max_nn);
		}
		for (size_t k = 1; k < all.size(); k++)
			if (all[k].first == all[k - 1].first)
			{
				search_census(3)++;
				break;
			}
		idx.resize(all.size());
		d2.resize(all.size());
		for (size_t k = 0; k < all.size(); k++)
		{
			d2[k] = all[k].first;
			idx[k] = all[k].second;
		}
	}

  private:
	size_t cell_of(double x, double y, double z) const
	{
		const long cx = (long)std::floor((x - lo_[0]) / cell_), cy = (long)std::floor((y - lo_[1]) / cell_), cz = (long)std::floor((z - lo_[2]) / cell_);
		return ((size_t)cx * dim_[1] + cy) * dim_[2] + cz;
	}
	const std::vector<P> *pts_ = nullptr;
	size_t n_ = 0;
	float cell_ = 1;
	bool finite_ = true;
	double lo_[3] = {0, 0, 0};
	long dim_[3] = {1, 1, 1};
	std::vector<size_t> start_;
	std::vector<int> order_;
};

// eigen-decomposition of a symmetric 3x3 (a6 = xx xy xz yy yz zz): cyclic Jacobi on the upper triangle; eigenvalues descending (the
// lower index first among equals), unit eigenvectors as the columns of v, each with its largest component positive
inline void jacobi3(const double a6[6], double lam[3], double v[3][3])
{
	double A[3][3] = {{a6[0], a6[1], a6[2]}, {a6[1], a6[3], a6[4]}, {a6[2], a6[4], a6[5]}};
	for (int r = 0; r < 3; r++)
		for (int c = 0; c < 3; c++)
			v[r][c] = r == c ? 1.0 : 0.0;
	for (int sweep = 0; sweep < 60; sweep++)
	{
		const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
		if (off < 1e-300)
			break;
		for (int p = 0; p < 2; p++)
			for (int q = p + 1; q < 3; q++)
			{
				if (A[p][q] == 0.0)
					continue;
				const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
				const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
				const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
				for (int k = 0; k < 3; k++) // A <- A J
				{
					const double akp = A[k][p], akq = A[k][q];
					A[k][p] = cs * akp - sn * akq;
					A[k][q] = sn * akp + cs * akq;
				}
				for (int k = 0; k < 3; k++) // A <- J^T A
				{
					const double apk = A[p][k], aqk = A[q][k];
					A[p][k] = cs * apk - sn * aqk;
					A[q][k] = sn * apk + cs * aqk;
				}
				for (int r = 0; r < 3; r++) // the upper triangle is the matrix
					for (int c2 = r + 1; c2 < 3; c2++)
						A[c2][r] = A[r][c2];
				for (int k = 0; k < 3; k++)
				{
					const double vkp = v[k][p], vkq = v[k][q];
					v[k][p] = cs * vkp - sn * vkq;
					v[k][q] = sn * vkp + cs * vkq;
				}
			}
	}
	int ord[3] = {0, 1, 2};
	for (int i = 0; i < 3; i++)
		for (int j = i + 1; j < 3; j++)
			if (A[ord[j]][ord[j]] > A[ord[i]][ord[i]])
				std::swap(ord[i], ord[j]);
	double vv[3][3];
	for (int i = 0; i < 3; i++)
	{
		lam[i] = A[ord[i]][ord[i]];
		double nrm = 0;
		for (int k = 0; k < 3; k++)
			nrm += v[k][ord[i]] * v[k][ord[i]];
		nrm = std::sqrt(nrm);
		int big = 0;
		for (int k = 1; k < 3; k++)
			if (std::fabs(v[k][ord[i]]) > std::fabs(v[big][ord[i]]))
				big = k;
		const double sgn = v[big][ord[i]] < 0 ? -1.0 : 1.0;
		for (int k = 0; k < 3; k++)
			vv[k][i] = sgn * v[k][ord[i]] / nrm;
	}
	std::memcpy(v, vv, sizeof(vv));
}

// pcl::PCA on pts[idx[0..n)]: eigenvalues (descending) and eigenvectors (evec[row][col]); needs n >= 2
template <typename P>
inline void pca(const std::vector<P> &pts, const std::vector<int> &idx, float eval[3], float evec[3][3])
{
	const int n = (int)idx.size();
	float mx = 0, my = 0, mz = 0;
	for (int i = 0; i < n; i++)
	{
		mx += pts[idx[i]].x;
		my += pts[idx[i]].y;
		mz += pts[idx[i]].z;
	}
	mx /= (float)n;
	my /= (float)n;
	mz /= (float)n;
	float s[6] = {0, 0, 0, 0, 0, 0};
	for (int i = 0; i < n; i++)
	{
		const float dx = pts[idx[i]].x - mx, dy = pts[idx[i]].y - my, dz = pts[idx[i]].z - mz;
		s[0] += dx * dx;
		s[1] += dx * dy;
		s[2] += dx * dz;
		s[3] += dy * dy;
		s[4] += dy * dz;
		s[5] += dz * dz;
	}
	const float alpha = 1.f / ((float)n - 1.f);
	double a6[6], lam[3], v[3][3];
	for (int k = 0; k < 6; k++)
		a6[k] = (double)(alpha * s[k]);
	jacobi3(a6, lam, v);
	// rounded to float through volatiles: g++ 11's vectoriser (-O3) otherwise forwards the doubles to a caller that widens the results
	// again, i.e. drops the rounding (seen in the reference-lines build: pca.hpp:424-426 assigns the eigenvalues to doubles)
	volatile float rounded[9];
	for (int c = 0; c < 3; c++)
		rounded[c] = (float)lam[c];
	for (int r = 0; r < 3; r++)
	{
		rounded[3 + r] = (float)v[r][0];
		rounded[6 + r] = (float)v[r][1];
	}
	for (int c = 0; c < 3; c++)
		eval[c] = rounded[c];
	for (int r = 0; r < 3; r++)
	{
		evec[r][0] = rounded[3 + r];
		evec[r][1] = rounded[6 + r];
	}
	evec[0][2] = evec[1][0] * evec[2][1] - evec[2][0] * evec[1][1]; // col(2) = col(0).cross(col(1))
	evec[1][2] = evec[2][0] * evec[0][1] - evec[0][0] * evec[2][1];
	evec[2][2] = evec[0][0] * evec[1][1] - evec[1][0] * evec[0][1];
}

// Eigen::Vector3f::normalize()
inline void normalize3(float v[3])
{
	const float z = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
	if (z > 0)
	{
		const float n = std::sqrt(z);
		v[0] /= n;
		v[1] /= n;
		v[2] /= n;
	}
}

// ---------------------------------------------------------------------------------------------------------------------------------
struct P4 // Eigen::Array4f map of a point: x, y, z, data[3]
{
	float x, y, z, w;
};
// pcl::computeMeanAndCovarianceMatrix (dense cloud, float): raw moments summed in the indices' order, divided by the count, then
// covariance = E[xx^T] - c c^T.  cov6 = xx xy xz yy yz zz.  Returns the count.
inline int mean_and_covariance(const std::vector<P4> &pts, const std::vector<int> &idx, float cov6[6], float centroid[3])
{
	float accu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
	for (size_t k = 0; k < idx.size(); k++)
	{
		const P4 &p = pts[idx[k]];
		accu[0] += p.x * p.x;
		accu[1] += p.x * p.y;
		accu[2] += p.x * p.z;
		accu[3] += p.y * p.y;
		accu[4] += p.y * p.z;
		accu[5] += p.z * p.z;
		accu[6] += p.x;
		accu[7] += p.y;
		accu[8] += p.z;
	}
	const float n = static_cast<float>(idx.size());
	for (int k = 0; k < 9; k++)
		accu[k] /= n;
	centroid[0] = accu[6], centroid[1] = accu[7], centroid[2] = accu[8];
	cov6[0] = accu[0] - accu[6] * accu[6];
	cov6[1] = accu[1] - accu[6] * accu[7];
	cov6[2] = accu[2] - accu[6] * accu[8];
	cov6[3] = accu[3] - accu[7] * accu[7];
	cov6[4] = accu[4] - accu[7] * accu[8];
	cov6[5] = accu[5] - accu[8] * accu[8];
	return (int)idx.size();
}
// the unit eigenvector of the smallest eigenvalue of a float covariance, rounded to float (largest component positive)
inline void smallest_eigenvector(const float cov6[6], float n[3])
{
	double a6[6], lam[3], v[3][3];
	for (int k = 0; k < 6; k++)
		a6[k] = (double)cov6[k];
	jacobi3(a6, lam, v);
	volatile float r[3] = {(float)v[0][2], (float)v[1][2], (float)v[2][2]}; // (see pca(): the rounding must not be forwarded away)
	n[0] = r[0], n[1] = r[1], n[2] = r[2];
}
// dot of a plane (4 coefficients) with (x, y, z, w) as Eigen's 4-float packet reduction adds it up
inline float plane_dot(const float c[4], float x, float y, float z, float w) { return (c[0] * x + c[2] * z) + (c[1] * y + c[3] * w); }

// counters of plane_ransac's stopping test (oracle census, tests/test_pcl_operators.py): [0] evaluations with `iterations` within 1e-9 of k, [1] evaluations where
// PCL's `iterations_ < k` and the libm-free product form disagree, [2] evaluations in all
inline unsigned long long &ransac_census(int which)
{
	static unsigned long long c[3] = {0, 0, 0};
	return c[which];
}

// pcl::SACSegmentation<PointT>::segment as CProceesing::plane_seg_ransac sets it up.  Out: the inliers (ascending indices) and the four
// plane coefficients; false when no model could be found (upstream then leaves both outputs empty).
inline bool plane_ransac(const std::vector<P4> &pts, double threshold, int max_iterations, std::vector<int> &inliers, float coeff[4])
{
	const int n = (int)pts.size();
	inliers.clear();
	if (n < 3)
		return false; // getSamples: "Can not select 3 unique points out of n"
	std::mt19937 eng(12345u);
	std::vector<int> shuffled(n);
	for (int i = 0; i < n; i++)
		shuffled[i] = i;
	int iterations = 0, n_best = -INT_MAX;
	float best[4] = {0, 0, 0, 0};
	bool have = false;
	double p_no_outliers = 0.0;
	const double one_over_indices = 1.0 / static_cast<double>(n), log_arg = 1.0 - 0.99; // probability_ = 0.99
	double k = 1.0; // ransac.hpp: "double k = 1.0" — the loop is `while (iterations_ < k && ...)`
	for (;;)
	{
		if (have)
		{
			// PCL's own test, `iterations_ < k` with k = log(1 - probability_) / log(p_no_outliers) (std::log / std::pow as ransac.hpp calls them), decides.
			// The device evaluates the same test without libm, as p_no_outliers^iterations > 1 - probability_ (k_gf_ransac: a product of doubles is the same bits
			// on every platform); the two can only disagree when `iterations` sits within rounding of k — counted, and asserted to be 0 on the demo data
			// (tests/test_pcl_operators.py)
			double pw = 1.0;
			for (int i = 0; i < iterations; i++)
				pw *= p_no_outliers;
			const bool go_product = pw > log_arg, go = (double)iterations < k;
			ransac_census(2)++;
			if (std::fabs((double)iterations - k) <= 1e-9 * std::max(k, 1.0))
				ransac_census(0)++;
			if (go != go_product)
				ransac_census(1)++;
			if (!go)
				break;
		}
		else if (iterations > 0)
			break; // (unreachable: the first good sample always installs a model)
		// getSamples: up to max_sample_checks_ = 1000 draws until isSampleGood
		int sel[3] = {0, 0, 0};
		bool good = false;
		for (int check = 0; check < 1000 && !good; check++)
		{
			for (int i = 0; i < 3; i++)
				std::swap(shuffled[i], shuffled[i + (int)((eng() >> 1) % (uint32_t)(n - i))]);
			sel[0] = shuffled[0], sel[1] = shuffled[1], sel[2] = shuffled[2];
			const P4 &p0 = pts[sel[0]], &p1 = pts[sel[1]], &p2 = pts[sel[2]];
			const float d0 = (p1.x - p0.x) / (p2.x - p0.x), d1 = (p1.y - p0.y) / (p2.y - p0.y), d2 = (p1.z - p0.z) / (p2.z - p0.z);
			good = (d0 != d1) || (d2 != d1);
		}
		if (!good)
			break; // "No samples could be selected!"
		// computeModelCoefficients (the collinearity test it repeats cannot fail after isSampleGood)
		const P4 &p0 = pts[sel[0]], &p1 = pts[sel[1]], &p2 = pts[sel[2]];
		const float ax = p1.x - p0.x, ay = p1.y - p0.y, az = p1.z - p0.z, bx = p2.x - p0.x, by = p2.y - p0.y, bz = p2.z - p0.z;
		float c[4];
		c[0] = ay * bz - az * by;
		c[1] = az * bx - ax * bz;
		c[2] = ax * by - ay * bx;
		c[3] = 0;
		const float z = (c[0] * c[0] + c[2] * c[2]) + (c[1] * c[1] + c[3] * c[3]);
		if (z > 0.0f)
		{
			const float nrm = std::sqrt(z);
			c[0] /= nrm, c[1] /= nrm, c[2] /= nrm, c[3] /= nrm;
		}
		c[3] = -1 * plane_dot(c, p0.x, p0.y, p0.z, p0.w);
		// countWithinDistance
		int count = 0;
		for (int i = 0; i < n; i++)
			if ((double)std::fabs(plane_dot(c, pts[i].x, pts[i].y, pts[i].z, 1.0f)) < threshold)
				count++;
		if (count > n_best)
		{
			n_best = count;
			for (int k = 0; k < 4; k++)
				best[k] = c[k];
			have = true;
			const double w = static_cast<double>(n_best) * one_over_indices;
			p_no_outliers = 1.0 - w * w * w; // the device's form of pow(w, 3.0): feeds the product test above only
			p_no_outliers = std::max(std::numeric_limits<double>::epsilon(), p_no_outliers);
			p_no_outliers = std::min(1.0 - std::numeric_limits<double>::epsilon(), p_no_outliers);
			{
				// ransac.hpp: p_no_outliers = 1 - pow(w, selection.size()), clamped to [eps, 1 - eps]; k = log_probability / log(p_no_outliers)
				double pno = 1.0 - std::pow(w, 3.0);
				pno = std::max(std::numeric_limits<double>::epsilon(), pno);
				pno = std::min(1.0 - std::numeric_limits<double>::epsilon(), pno);
				k = std::log(1.0 - 0.99) / std::log(pno);
			}
		}
		++iterations;
		if (iterations > max_iterations)
			break;
	}
	if (!have)
		return false;
	auto select = [&](const float c[4]) {
		inliers.clear();
		for (int i = 0; i < n; i++)
			if ((double)std::fabs(plane_dot(c, pts[i].x, pts[i].y, pts[i].z, 1.0f)) < threshold)
				inliers.push_back(i);
	};
	select(best);
	// optimizeModelCoefficients: least squares through the inliers (needs more than 3), then the inliers of the refined plane
	for (int k = 0; k < 4; k++)
		coeff[k] = best[k];
	if (inliers.size() >= 4)
	{
		float cov6[6], cen[3], nv[3];
		mean_and_covariance(pts, inliers, cov6, cen);
		smallest_eigenvector(cov6, nv);
		coeff[0] = nv[0], coeff[1] = nv[1], coeff[2] = nv[2], coeff[3] = 0;
		coeff[3] = -1 * plane_dot(coeff, cen[0], cen[1], cen[2], 1.0f);
	}
	select(coeff);
	return true;
}

// pcl::NormalEstimationOMP on a cloud against itself: out[i] = (nx, ny, nz) of point i, NaN where fewer than 3 neighbours were found.
// radius > 0: every point within the radius (setRadiusSearch); else the k nearest (setKSearch).
template <typename P>
inline void normal_estimation(const std::vector<P> &cloud, double radius, int k, std::vector<float> &out)
{
	const size_t n = cloud.size();
	out.assign(3 * n, std::numeric_limits<float>::quiet_NaN());
	if (!n)
		return;
	std::vector<P4> pts(n);
	for (size_t i = 0; i < n; i++)
		pts[i] = P4{cloud[i].x, cloud[i].y, cloud[i].z, 1.0f};
	RadiusIndex<P4> index;
	index.build(pts, radius > 0 ? (float)radius : 1.0f);
	std::vector<int> idx;
	std::vector<float> d2;
	for (size_t i = 0; i < n; i++)
	{
		if (radius > 0)
			index.search(pts[i], radius, 0, idx, d2);
		else
		{
			// the k nearest: every point within R for a growing R — once k are inside, the k nearest are among them
			double R = 1.0;
			for (;;)
			{
				index.search(pts[i], R, 0, idx, d2);
				if ((int)idx.size() >= k || idx.size() == n || R > 1e7)
					break;
				R *= 2.0;
			}
			if ((int)idx.size() > k)
				idx.resize(k);
		}
		if (idx.size() < 3)
			continue;
		float cov6[6], cen[3], nv[3];
		mean_and_covariance(pts, idx, cov6, cen);
		smallest_eigenvector(cov6, nv);
		// flipNormalTowardsViewpoint(point, 0, 0, 0, nx, ny, nz)
		const float vx = 0.0f - pts[i].x, vy = 0.0f - pts[i].y, vz = 0.0f - pts[i].z;
		const float cos_theta = (vx * nv[0] + vy * nv[1] + vz * nv[2]);
		if (cos_theta < 0)
			nv[0] *= -1, nv[1] *= -1, nv[2] *= -1;
		out[3 * i] = nv[0], out[3 * i + 1] = nv[1], out[3 * i + 2] = nv[2];
	}
}
} // namespace restated
