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
#pragma once
#include <algorithm>
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
							if (d < r2)
								all.push_back(std::make_pair(d, t));
						}
					}
		}
		std::sort(all.begin(), all.end()); // (distance, index)
		if (max_nn > 0 && all.size() > (size_t)max_nn)
			all.resize(max_nn);
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
} // namespace restated
