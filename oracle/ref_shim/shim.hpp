// shim.hpp — minimal stand-ins for the third-party APIs the MULLS-ICP function bodies touch (Eigen, PCL, boost, glog).
//
// TEST INFRASTRUCTURE.  Purpose: let the reference's OWN source lines for the hot path (include/common/
// cregistration.hpp, cfilter.hpp, utility.hpp under /root/reference) compile in this image, which has none of those
// libraries, so that the CPU oracle restatement can be pinned against the reference's real control flow and
// arithmetic (oracle/build_ref.sh -> oracle/_ref/libmulls_ref.so).  What this pins: every line MULLS itself wrote for
// the path.  What it does NOT pin: the behaviour of PCL / FLANN / Eigen themselves — the classes below restate the
// documented behaviour of exactly the calls the path makes (PCL 1.8-1.10, Eigen 3.3), they are
// not those libraries.
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <set>
#include <vector>

#include "pcl_restated.h" // oracle/: the PCL / FLANN / Eigen behaviours of the feature-extraction lines, restated once

// ------------------------------------------------------------------------------------------------------------------
// glog
struct ShimNullStream
{
	template <typename T>
	ShimNullStream &operator<<(const T &) { return *this; }
	ShimNullStream &operator<<(std::ostream &(*)(std::ostream &)) { return *this; }
};
#define LOG(severity) ShimNullStream()

// ------------------------------------------------------------------------------------------------------------------
// boost
namespace boost
{
template <typename T>
using shared_ptr = std::shared_ptr<T>;
template <typename T, typename... A>
shared_ptr<T> make_shared(A &&... a) { return std::make_shared<T>(std::forward<A>(a)...); }
} // namespace boost

// ------------------------------------------------------------------------------------------------------------------
// Eigen (fixed-size, column-major, eager evaluation)
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
namespace Eigen
{
template <typename T>
using aligned_allocator = std::allocator<T>;

template <typename T, int R, int C>
struct Matrix;
template <typename T, int R, int C>
struct CommaInit
{
	Matrix<T, R, C> *m;
	int k;
	template <typename S>
	CommaInit &operator,(S v)
	{
		m->set_rowmajor(k++, (T)v);
		return *this;
	}
};
template <typename T, int PR, int PC, int R, int C>
struct BlockRef;

template <typename T, int R, int C>
struct Matrix
{
	T v[R * C];
	Matrix() { std::memset(v, 0, sizeof(v)); }
	Matrix(T a, T b, T c)
	{
		static_assert(R * C == 3, "3-vector constructor");
		v[0] = a, v[1] = b, v[2] = c;
	}
	static Matrix Identity()
	{
		Matrix m;
		for (int i = 0; i < (R < C ? R : C); i++)
			m(i, i) = 1;
		return m;
	}
	static Matrix Identity(int, int) { return Identity(); }
	static Matrix Zero() { return Matrix(); }
	void setIdentity() { *this = Identity(); }
	void setZero() { std::memset(v, 0, sizeof(v)); }
	T &operator()(int r, int c) { return v[r + R * c]; }
	const T &operator()(int r, int c) const { return v[r + R * c]; }
	T &operator()(int i) { return v[i]; }
	const T &operator()(int i) const { return v[i]; }
	T &coeffRef(int i) { return v[i]; }
	T *data() { return v; }
	const T *data() const { return v; }
	void set_rowmajor(int k, T x) { (*this)(k / C, k % C) = x; }
	template <typename S>
	CommaInit<T, R, C> operator<<(S first)
	{
		set_rowmajor(0, (T)first);
		return CommaInit<T, R, C>{this, 1};
	}
	Matrix<T, C, R> transpose() const
	{
		Matrix<T, C, R> t;
		for (int r = 0; r < R; r++)
			for (int c = 0; c < C; c++)
				t(c, r) = (*this)(r, c);
		return t;
	}
	T norm() const
	{
		T s = 0;
		for (int i = 0; i < R * C; i++)
			s += v[i] * v[i];
		return std::sqrt(s);
	}
	T dot(const Matrix &o) const
	{
		T s = 0;
		for (int i = 0; i < R * C; i++)
			s += v[i] * o.v[i];
		return s;
	}
	operator T() const
	{
		static_assert(R == 1 && C == 1, "only 1x1 converts to scalar");
		return v[0];
	}
	const T &coeff(int i) const { return v[i]; }
	const T &x() const { return v[0]; }
	const T &y() const { return v[1]; }
	const T &z() const { return v[2]; }
	void normalize() // Eigen 3.3: if (squaredNorm() > 0) *this /= sqrt(squaredNorm())
	{
		T z2 = 0;
		for (int i = 0; i < R * C; i++)
			z2 += v[i] * v[i];
		if (z2 > T(0))
		{
			const T n = std::sqrt(z2);
			for (int i = 0; i < R * C; i++)
				v[i] /= n;
		}
	}
	BlockRef<T, R, C, R, 1> col(int c) { return BlockRef<T, R, C, R, 1>(this, 0, c); }
	template <int BR, int BC>
	BlockRef<T, R, C, BR, BC> block(int r0, int c0)
	{
		return BlockRef<T, R, C, BR, BC>(this, r0, c0);
	}
	template <int BR, int BC>
	Matrix<T, BR, BC> block(int r0, int c0) const
	{
		Matrix<T, BR, BC> b;
		for (int r = 0; r < BR; r++)
			for (int c = 0; c < BC; c++)
				b(r, c) = (*this)(r0 + r, c0 + c);
		return b;
	}
	// Eigen 3.3 fixed 3x3 inverse (compute_inverse_size3_helper): cofactors over the determinant
	Matrix inverse3() const
	{
		Matrix out;
		auto M = [&](int r, int c) { return v[(r % R) + R * (c % C)]; };
		auto cof = [&](int i, int j) {
			const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
			return M(i1, j1) * M(i2, j2) - M(i1, j2) * M(i2, j1);
		};
		const T c0 = cof(0, 0), c1 = cof(1, 0), c2 = cof(2, 0);
		const T det = (c0 * M(0, 0) + c1 * M(1, 0)) + c2 * M(2, 0);
		const T invdet = T(1) / det;
		T *o = out.v;
		o[(0) + R * (0 % C)] = c0 * invdet;
		o[(0) + R * (1 % C)] = c1 * invdet;
		o[(0) + R * (2 % C)] = c2 * invdet;
		o[(1 % R) + R * (0 % C)] = cof(0, 1) * invdet;
		o[(1 % R) + R * (1 % C)] = cof(1, 1) * invdet;
		o[(1 % R) + R * (2 % C)] = cof(2, 1) * invdet;
		o[(2 % R) + R * (0 % C)] = cof(0, 2) * invdet;
		o[(2 % R) + R * (1 % C)] = cof(1, 2) * invdet;
		o[(2 % R) + R * (2 % C)] = cof(2, 2) * invdet;
		return out;
	}
	// partial-pivot LU inverse (what Eigen 3.3 does for fixed sizes > 4; the 4x4 cofactor path agrees to rounding)
	Matrix inverse() const
	{
		static_assert(R == C, "square");
		if (R == 3)
			return inverse3();
		const int n = R;
		T a[R * C];
		int perm[R];
		std::memcpy(a, v, sizeof(a));
		for (int i = 0; i < n; i++)
			perm[i] = i;
		for (int k = 0; k < n; k++)
		{
			int p = k;
			T big = std::fabs(a[k + n * k]);
			for (int r = k + 1; r < n; r++)
				if (std::fabs(a[r + n * k]) > big)
				{
					big = std::fabs(a[r + n * k]);
					p = r;
				}
			if (p != k)
			{
				for (int c = 0; c < n; c++)
					std::swap(a[k + n * c], a[p + n * c]);
				std::swap(perm[k], perm[p]);
			}
			for (int r = k + 1; r < n; r++)
				a[r + n * k] /= a[k + n * k];
			for (int c = k + 1; c < n; c++)
				for (int r = k + 1; r < n; r++)
					a[r + n * c] -= a[r + n * k] * a[k + n * c];
		}
		Matrix out;
		for (int c = 0; c < n; c++)
		{
			T y[R];
			for (int r = 0; r < n; r++)
				y[r] = perm[r] == c ? 1 : 0;
			for (int r = 0; r < n; r++)
				for (int k = 0; k < r; k++)
					y[r] -= a[r + n * k] * y[k];
			for (int r = n - 1; r >= 0; r--)
			{
				for (int k = r + 1; k < n; k++)
					y[r] -= a[r + n * k] * y[k];
				y[r] /= a[r + n * r];
			}
			for (int r = 0; r < n; r++)
				out(r, c) = y[r];
		}
		return out;
	}
};

template <typename T, int PR, int PC, int R, int C>
struct BlockRef : Matrix<T, R, C>
{
	Matrix<T, PR, PC> *parent;
	int r0, c0;
	BlockRef(Matrix<T, PR, PC> *p, int r, int c) : parent(p), r0(r), c0(c)
	{
		for (int i = 0; i < R; i++)
			for (int j = 0; j < C; j++)
				(*this)(i, j) = (*p)(r0 + i, c0 + j);
	}
	BlockRef &operator=(const Matrix<T, R, C> &m)
	{
		for (int i = 0; i < R; i++)
			for (int j = 0; j < C; j++)
				(*parent)(r0 + i, c0 + j) = (*this)(i, j) = m(i, j);
		return *this;
	}
	BlockRef &operator=(const BlockRef &m) { return *this = static_cast<const Matrix<T, R, C> &>(m); }
	struct Comma
	{
		BlockRef *b;
		int k;
		template <typename S>
		Comma &operator,(S v)
		{
			(*b->parent)(b->r0 + k / C, b->c0 + k % C) = (T)v;
			k++;
			return *this;
		}
	};
	template <typename S>
	Comma operator<<(S first)
	{
		(*parent)(r0, c0) = (T)first;
		return Comma{this, 1};
	}
};

template <typename T, int R, int K, int C>
Matrix<T, R, C> operator*(const Matrix<T, R, K> &a, const Matrix<T, K, C> &b)
{
	Matrix<T, R, C> o;
	for (int c = 0; c < C; c++)
		for (int r = 0; r < R; r++)
		{
			T s = 0;
			for (int k = 0; k < K; k++)
				s += a(r, k) * b(k, c);
			o(r, c) = s;
		}
	return o;
}
template <typename T, int R, int C>
Matrix<T, R, C> operator+(const Matrix<T, R, C> &a, const Matrix<T, R, C> &b)
{
	Matrix<T, R, C> o;
	for (int i = 0; i < R * C; i++)
		o.v[i] = a.v[i] + b.v[i];
	return o;
}
template <typename T, int R, int C>
Matrix<T, R, C> operator-(const Matrix<T, R, C> &a, const Matrix<T, R, C> &b)
{
	Matrix<T, R, C> o;
	for (int i = 0; i < R * C; i++)
		o.v[i] = a.v[i] - b.v[i];
	return o;
}
template <typename T, int R, int C, typename S, typename = typename std::enable_if<std::is_arithmetic<S>::value>::type>
Matrix<T, R, C> operator*(S s, const Matrix<T, R, C> &a)
{
	Matrix<T, R, C> o;
	for (int i = 0; i < R * C; i++)
		o.v[i] = (T)s * a.v[i];
	return o;
}
template <typename T, int R, int C>
bool operator==(const Matrix<T, R, C> &a, const Matrix<T, R, C> &b)
{
	return std::memcmp(a.v, b.v, sizeof(a.v)) == 0;
}

typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<float, 4, 4> Matrix4f;
typedef Matrix<float, 3, 3> Matrix3f;
typedef Matrix<float, 4, 1> Vector4f;
typedef Matrix<float, 3, 1> Vector3f;

// unit quaternion, as far as the path uses it (rotation matrix -> quaternion, slerp from identity, rotate a vector)
struct Quaterniond
{
	double qw, qx, qy, qz;
	Quaterniond() : qw(1), qx(0), qy(0), qz(0) {}
	Quaterniond(double w, double x, double y, double z) : qw(w), qx(x), qy(y), qz(z) {}
	explicit Quaterniond(const Matrix3d &m)
	{
		double q[3];
		double t = m(0, 0) + m(1, 1) + m(2, 2);
		if (t > 0.0)
		{
			t = std::sqrt(t + 1.0);
			qw = 0.5 * t;
			t = 0.5 / t;
			qx = (m(2, 1) - m(1, 2)) * t;
			qy = (m(0, 2) - m(2, 0)) * t;
			qz = (m(1, 0) - m(0, 1)) * t;
		}
		else
		{
			int i = 0;
			if (m(1, 1) > m(0, 0))
				i = 1;
			if (m(2, 2) > m(i, i))
				i = 2;
			int j = (i + 1) % 3, k = (j + 1) % 3;
			t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
			q[i] = 0.5 * t;
			t = 0.5 / t;
			qw = (m(k, j) - m(j, k)) * t;
			q[j] = (m(j, i) + m(i, j)) * t;
			q[k] = (m(k, i) + m(i, k)) * t;
			qx = q[0], qy = q[1], qz = q[2];
		}
	}
	static Quaterniond Identity() { return Quaterniond(); }
	Quaterniond slerp(double t, const Quaterniond &o) const
	{
		const double one = 1.0 - DBL_EPSILON;
		double d = qw * o.qw + qx * o.qx + qy * o.qy + qz * o.qz;
		double absD = std::fabs(d), s0, s1;
		if (absD >= one)
		{
			s0 = 1.0 - t;
			s1 = t;
		}
		else
		{
			double theta = std::acos(absD), sinTheta = std::sin(theta);
			s0 = std::sin((1.0 - t) * theta) / sinTheta;
			s1 = std::sin((t * theta)) / sinTheta;
		}
		if (d < 0)
			s1 = -s1;
		return Quaterniond(s0 * qw + s1 * o.qw, s0 * qx + s1 * o.qx, s0 * qy + s1 * o.qy, s0 * qz + s1 * o.qz);
	}
	Vector3d operator*(const Vector3d &v) const
	{
		double uvx = 2.0 * (qy * v(2) - qz * v(1)), uvy = 2.0 * (qz * v(0) - qx * v(2)), uvz = 2.0 * (qx * v(1) - qy * v(0));
		return Vector3d(v(0) + qw * uvx + (qy * uvz - qz * uvy), v(1) + qw * uvy + (qz * uvx - qx * uvz), v(2) + qw * uvz + (qx * uvy - qy * uvx));
	}
};

struct AngleAxisd
{
	double ang;
	explicit AngleAxisd(const Matrix3d &m)
	{
		Quaterniond q(m);
		double n = std::sqrt(q.qx * q.qx + q.qy * q.qy + q.qz * q.qz);
		ang = n != 0.0 ? 2.0 * std::atan2(n, std::fabs(q.qw)) : 0.0;
	}
	double angle() const { return ang; }
};
} // namespace Eigen

// ------------------------------------------------------------------------------------------------------------------
// PCL
namespace pcl
{
struct PointXYZINormal
{
	union {
		float data[4];
		struct
		{
			float x, y, z;
		};
	};
	union {
		float normal[4];
		struct
		{
			float normal_x, normal_y, normal_z;
		};
	};
	float intensity, curvature, pad_[2];
};
static_assert(sizeof(PointXYZINormal) == 48, "pcl::PointXYZINormal is 48 bytes");

template <typename PointT>
struct PointCloud
{
	typedef boost::shared_ptr<PointCloud<PointT>> Ptr;
	std::vector<PointT> points;
	size_t size() const { return points.size(); }
	void push_back(const PointT &p) { points.push_back(p); }
	void swap(PointCloud<PointT> &o) { points.swap(o.points); }
};

struct Correspondence
{
	int index_query, index_match;
	union {
		float distance;
		float weight;
	};
};
typedef std::vector<Correspondence> Correspondences;

// FLANN L2_Simple<float> over x,y,z
template <typename PointT>
inline float shim_l2(const PointT &a, const PointT &b)
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

namespace search
{
// exact nearest neighbours (brute force; ties -> lowest index), standing in for KdTree -> KdTreeFLANN
template <typename PointT>
struct KdTree
{
	typedef boost::shared_ptr<KdTree<PointT>> Ptr;
	typename PointCloud<PointT>::Ptr cloud;
	void setInputCloud(const typename PointCloud<PointT>::Ptr &c)
	{
		cloud = c;
		index_built = false;
	}
	int nearestKSearch(const PointT &q, int k, std::vector<int> &idx, std::vector<float> &d2) const
	{
		// FLANN's KNN result sets admit a candidate only while `dist < worst_dist`: a NaN distance (a query with a NaN coordinate — the
		// clouds after a singular solve) never is, so such a query finds nothing.  (pcl::KdTreeFLANN::nearestKSearch then still returns k
		// with its output vectors as the previous query left them — undefined contents upstream; defined here, in the oracle and on the
		// device as "no neighbour", SURVEY B-11.)
		std::vector<std::pair<float, int>> all;
		if (cloud)
			for (size_t i = 0; i < cloud->points.size(); i++)
			{
				const float d = shim_l2(q, cloud->points[i]);
				if (d == d)
					all.push_back(std::make_pair(d, (int)i));
			}
		int kk = std::min<int>(k, (int)all.size());
		std::partial_sort(all.begin(), all.begin() + kk, all.end());
		idx.resize(kk);
		d2.resize(kk);
		for (int i = 0; i < kk; i++)
		{
			idx[i] = all[i].second;
			d2[i] = all[i].first;
		}
		return kk;
	}
	// FLANN radius search (oracle/pcl_restated.h): squared distance < (float)(radius * radius), ascending, cut to max_nn if > 0
	int radiusSearch(const PointT &q, double radius, std::vector<int> &idx, std::vector<float> &d2, unsigned max_nn = 0) const
	{
		if (!index_built)
		{
			index.build(cloud->points, (float)radius);
			index_built = true;
		}
		index.search(q, radius, max_nn, idx, d2);
		return (int)idx.size();
	}
	int radiusSearch(int i, double radius, std::vector<int> &idx, std::vector<float> &d2, unsigned max_nn = 0) const
	{
		return radiusSearch(cloud->points[i], radius, idx, d2, max_nn);
	}
	mutable restated::RadiusIndex<PointT> index;
	mutable bool index_built = false;
};
} // namespace search

// pcl::getMinMax3D (pcl/common/impl/common.hpp): component-wise bounds of the finite points
template <typename PointT>
inline void getMinMax3D(const PointCloud<PointT> &cloud, Eigen::Matrix<float, 4, 1> &min_pt, Eigen::Matrix<float, 4, 1> &max_pt)
{
	for (int k = 0; k < 4; k++)
	{
		min_pt(k) = FLT_MAX;
		max_pt(k) = -FLT_MAX;
	}
	for (size_t i = 0; i < cloud.points.size(); i++)
	{
		const PointT &p = cloud.points[i];
		if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z))
			continue;
		const float v[3] = {p.x, p.y, p.z};
		for (int k = 0; k < 3; k++)
		{
			min_pt(k) = std::min(min_pt(k), v[k]);
			max_pt(k) = std::max(max_pt(k), v[k]);
		}
	}
}

// pcl::KdTreeFLANN: the same index under its other name (the searches copy the positions at setInputCloud time: later writes to the
// cloud's normals, as get_pc_pca_feature does, are not seen — positions never change)
template <typename PointT>
struct KdTreeFLANN
{
	typedef boost::shared_ptr<KdTreeFLANN<PointT>> Ptr;
	std::vector<PointT> pts;
	mutable restated::RadiusIndex<PointT> index;
	mutable bool index_built = false;
	void setInputCloud(const typename PointCloud<PointT>::Ptr &c)
	{
		pts = c->points;
		index_built = false;
	}
	int radiusSearch(const PointT &q, double radius, std::vector<int> &idx, std::vector<float> &d2, unsigned max_nn = 0) const
	{
		if (!index_built)
		{
			index.build(pts, (float)radius);
			index_built = true;
		}
		index.search(q, radius, max_nn, idx, d2);
		return (int)idx.size();
	}
	int radiusSearch(int i, double radius, std::vector<int> &idx, std::vector<float> &d2, unsigned max_nn = 0) const
	{
		return radiusSearch(pts[i], radius, idx, d2, max_nn);
	}
};

// pcl::PCA (oracle/pcl_restated.h)
template <typename PointT>
struct PCA
{
	Eigen::Matrix<float, 3, 3> vectors;
	Eigen::Matrix<float, 3, 1> values;
	void setInputCloud(const typename PointCloud<PointT>::Ptr &c)
	{
		std::vector<int> all(c->points.size());
		for (size_t i = 0; i < all.size(); i++)
			all[i] = (int)i;
		float eval[3], evec[3][3];
		restated::pca(c->points, all, eval, evec);
		for (int r = 0; r < 3; r++)
		{
			values(r) = eval[r];
			for (int k = 0; k < 3; k++)
				vectors(r, k) = evec[r][k];
		}
	}
	Eigen::Matrix<float, 3, 3> &getEigenVectors() { return vectors; }
	Eigen::Matrix<float, 3, 1> &getEigenValues() { return values; }
};

struct PointNormal
{
	float x = 0, y = 0, z = 0, d3 = 0;
	float normal_x = 0, normal_y = 0, normal_z = 0, n3 = 0;
	float curvature = 0, pad_[3] = {0, 0, 0};
};

namespace registration
{
// CorrespondenceEstimation<S,T>::determineCorrespondences(corrs, double max_distance) (PCL 1.8-1.10)
template <typename S, typename T>
struct CorrespondenceEstimation
{
	typename PointCloud<S>::Ptr source;
	typename PointCloud<T>::Ptr target;
	typename search::KdTree<T>::Ptr tree;
	void setInputCloud(const typename PointCloud<S>::Ptr &c) { source = c; } // deprecated alias of setInputSource
	void setInputSource(const typename PointCloud<S>::Ptr &c) { source = c; }
	void setInputTarget(const typename PointCloud<T>::Ptr &c) { target = c; }
	void setSearchMethodTarget(const typename search::KdTree<T>::Ptr &t, bool) { tree = t; }
	void determineCorrespondences(Correspondences &corrs, double max_distance)
	{
		const double max_dist_sqr = max_distance * max_distance;
		corrs.clear();
		std::vector<int> index(1);
		std::vector<float> distance(1);
		for (size_t i = 0; i < source->points.size(); i++)
		{
			if (tree->nearestKSearch(source->points[i], 1, index, distance) < 1)
				continue;
			if (distance[0] > max_dist_sqr)
				continue;
			Correspondence c;
			c.index_query = (int)i;
			c.index_match = index[0];
			c.distance = distance[0];
			corrs.push_back(c);
		}
	}
};

// CorrespondenceEstimationNormalShooting::determineCorrespondences (PCL 1.8-1.10)
template <typename S, typename T, typename N>
struct CorrespondenceEstimationNormalShooting
{
	typename PointCloud<S>::Ptr source;
	typename PointCloud<T>::Ptr target;
	typename PointCloud<N>::Ptr normals;
	typename search::KdTree<T>::Ptr tree;
	int k = 10;
	void setInputSource(const typename PointCloud<S>::Ptr &c) { source = c; }
	void setInputTarget(const typename PointCloud<T>::Ptr &c) { target = c; }
	void setSourceNormals(const typename PointCloud<N>::Ptr &c) { normals = c; }
	void setSearchMethodTarget(const typename search::KdTree<T>::Ptr &t, bool) { tree = t; }
	void setKSearch(int kk) { k = kk; }
	void determineCorrespondences(Correspondences &corrs, double max_distance)
	{
		corrs.clear();
		std::vector<int> nn_indices(k);
		std::vector<float> nn_dists(k);
		for (size_t i = 0; i < source->points.size(); i++)
		{
			int found = tree->nearestKSearch(source->points[i], k, nn_indices, nn_dists);
			double min_dist = std::numeric_limits<double>::max();
			int min_index = 0;
			for (int j = 0; j < found; j++)
			{
				const T &t = target->points[nn_indices[j]];
				float ptx = source->points[i].x - t.x, pty = source->points[i].y - t.y, ptz = source->points[i].z - t.z;
				const N &normal = normals->points[i];
				double Vx = ptx, Vy = pty, Vz = ptz, Nx = normal.normal_x, Ny = normal.normal_y, Nz = normal.normal_z;
				double cx = Ny * Vz - Nz * Vy, cy = Nz * Vx - Nx * Vz, cz = Nx * Vy - Ny * Vx;
				double dist = cx * cx + cy * cy + cz * cz;
				if (dist < min_dist)
				{
					min_dist = dist;
					min_index = j;
				}
			}
			if (found < 1 || min_dist > max_distance)
				continue;
			Correspondence c;
			c.index_query = (int)i;
			c.index_match = nn_indices[min_index];
			c.distance = nn_dists[min_index];
			corrs.push_back(c);
		}
	}
};

// CorrespondenceRejectorDistance without a data container (registration/src/correspondence_rejection_distance.cpp: `distance < max_distance_`;
// ref_driver.cpp switches to the `<=` reading when mulls_params.rejector_strict is 0, so both forms stay pinned against the reference lines)
inline bool &rejector_strict_flag()
{
	static bool strict = true;
	return strict;
}
struct CorrespondenceRejectorDistance
{
	boost::shared_ptr<Correspondences> input;
	float max_distance_ = FLT_MAX;
	void setInputCorrespondences(const boost::shared_ptr<Correspondences> &c) { input = c; }
	void setMaximumDistance(float d) { max_distance_ = d * d; }
	void getCorrespondences(Correspondences &out)
	{
		if (!input || input->empty())
			return; // PCL returns without touching the output
		Correspondences kept;
		for (size_t i = 0; i < input->size(); i++)
			if (rejector_strict_flag() ? (*input)[i].distance < max_distance_ : !((*input)[i].distance > max_distance_))
				kept.push_back((*input)[i]);
		out.swap(kept);
	}
};
} // namespace registration

// transformPointCloudWithNormals<PointT, double>(in, out, T), in == out allowed
template <typename PointT>
void transformPointCloudWithNormals(const PointCloud<PointT> &in, PointCloud<PointT> &out, const Eigen::Matrix4d &tf)
{
	if (&in != &out)
		out.points = in.points;
	for (size_t i = 0; i < out.points.size(); i++)
	{
		PointT &p = out.points[i];
		double x = p.x, y = p.y, z = p.z;
		p.x = static_cast<float>(tf(0, 0) * x + tf(0, 1) * y + tf(0, 2) * z + tf(0, 3));
		p.y = static_cast<float>(tf(1, 0) * x + tf(1, 1) * y + tf(1, 2) * z + tf(1, 3));
		p.z = static_cast<float>(tf(2, 0) * x + tf(2, 1) * y + tf(2, 2) * z + tf(2, 3));
		double nx = p.normal_x, ny = p.normal_y, nz = p.normal_z;
		p.normal_x = static_cast<float>(tf(0, 0) * nx + tf(0, 1) * ny + tf(0, 2) * nz);
		p.normal_y = static_cast<float>(tf(1, 0) * nx + tf(1, 1) * ny + tf(1, 2) * nz);
		p.normal_z = static_cast<float>(tf(2, 0) * nx + tf(2, 1) * ny + tf(2, 2) * nz);
	}
}

// transformPointCloud<PointT, double>(in, out, T): positions only
template <typename PointT>
void transformPointCloud(const PointCloud<PointT> &in, PointCloud<PointT> &out, const Eigen::Matrix4d &tf)
{
	if (&in != &out)
		out.points = in.points;
	for (size_t i = 0; i < out.points.size(); i++)
	{
		PointT &p = out.points[i];
		double x = p.x, y = p.y, z = p.z;
		p.x = static_cast<float>(tf(0, 0) * x + tf(0, 1) * y + tf(0, 2) * z + tf(0, 3));
		p.y = static_cast<float>(tf(1, 0) * x + tf(1, 1) * y + tf(1, 2) * z + tf(1, 3));
		p.z = static_cast<float>(tf(2, 0) * x + tf(2, 1) * y + tf(2, 2) * z + tf(2, 3));
	}
}

// pcl::RandomSample (selection sampling driven by rand()); the reference seeds it with time(NULL), this stand-in
// with a fixed seed — results of keep_less_source_points are not reproducible upstream either (SURVEY B-13)
template <typename PointT>
struct RandomSample
{
	typename PointCloud<PointT>::Ptr in;
	unsigned sample = 0;
	explicit RandomSample(bool = false) {}
	void setInputCloud(const typename PointCloud<PointT>::Ptr &c) { in = c; }
	void setSample(unsigned s) { sample = s; }
	void filter(PointCloud<PointT> &out)
	{
		out.points.clear();
		size_t N = in->points.size();
		if (sample >= N)
		{
			out.points = in->points;
			return;
		}
		std::srand(12345u);
		size_t top = N - sample, i = 0, index = 0;
		for (size_t n = sample; n >= 2; n--)
		{
			float V = (float)std::rand() / (float)RAND_MAX;
			size_t S = 0;
			float quot = (float)top / (float)N;
			while (quot > V)
			{
				S++;
				top--;
				N--;
				quot = quot * (float)top / (float)N;
			}
			index += S;
			out.points.push_back(in->points[index++]);
			N--;
			i++;
		}
		index += N * (size_t)((float)std::rand() / (float)RAND_MAX);
		out.points.push_back(in->points[std::min(index, in->points.size() - 1)]);
	}
};
} // namespace pcl

// what CFilter::fast_ground_filter calls for its ground normals besides the point cloud types (estimate_ground_normal_method 1 - 3): the
// stand-ins hand the work to oracle/pcl_restated.h, the one restatement of these PCL classes the oracle uses too — a comparison "oracle ==
// reference lines" therefore pins what MULLS wrote around the calls (which points go in, which come out, the j % rate rule, the abs(normal_z)
// gate, check_normal) and nothing inside them
namespace pcl
{
struct Normal
{
	float normal_x = 0, normal_y = 0, normal_z = 0, curvature = 0;
};
template <typename T>
inline bool isFinite(const T &p);
template <>
inline bool isFinite<Normal>(const Normal &n)
{
	return std::isfinite(n.normal_x) && std::isfinite(n.normal_y) && std::isfinite(n.normal_z);
}
struct ModelCoefficients
{
	typedef boost::shared_ptr<ModelCoefficients> Ptr;
	std::vector<float> values;
};
struct PointIndices
{
	typedef boost::shared_ptr<PointIndices> Ptr;
	std::vector<int> indices;
};
enum
{
	SACMODEL_PLANE = 0
};
enum
{
	SAC_RANSAC = 0
};
template <typename PointT>
struct SACSegmentation
{
	typename PointCloud<PointT>::Ptr cloud;
	double threshold = 0;
	int max_iter = 50;
	void setOptimizeCoefficients(bool) {}
	void setModelType(int) {}
	void setMethodType(int) {}
	void setDistanceThreshold(double t) { threshold = t; }
	void setMaxIterations(int m) { max_iter = m; }
	void setInputCloud(const typename PointCloud<PointT>::Ptr &c) { cloud = c; }
	void segment(PointIndices &inliers, ModelCoefficients &coefficients)
	{
		std::vector<restated::P4> pts(cloud->points.size());
		for (size_t i = 0; i < pts.size(); i++)
			pts[i] = restated::P4{cloud->points[i].x, cloud->points[i].y, cloud->points[i].z, cloud->points[i].data[3]};
		float c[4];
		inliers.indices.clear();
		coefficients.values.clear();
		if (restated::plane_ransac(pts, threshold, max_iter, inliers.indices, c))
			coefficients.values.assign(c, c + 4);
	}
};
template <typename PointT, typename NormalT>
struct NormalEstimationOMP
{
	typename PointCloud<PointT>::Ptr cloud;
	double radius = 0;
	int k = 0;
	void setNumberOfThreads(int) {}
	void setInputCloud(const typename PointCloud<PointT>::Ptr &c) { cloud = c; }
	void setSearchMethod(const typename search::KdTree<PointT>::Ptr &) {}
	void setRadiusSearch(double r) { radius = r; }
	void setKSearch(int kk) { k = kk; }
	void compute(PointCloud<NormalT> &out)
	{
		std::vector<float> n;
		restated::normal_estimation(cloud->points, radius, k, n);
		out.points.resize(cloud->points.size());
		for (size_t i = 0; i < out.points.size(); i++)
		{
			out.points[i].normal_x = n[3 * i];
			out.points[i].normal_y = n[3 * i + 1];
			out.points[i].normal_z = n[3 * i + 2];
		}
	}
};
} // namespace pcl
#define PCL_ERROR(...) ((void)0)
// OpenMP calls made by CFilter::apply_motion_compensation
inline void omp_set_num_threads(int) {}
inline int omp_get_max_threads() { return 1; }
