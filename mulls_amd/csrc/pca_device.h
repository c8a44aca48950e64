// pca_device.h — the neighbourhood PCA of the feature-extraction kernels (k_map_pca of map_kernels.hip, k_cl_pca of k_classify.hip):
// what pcl::PCA computes for pca.hpp:392-437, as the ABI defines it (include/mulls_hip.h) —
//   float centroid and float demeaned covariance summed in the neighbours' order, scaled by 1 / (n - 1),
//   eigen-decomposition of that float matrix by cyclic Jacobi rotations in double (upper triangle), eigenvalues descending,
//   every eigenvector with its largest component positive, results rounded to float, third direction = first x second.
// All of it in named scalars: nothing is indexed dynamically, so nothing lives in scratch memory.
#pragma once
#include <hip/hip_runtime.h>

namespace mulls_pca
{
// one Jacobi rotation annihilating a(p,q)
#define MULLS_PCA_ROT(app, aqq, apq, apr, aqr, vxp, vxq, vyp, vyq, vzp, vzq)                            \
	if (apq != 0.0)                                                                                      \
	{                                                                                                    \
		const double theta = (aqq - app) / (2.0 * apq);                                                  \
		const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));          \
		const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;                                          \
		/* A <- A J : columns p and q */                                                                 \
		const double c_pp = cs * app - sn * apq, c_pq = sn * app + cs * apq;                             \
		const double c_qp = cs * apq - sn * aqq, c_qq = sn * apq + cs * aqq;                             \
		const double c_rp = cs * apr - sn * aqr, c_rq = sn * apr + cs * aqr;                             \
		/* A <- J^T A : rows p and q; the upper triangle is the matrix */                                \
		app = cs * c_pp - sn * c_qp;                                                                     \
		aqq = sn * c_pq + cs * c_qq;                                                                     \
		apq = cs * c_pq - sn * c_qq;                                                                     \
		apr = c_rp;                                                                                      \
		aqr = c_rq;                                                                                      \
		double u = vxp, w = vxq;                                                                         \
		vxp = cs * u - sn * w, vxq = sn * u + cs * w;                                                    \
		u = vyp, w = vyq;                                                                                \
		vyp = cs * u - sn * w, vyq = sn * u + cs * w;                                                    \
		u = vzp, w = vzq;                                                                                \
		vzp = cs * u - sn * w, vzq = sn * u + cs * w;                                                    \
	}

struct Eig
{
	float e1, e2, e3;	 // eigenvalues, descending
	float px, py, pz;	 // first eigenvector (unit in double, rounded)
	float mx, my, mz;	 // second
};

// unit vector with its largest component positive, rounded to float
__device__ __forceinline__ void signed_unit(double x, double y, double z, float &fx, float &fy, float &fz)
{
	const double nrm = sqrt(x * x + y * y + z * z);
	int big = 0;
	double bigv = fabs(x);
	if (fabs(y) > bigv)
		big = 1, bigv = fabs(y);
	if (fabs(z) > bigv)
		big = 2;
	const double lead = big == 0 ? x : (big == 1 ? y : z);
	const double sgn = lead < 0 ? -1.0 : 1.0;
	fx = (float)(sgn * x / nrm), fy = (float)(sgn * y / nrm), fz = (float)(sgn * z / nrm);
}

// s0..s5 = xx xy xz yy yz zz of a float covariance: eigenvalues descending (the lower index first among equals) and their eigenvectors, raw (double)
struct EigD
{
	double e0, e1, e2;
	double x0, y0, z0, x1, y1, z1, x2, y2, z2;
};
__device__ __forceinline__ EigD eigen3_d(float s0, float s1, float s2, float s3, float s4, float s5)
{
	double a00 = (double)s0, a01 = (double)s1, a02 = (double)s2, a11 = (double)s3, a12 = (double)s4, a22 = (double)s5;
	double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
	for (int sweep = 0; sweep < 60; sweep++)
	{
		const double off = a01 * a01 + a02 * a02 + a12 * a12;
		if (off < 1e-300)
			break;
		MULLS_PCA_ROT(a00, a11, a01, a02, a12, v00, v01, v10, v11, v20, v21) // (0,1)
		MULLS_PCA_ROT(a00, a22, a02, a01, a12, v00, v02, v10, v12, v20, v22) // (0,2)
		MULLS_PCA_ROT(a11, a22, a12, a01, a02, v01, v02, v11, v12, v21, v22) // (1,2)
	}
	// descending selection, the lower index first among equals (three compare-and-swaps on value + column)
	EigD d;
	d.e0 = a00, d.e1 = a11, d.e2 = a22;
	d.x0 = v00, d.y0 = v10, d.z0 = v20, d.x1 = v01, d.y1 = v11, d.z1 = v21, d.x2 = v02, d.y2 = v12, d.z2 = v22;
#define MULLS_PCA_SWAP(ea, eb, xa, ya, za, xb, yb, zb) \
	if (eb > ea)                                       \
	{                                                  \
		double w = ea;                                 \
		ea = eb, eb = w;                               \
		w = xa, xa = xb, xb = w;                       \
		w = ya, ya = yb, yb = w;                       \
		w = za, za = zb, zb = w;                       \
	}
	MULLS_PCA_SWAP(d.e0, d.e1, d.x0, d.y0, d.z0, d.x1, d.y1, d.z1)
	MULLS_PCA_SWAP(d.e0, d.e2, d.x0, d.y0, d.z0, d.x2, d.y2, d.z2)
	MULLS_PCA_SWAP(d.e1, d.e2, d.x1, d.y1, d.z1, d.x2, d.y2, d.z2)
#undef MULLS_PCA_SWAP
	return d;
}
// ... the two leading directions as pcl::PCA's callers use them
__device__ __forceinline__ Eig eigen3(float s0, float s1, float s2, float s3, float s4, float s5)
{
	const EigD d = eigen3_d(s0, s1, s2, s3, s4, s5);
	Eig r;
	r.e1 = (float)d.e0, r.e2 = (float)d.e1, r.e3 = (float)d.e2;
	signed_unit(d.x0, d.y0, d.z0, r.px, r.py, r.pz);
	signed_unit(d.x1, d.y1, d.z1, r.mx, r.my, r.mz);
	return r;
}
// ... the direction of the smallest eigenvalue: the normal of the least-squares plane (ground normals, k_ground.hip)
__device__ __forceinline__ void smallest_eigenvector(float s0, float s1, float s2, float s3, float s4, float s5, float &nx, float &ny, float &nz)
{
	const EigD d = eigen3_d(s0, s1, s2, s3, s4, s5);
	signed_unit(d.x2, d.y2, d.z2, nx, ny, nz);
}

// Eigen::Vector3f::normalize()
__device__ __forceinline__ void normalize3(float &x, float &y, float &z)
{
	const float z2 = x * x + y * y + z * z;
	if (z2 > 0.f)
	{
		const float n = sqrtf(z2);
		x /= n, y /= n, z /= n;
	}
}
} // namespace mulls_pca
