// k_ground_normals.hip — estimate_ground_normal_method 1 / 2 of CFilter::fast_ground_filter (cfilter.hpp:1943-1954): pcl::NormalEstimationOMP over
// cloud_ground with setRadiusSearch(normal_estimation_radius) / setKSearch(2 * min_grid_pt_num) (pca.hpp:66-119) and check_normal (:462-475), as
// include/mulls_hip.h defines it and oracle/pcl_restated.h::normal_estimation states it:
//   neighbours ascending by (squared float distance, index), the query among them; fewer than three -> (0.577, 0.577, 0.577);
//   pcl::computeMeanAndCovarianceMatrix: nine float sums in that order, divided by the count, covariance = E[x x^T] - c c^T;
//   the plane's normal = the smallest eigenvector of that float matrix (Jacobi rotations in double, pca_device.h), turned towards the origin
//   (flipNormalTowardsViewpoint with the view point (0, 0, 0)).
// One wavefront per ground point on the uniform grid of k_classify.hip (cell = the radius; the k-nearest search sweeps a growing radius until k
// points lie inside it).  The in-radius candidates are buffered in LDS (1024 per query), ranked by counting, and summed 64 at a time.
#include <hip/hip_runtime.h>

#include "classify_launch.h"
#include "ground_launch.h"
#include "pca_device.h"

#define GN_CAP 1024u
#define GN_WSYNC() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")
namespace
{
struct GnLds
{
	float d[GN_CAP];
	uint32_t i[GN_CAP];
	uint32_t order[GN_CAP]; // indices ascending by (distance, index)
};
__device__ __forceinline__ int gn_cell_axis(double v, double lo, double cell, uint32_t dim)
{
	const double c = floor((v - lo) / cell);
	if (!(c > 0.0))
		return 0;
	if (c >= (double)dim)
		return (int)dim - 1;
	return (int)c;
}
__device__ __forceinline__ float gn_rl(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
// order[rank] = index of the rank-th smallest (distance, index) among the c buffered candidates (only ranks below `keep` are stored)
__device__ __forceinline__ void gn_rank(GnLds &L, uint32_t lane, uint32_t c, uint32_t keep)
{
	GN_WSYNC();
	for (uint32_t j = lane; j < c; j += 64u)
	{
		const float dj = L.d[j];
		const uint32_t ij = L.i[j];
		uint32_t rank = 0;
		for (uint32_t u = 0; u < c; u++)
		{
			const float du = L.d[u];
			const uint32_t iu = L.i[u];
			rank += (du < dj || (du == dj && iu < ij)) ? 1u : 0u;
		}
		if (rank < keep)
			L.order[rank] = j;
	}
	GN_WSYNC();
}
} // namespace

// radius > 0: every neighbour within it; else the K nearest.  error |= 4: a neighbourhood of more than GN_CAP points within the radius.
__global__ __launch_bounds__(256) void k_gf_normals(float4 *__restrict__ ground, uint32_t n, float radius, uint32_t K, const float4 *__restrict__ sorted,
													const uint32_t *__restrict__ cell_start, const ClGrid *__restrict__ g, uint32_t *__restrict__ error)
{
	__shared__ GnLds Lw[4];
	const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
	GnLds &L = Lw[w];
	const uint32_t i = blockIdx.x * 4u + w;
	if (i >= n)
		return;
	const float4 q0 = ground[(size_t)i * 3];
	const float qx = q0.x, qy = q0.y, qz = q0.z;
	const double cell = (double)g->cell;
	const uint32_t d1 = g->dim[1], d2n = g->dim[2];
	// the reach beyond which the grid holds nothing: the k-nearest sweep stops growing there
	const double span = (double)g->cell * (double)(g->dim[0] + g->dim[1] + g->dim[2] + 3u);
	double R = radius > 0.0f ? (double)radius : 1.0;
	uint32_t c = 0, m = 0;
	for (;;)
	{
		const float r2 = (float)(R * R);
		const double reach = R * (1.0 + 1e-6) + 1e-6;
		const int x0 = gn_cell_axis((double)qx - reach, (double)g->lo[0], cell, g->dim[0]), x1 = gn_cell_axis((double)qx + reach, (double)g->lo[0], cell, g->dim[0]);
		const int y0 = gn_cell_axis((double)qy - reach, (double)g->lo[1], cell, d1), y1 = gn_cell_axis((double)qy + reach, (double)g->lo[1], cell, d1);
		const int z0 = gn_cell_axis((double)qz - reach, (double)g->lo[2], cell, d2n), z1 = gn_cell_axis((double)qz + reach, (double)g->lo[2], cell, d2n);
		c = 0;
		bool pruned = false, overflow = false;
		float thr_d = r2;
		uint32_t thr_i = 0;
		for (int cx = x0; cx <= x1 && !overflow; cx++)
			for (int cy = y0; cy <= y1 && !overflow; cy++)
			{
				const uint32_t row = ((uint32_t)cx * d1 + (uint32_t)cy) * d2n;
				const uint32_t s0 = cell_start[row + (uint32_t)z0], s1 = cell_start[row + (uint32_t)z1 + 1u]; // the z cells of one column are contiguous
				for (uint32_t base = s0; base < s1; base += 64u)
				{
					const uint32_t s = base + lane;
					bool accept = false;
					float d = 0.f;
					uint32_t idx = 0;
					if (s < s1)
					{
						const float4 cpt = sorted[s];
						float diff = qx - cpt.x;
						d = 0.0f + diff * diff; // L2_Simple<float>: result += diff * diff, x then y then z
						diff = qy - cpt.y;
						d += diff * diff;
						diff = qz - cpt.z;
						d += diff * diff;
						idx = __float_as_uint(cpt.w);
						accept = pruned ? (d < thr_d || (d == thr_d && idx < thr_i)) : (d < r2);
					}
					const unsigned long long bal = __ballot(accept);
					const uint32_t add = (uint32_t)__popcll(bal);
					if (c + add > GN_CAP)
					{
						if (radius > 0.0f)
						{
							overflow = true; // a radius search keeps every neighbour: more than the buffer holds
							break;
						}
						// k nearest: the buffer's K best stay, their last entry becomes the admission threshold
						gn_rank(L, lane, c, K);
						float kd = 0.f;
						uint32_t ki = 0;
						if (lane < K)
						{
							kd = L.d[L.order[lane]];
							ki = L.i[L.order[lane]];
						}
						GN_WSYNC();
						if (lane < K)
						{
							L.d[lane] = kd;
							L.i[lane] = ki;
						}
						c = K;
						pruned = true;
						thr_d = gn_rl(kd, (int)K - 1);
						thr_i = (uint32_t)__builtin_amdgcn_readlane((int)ki, (int)K - 1);
						GN_WSYNC();
						// (the pending candidates were tested against the old threshold: admitting them is still correct, the ranking decides)
					}
					if (accept)
					{
						const uint32_t pos = c + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
						L.d[pos] = d;
						L.i[pos] = idx;
					}
					c += add;
				}
			}
		if (overflow)
		{
			if (lane == 0)
				atomicOr(error, 4u);
			return;
		}
		if (radius > 0.0f)
		{
			m = c;
			break;
		}
		if (c >= K || R > span)
		{
			m = c < K ? c : K;
			break;
		}
		R *= 2.0;
	}
	float nx = (float)0.577, ny = (float)0.577, nz = (float)0.577; // check_normal (pca.hpp:467-472)
	if (m >= 3u)
	{
		gn_rank(L, lane, c, m);
		float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, a8 = 0;
		for (uint32_t k0 = 0; k0 < m; k0 += 64u)
		{
			const uint32_t k = k0 + lane;
			float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
			if (k < m)
				p = ground[(size_t)L.i[L.order[k]] * 3];
			const int cnt = (int)min(64u, m - k0);
			for (int l = 0; l < cnt; l++)
			{
				const float x = gn_rl(p.x, l), y = gn_rl(p.y, l), z = gn_rl(p.z, l);
				a0 += x * x;
				a1 += x * y;
				a2 += x * z;
				a3 += y * y;
				a4 += y * z;
				a5 += z * z;
				a6 += x;
				a7 += y;
				a8 += z;
			}
		}
		const float nm = (float)m;
		a0 /= nm, a1 /= nm, a2 /= nm, a3 /= nm, a4 /= nm, a5 /= nm, a6 /= nm, a7 /= nm, a8 /= nm;
		const float v0 = a0 - a6 * a6, v1 = a1 - a6 * a7, v2 = a2 - a6 * a8, v3 = a3 - a7 * a7, v4 = a4 - a7 * a8, v5 = a5 - a8 * a8;
		mulls_pca::smallest_eigenvector(v0, v1, v2, v3, v4, v5, nx, ny, nz);
		// flipNormalTowardsViewpoint(point, 0, 0, 0, nx, ny, nz)
		const float vx = 0.0f - qx, vy = 0.0f - qy, vz = 0.0f - qz;
		const float cos_theta = (vx * nx + vy * ny + vz * nz);
		if (cos_theta < 0)
			nx *= -1, ny *= -1, nz *= -1;
		if (!(isfinite(nx) && isfinite(ny) && isfinite(nz)))
			nx = ny = nz = (float)0.577;
	}
	if (lane == 0)
	{
		float4 b = ground[(size_t)i * 3 + 1];
		b.x = nx, b.y = ny, b.z = nz;
		ground[(size_t)i * 3 + 1] = b;
	}
}

// scratch layout: ClGrid | seg_sum | cellof[n] | cell_start | cell_fill | sorted[n]
size_t ground_normals_bytes(uint32_t n)
{
	return 256 + 1040 * 4 + (size_t)n * 4 + 2 * ((size_t)MULLS_CL_MAX_CELLS + 2) * 4 + 64 + (size_t)n * 16 + 256;
}
void launch_ground_normals(hipStream_t st, float4 *ground, uint32_t n, float radius, int k, void *grid_mem, uint32_t *error)
{
	if (!n)
		return;
	unsigned char *b = static_cast<unsigned char *>(grid_mem);
	ClArrays A = {};
	A.recs = ground;
	A.grid = reinterpret_cast<ClGrid *>(b);
	b += 256;
	A.seg_sum = reinterpret_cast<uint32_t *>(b);
	b += 1040 * 4;
	A.cellof = reinterpret_cast<uint32_t *>(b);
	b += (size_t)n * 4;
	A.cell_start = reinterpret_cast<uint32_t *>(b);
	b += ((size_t)MULLS_CL_MAX_CELLS + 2) * 4;
	A.cell_fill = reinterpret_cast<uint32_t *>(b);
	b += ((size_t)MULLS_CL_MAX_CELLS + 2) * 4;
	b = reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(b) + 63) & ~(uintptr_t)63);
	A.sorted = reinterpret_cast<float4 *>(b);
	ClParams P = {};
	P.n = n;
	P.radius = radius > 0.0f ? radius : 1.0f; // the grid's cell edge
	launch_cl_grid(st, A, P);
	hipLaunchKernelGGL(k_gf_normals, dim3((n + 3u) / 4u), dim3(256), 0, st, ground, n, radius, (uint32_t)(k > 0 ? k : 0), A.sorted, A.cell_start, A.grid, error);
}
