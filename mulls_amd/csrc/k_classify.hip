// k_classify.hip — CFilter::classify_nground_pts (include/common/cfilter.hpp:2058-2290) on the device: the neighbourhood PCA of every
// non-ground point (pca.hpp:292-352, :392-456), the class decision, the promotion of high-curvature points (:2160-2198), the key-point
// descriptors (encode_stable_points, :1071-1181) and non-maximum suppression (:1243-1312).  Host side: classify.cpp.
//
// A scan's non-ground cloud is 20-60 k points: the whole working set (48-B records, a [K][n] neighbour table, per-point features) is a few
// tens of MB and stays in HBM / L2 between the passes.  The two loops upstream that are sequential by nature (promotion, suppression) run
// as fixed-point rounds that settle every point whose predecessors are settled.
//   k_cl_bbox / _setup / _count / scan / _scatter   uniform search grid (cell = radius, coarsened to fit 4 M cells), points cell-sorted
//   k_cl_knn        a wavefront per query point: the in-radius candidates buffered in LDS, the neighbor_k nearest by rank counting, the sums
//   k_cl_eig        a lane per query point: eigen-decomposition, pca_feature_t
//   k_cl_label      class decision per point, normals written as the reference writes them
//   k_cl_promote    fixed-point rounds of the one loop upstream that reads labels it has just written (:2166-2197)
//   k_cl_encode     key points and their neighbourhood descriptors; masks for the stable compactions (map_kernels.hip)
//   k_cl_nms_*      greedy suppression in visiting order as earlier-neighbour lists + fixed-point rounds
#include <hip/hip_runtime.h>

#include "classify_launch.h"
#include "pca_device.h"

namespace
{
__device__ __forceinline__ uint32_t ford(float f)
{
	const uint32_t u = __float_as_uint(f);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ford_inv(uint32_t k)
{
	return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ bool finite3(float x, float y, float z) { return isfinite(x) && isfinite(y) && isfinite(z); }
// cell coordinate along one axis; double so that it is monotone in the coordinate (the query's box is computed the same way)
__device__ __forceinline__ int cell_axis(double v, double lo, double cell, uint32_t dim)
{
	const double c = floor((v - lo) / cell);
	if (!(c > 0.0))
		return 0;
	if (c >= (double)dim)
		return (int)dim - 1;
	return (int)c;
}
} // namespace

__global__ __launch_bounds__(256) void k_cl_bbox(const float4 *__restrict__ recs, uint32_t n, ClGrid *g)
{
	__shared__ uint32_t s[7][4];
	uint32_t k[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u}, bad = 0;
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u)
	{
		const float4 r0 = recs[(size_t)i * 3];
		if (!finite3(r0.x, r0.y, r0.z))
		{
			bad++;
			continue;
		}
		k[0] = min(k[0], ford(r0.x)), k[1] = min(k[1], ford(r0.y)), k[2] = min(k[2], ford(r0.z));
		k[3] = max(k[3], ford(r0.x)), k[4] = max(k[4], ford(r0.y)), k[5] = max(k[5], ford(r0.z));
	}
	for (int off = 32; off; off >>= 1)
	{
		for (int c = 0; c < 3; c++)
		{
			k[c] = min(k[c], (uint32_t)__shfl_xor((int)k[c], off));
			k[3 + c] = max(k[3 + c], (uint32_t)__shfl_xor((int)k[3 + c], off));
		}
		bad += (uint32_t)__shfl_xor((int)bad, off);
	}
	const uint32_t w = threadIdx.x >> 6;
	if ((threadIdx.x & 63u) == 0)
	{
		for (int c = 0; c < 6; c++)
			s[c][w] = k[c];
		s[6][w] = bad;
	}
	__syncthreads();
	if (threadIdx.x < 7)
	{
		const uint32_t c = threadIdx.x;
		if (c < 3)
			atomicMin(&g->keys[c], min(min(s[c][0], s[c][1]), min(s[c][2], s[c][3])));
		else if (c < 6)
			atomicMax(&g->keys[c], max(max(s[c][0], s[c][1]), max(s[c][2], s[c][3])));
		else
			atomicAdd(&g->nonfinite, s[6][0] + s[6][1] + s[6][2] + s[6][3]);
	}
}

__global__ void k_cl_setup(ClGrid *g, float cell_hint)
{
	if (threadIdx.x || blockIdx.x)
		return;
	float cell = cell_hint > 1e-3f ? cell_hint : 1e-3f;
	double lo[3], hi[3];
	for (int c = 0; c < 3; c++)
	{
		lo[c] = (double)ford_inv(g->keys[c]);
		hi[c] = (double)ford_inv(g->keys[3 + c]);
		g->lo[c] = ford_inv(g->keys[c]);
	}
	for (;;)
	{
		double cells = 1.0;
		for (int c = 0; c < 3; c++)
		{
			const double d = floor((hi[c] - lo[c]) / (double)cell) + 1.0;
			g->dim[c] = d < 4294967295.0 ? (uint32_t)d : 0xffffffffu;
			cells *= d;
		}
		if (cells <= (double)MULLS_CL_MAX_CELLS)
			break;
		cell *= 2.f;
	}
	g->cell = cell;
	g->ncell = g->dim[0] * g->dim[1] * g->dim[2];
}

__global__ __launch_bounds__(256) void k_cl_count(const float4 *__restrict__ recs, uint32_t n, const ClGrid *__restrict__ g, uint32_t *__restrict__ cellof,
												  uint32_t *__restrict__ cnt)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n)
		return;
	const float4 r0 = recs[(size_t)i * 3];
	const double cell = (double)g->cell;
	const uint32_t cx = (uint32_t)cell_axis((double)r0.x, (double)g->lo[0], cell, g->dim[0]), cy = (uint32_t)cell_axis((double)r0.y, (double)g->lo[1], cell, g->dim[1]),
				   cz = (uint32_t)cell_axis((double)r0.z, (double)g->lo[2], cell, g->dim[2]);
	const uint32_t c = (cx * g->dim[1] + cy) * g->dim[2] + cz;
	cellof[i] = c;
	atomicAdd(&cnt[c], 1u);
}

// exclusive prefix sum of v[0 .. ncell] (ncell + 1 entries, v[ncell] = 0 on entry) in three steps over 4096-entry segments
#define CL_SEG 4096u
__global__ __launch_bounds__(256) void k_cl_scan_seg(const uint32_t *__restrict__ v, const ClGrid *__restrict__ g, uint32_t *__restrict__ seg_sum)
{
	__shared__ uint32_t s[4];
	const uint32_t total = g->ncell + 1u, base = blockIdx.x * CL_SEG;
	if (base >= total)
		return;
	uint32_t acc = 0;
	for (uint32_t j = threadIdx.x; j < CL_SEG; j += 256u)
		if (base + j < total)
			acc += v[base + j];
	for (int off = 32; off; off >>= 1)
		acc += (uint32_t)__shfl_xor((int)acc, off);
	if ((threadIdx.x & 63u) == 0)
		s[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0)
		seg_sum[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ __launch_bounds__(1024) void k_cl_scan_top(uint32_t *__restrict__ seg_sum, const ClGrid *__restrict__ g)
{
	__shared__ uint32_t s[1025];
	const uint32_t nseg = (g->ncell + 1u + CL_SEG - 1u) / CL_SEG; // <= 1025
	for (uint32_t j = threadIdx.x; j < 1025u; j += 1024u)
		s[j] = j < nseg ? seg_sum[j] : 0u;
	__syncthreads();
	if (threadIdx.x == 0)
	{
		uint32_t run = 0;
		for (uint32_t j = 0; j < nseg; j++)
		{
			const uint32_t c = s[j];
			s[j] = run;
			run += c;
		}
	}
	__syncthreads();
	for (uint32_t j = threadIdx.x; j < nseg; j += 1024u)
		seg_sum[j] = s[j];
}
__global__ __launch_bounds__(256) void k_cl_scan_apply(uint32_t *__restrict__ v, const ClGrid *__restrict__ g, const uint32_t *__restrict__ seg_sum, uint32_t *__restrict__ copy)
{
	__shared__ uint32_t wsum[4];
	const uint32_t total = g->ncell + 1u, base = blockIdx.x * CL_SEG;
	if (base >= total)
		return;
	// lane t owns entries [16 t, 16 t + 16) of the segment
	const uint32_t t = threadIdx.x, j0 = base + t * 16u;
	uint32_t loc[16], acc = 0;
#pragma unroll
	for (int k = 0; k < 16; k++)
	{
		loc[k] = (j0 + k < total) ? v[j0 + k] : 0u;
		acc += loc[k];
	}
	uint32_t incl = acc;
	for (int off = 1; off < 64; off <<= 1)
	{
		const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
		if ((t & 63u) >= (uint32_t)off)
			incl += o;
	}
	if ((t & 63u) == 63u)
		wsum[t >> 6] = incl;
	__syncthreads();
	uint32_t run = seg_sum[blockIdx.x] + incl - acc;
	for (uint32_t w = 0; w < (t >> 6); w++)
		run += wsum[w];
#pragma unroll
	for (int k = 0; k < 16; k++)
		if (j0 + k < total)
		{
			v[j0 + k] = run;
			copy[j0 + k] = run;
			run += loc[k];
		}
}
__global__ __launch_bounds__(256) void k_cl_scatter(const float4 *__restrict__ recs, uint32_t n, const uint32_t *__restrict__ cellof, uint32_t *__restrict__ fill,
													float4 *__restrict__ sorted)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n)
		return;
	const float4 r0 = recs[(size_t)i * 3];
	const uint32_t pos = atomicAdd(&fill[cellof[i]], 1u);
	sorted[pos] = make_float4(r0.x, r0.y, r0.z, __uint_as_float(i));
}

// ---------------------------------------------------------------------------------------------------------------------
// get_pc_pca_feature with the kd-tree argument (pca.hpp:292-352) + get_pca_feature (:392-437), in two kernels.
// k_cl_knn: one wavefront per query point (a scan's non-ground cloud is only 20-60 k queries: a lane per query leaves the chip to one
// latency-bound wavefront per CU, and sub-groups of a wavefront diverge at every prune).  The 64 lanes stream the candidates of the 3 x 3
// cell columns around the query (coalesced), append those within the radius to the query's buffer in LDS (ballot prefix), and pick the
// neighbor_k smallest (distance, index) pairs by rank counting — every lane ranks its share of the buffer against all of it, four
// candidates per LDS read.  A full buffer is pruned to its neighbor_k best, whose last entry becomes the admission threshold.  The sums of
// the PCA are sequential in the neighbours' order (that order is part of the result's rounding): one lane per sum.
// k_cl_eig: one lane per query, the 3 x 3 eigen-decomposition and what pca_feature_t derives from it.
#define CL_CAND 256u // candidate buffer entries per query
#define WSYNC() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")
namespace
{
struct KnnLds
{
	float4 cand_d[CL_CAND / 4u]; // squared distances
	uint4 cand_i[CL_CAND / 4u];	 // indices
	float sel_d[MULLS_CL_MAX_K];
	uint32_t sel_i[MULLS_CL_MAX_K];
	float sel_x[MULLS_CL_MAX_K], sel_y[MULLS_CL_MAX_K], sel_z[MULLS_CL_MAX_K];
};
// the K smallest (distance, index) of the c buffered candidates, ascending, into sel_*; returns min(c, K)
__device__ __forceinline__ uint32_t knn_select(KnnLds &L, uint32_t lane, uint32_t c, uint32_t K)
{
	float *cd = reinterpret_cast<float *>(L.cand_d);
	uint32_t *ci = reinterpret_cast<uint32_t *>(L.cand_i);
	const uint32_t c4 = (c + 3u) & ~3u;
	if (lane < c4 - c) // pad to a multiple of four with entries that rank behind everything
	{
		cd[c + lane] = __builtin_inff();
		ci[c + lane] = 0xffffffffu;
	}
	WSYNC();
	for (uint32_t j = lane; j < c; j += 64u)
	{
		const float dj = cd[j];
		const uint32_t ij = ci[j];
		uint32_t rank = 0;
		for (uint32_t u = 0; u < c4 / 4u; u++)
		{
			const float4 d4 = L.cand_d[u];
			const uint4 i4 = L.cand_i[u];
			rank += (d4.x < dj || (d4.x == dj && i4.x < ij)) ? 1u : 0u;
			rank += (d4.y < dj || (d4.y == dj && i4.y < ij)) ? 1u : 0u;
			rank += (d4.z < dj || (d4.z == dj && i4.z < ij)) ? 1u : 0u;
			rank += (d4.w < dj || (d4.w == dj && i4.w < ij)) ? 1u : 0u;
		}
		if (rank < K)
		{
			L.sel_d[rank] = dj;
			L.sel_i[rank] = ij;
		}
	}
	WSYNC();
	return c < K ? c : K;
}
} // namespace

__global__ __launch_bounds__(256) void k_cl_knn(ClArrays A, ClParams P)
{
	__shared__ KnnLds Lw[4];
	const uint32_t t = threadIdx.x, w = t >> 6, lane = t & 63u;
	KnnLds &L = Lw[w];
	const uint64_t i64 = ((uint64_t)blockIdx.x * 4u + w) * (uint64_t)P.down_rate;
	if (i64 >= P.n)
		return; // the whole wavefront leaves
	const uint32_t i = (uint32_t)i64, n = P.n, K = P.K;
	const float4 q0 = A.recs[(size_t)i * 3];
	const float qx = q0.x, qy = q0.y, qz = q0.z;
	float neighborhood_r = P.radius;
	if (P.adaptive)
	{
		const double dist = (double)sqrtf(qx * qx + qy * qy + qz * qz);
		if (dist > (double)P.unit_distance)
			neighborhood_r = (float)(sqrt(dist / (double)P.unit_distance) * (double)P.radius);
	}
	const float r2 = (float)((double)neighborhood_r * (double)neighborhood_r);
	const ClGrid *g = A.grid;
	const double cell = (double)g->cell, reach = (double)neighborhood_r * (1.0 + 1e-6) + 1e-6;
	const uint32_t d1 = g->dim[1], d2n = g->dim[2];
	const int x0 = cell_axis((double)qx - reach, (double)g->lo[0], cell, g->dim[0]), x1 = cell_axis((double)qx + reach, (double)g->lo[0], cell, g->dim[0]);
	const int y0 = cell_axis((double)qy - reach, (double)g->lo[1], cell, d1), y1 = cell_axis((double)qy + reach, (double)g->lo[1], cell, d1);
	const int z0 = cell_axis((double)qz - reach, (double)g->lo[2], cell, d2n), z1 = cell_axis((double)qz + reach, (double)g->lo[2], cell, d2n);
	float *cd = reinterpret_cast<float *>(L.cand_d);
	uint32_t *ci = reinterpret_cast<uint32_t *>(L.cand_i);
	uint32_t c = 0;		  // buffered candidates (wave-uniform)
	bool pruned = false; // after a prune a candidate has to beat the kept list's last entry
	float thr_d = r2;
	uint32_t thr_i = 0;
	for (int cx = x0; cx <= x1; cx++)
		for (int cy = y0; cy <= y1; cy++)
		{
			const uint32_t row = ((uint32_t)cx * d1 + (uint32_t)cy) * d2n;
			const uint32_t s0 = A.cell_start[row + (uint32_t)z0], s1 = A.cell_start[row + (uint32_t)z1 + 1u]; // the z cells of one column are contiguous
			for (uint32_t base = s0; base < s1; base += 64u)
			{
				const uint32_t s = base + lane;
				bool accept = false;
				float d = 0.f;
				uint32_t idx = 0;
				if (s < s1)
				{
					const float4 cpt = A.sorted[s];
					const float dx = qx - cpt.x, dy = qy - cpt.y, dz = qz - cpt.z;
					d = dx * dx + dy * dy + dz * dz;
					idx = __float_as_uint(cpt.w);
					accept = pruned ? (d < thr_d || (d == thr_d && idx < thr_i)) : (d < r2);
				}
				const unsigned long long bal = __ballot(accept);
				const uint32_t add = (uint32_t)__popcll(bal);
				if (c + add > CL_CAND) // make room first: the buffer's neighbor_k best stay
				{
					const uint32_t m = knn_select(L, lane, c, K);
					if (lane < m)
					{
						cd[lane] = L.sel_d[lane];
						ci[lane] = L.sel_i[lane];
					}
					c = m;
					if (m == K)
					{
						pruned = true;
						thr_d = L.sel_d[K - 1u];
						thr_i = L.sel_i[K - 1u];
					}
					WSYNC();
					// the pending candidates face the new threshold as well (admitting them anyway would still be correct: the selection decides)
				}
				if (accept)
				{
					const uint32_t pos = c + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
					cd[pos] = d;
					ci[pos] = idx;
				}
				c += add;
			}
		}
	const uint32_t m = knn_select(L, lane, c, K);
	// features[i].pt_num, neighbor_indices, close_to_query_point (squared_distances[j] < 0.64 * radius * radius, in double)
	const double close_thr = 0.64 * (double)P.radius * (double)P.radius;
	bool close = false;
	if (lane < m)
	{
		const uint32_t j = L.sel_i[lane];
		A.nbr[(size_t)lane * n + i] = j;
		close = (double)L.sel_d[lane] < close_thr;
		const float4 r0 = A.recs[(size_t)j * 3];
		L.sel_x[lane] = r0.x, L.sel_y[lane] = r0.y, L.sel_z[lane] = r0.z;
	}
	const unsigned long long bits = __ballot(close);
	WSYNC();
	float out = 0.f; // lanes 0..5: xx xy xz yy yz zz of the scaled covariance
	if (m > 3u)
	{
		// lane 0 / 1 / 2: the centroid's x / y / z; lanes 0..5: one covariance sum each — every sum sequential in the neighbours' order
		float mean = 0;
		if (lane < 3u)
		{
			const float *v = lane == 0 ? L.sel_x : (lane == 1 ? L.sel_y : L.sel_z);
			for (uint32_t k = 0; k < m; k++)
				mean += v[k];
			mean /= (float)m;
		}
		const float mx = __shfl(mean, 0), my = __shfl(mean, 1), mz = __shfl(mean, 2);
		if (lane < 6u)
		{
			const float *va = lane < 3u ? L.sel_x : (lane < 5u ? L.sel_y : L.sel_z);
			const float *vb = (lane == 0) ? L.sel_x : ((lane == 1 || lane == 3) ? L.sel_y : L.sel_z);
			const float ma = lane < 3u ? mx : (lane < 5u ? my : mz);
			const float mb = (lane == 0) ? mx : ((lane == 1 || lane == 3) ? my : mz);
			float sum = 0;
			for (uint32_t k = 0; k < m; k++)
				sum += (va[k] - ma) * (vb[k] - mb);
			const float alpha = 1.f / ((float)m - 1.f);
			out = alpha * sum;
		}
	}
	if (lane < 6u)
		A.cov[(size_t)lane * n + i] = out;
	if (lane == 0)
	{
		A.closebits[i] = bits;
		A.f_cnt[i] = (int32_t)m;
	}
}

__global__ __launch_bounds__(256) void k_cl_eig(ClArrays A, ClParams P)
{
	const uint32_t q = blockIdx.x * 256u + threadIdx.x;
	const uint64_t i64 = (uint64_t)q * (uint64_t)P.down_rate;
	if (i64 >= P.n)
		return;
	const uint32_t i = (uint32_t)i64, n = P.n;
	const int m = A.f_cnt[i];
	double curvature = 0, linear_2 = 0, planar_2 = 0;
	float px = 0, py = 0, pz = 0, nx = 0, ny = 0, nz = 0;
	if (m > 3)
	{
		const mulls_pca::Eig E = mulls_pca::eigen3(A.cov[i], A.cov[(size_t)n + i], A.cov[(size_t)2 * n + i], A.cov[(size_t)3 * n + i], A.cov[(size_t)4 * n + i],
												  A.cov[(size_t)5 * n + i]);
		px = E.px, py = E.py, pz = E.pz;
		nx = E.py * E.mz - E.pz * E.my; // col(2) = col(0).cross(col(1))
		ny = E.pz * E.mx - E.px * E.mz;
		nz = E.px * E.my - E.py * E.mx;
		mulls_pca::normalize3(px, py, pz);
		mulls_pca::normalize3(nx, ny, nz);
		const double l1 = (double)E.e1, l2 = (double)E.e2, l3 = (double)E.e3;
		curvature = ((l1 + l2 + l3) == 0) ? 0.0 : l3 / (l1 + l2 + l3);
		linear_2 = (l1 - l2) / l1;
		planar_2 = (l2 - l3) / l1;
	}
	A.f_curv[i] = curvature;
	A.f_lin[i] = linear_2;
	A.f_pla[i] = planar_2;
	A.f_pd[i] = make_float4(px, py, pz, 0.f);
	A.f_nd[i] = make_float4(nx, ny, nz, 0.f);
	if (m > 1) // features[i].pt_num > min_k (= 1): assign_normal(in_cloud->points[i], features[i]) — the plane's normal and planarity
		A.recs[(size_t)i * 3 + 1] = make_float4(nx, ny, nz, (float)planar_2);
}

// ---------------------------------------------------------------------------------------------------------------------
// the class decision (cfilter.hpp:2105-2158).  lab: 0 none, 1 pillar, 2 beam, 3 facade, 4 roof; down: the same codes for the *_down clouds
// of sharpen_with_nms = false; cand: bit 0 = candidate of the promotion loop (:2170), bit 1 / 2 = would become a pillar / beam there
__global__ __launch_bounds__(256) void k_cl_label(ClArrays A, ClParams P)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= P.n)
		return;
	const int pt_num = A.f_cnt[i];
	uint8_t lab = 0, down = 0, cand = 0;
	if (pt_num > P.k_min)
	{
		const double lin = A.f_lin[i], pla = A.f_pla[i];
		const float4 pd = A.f_pd[i], nd = A.f_nd[i];
		const float z = A.recs[(size_t)i * 3].z;
		const bool pillar_dir = fabsf(pd.z) > P.lin_high, beam_dir = !pillar_dir && fabsf(pd.z) < P.lin_low && z < P.beam_height_max;
		if (lin > (double)P.edge_thre)
		{
			if (pillar_dir)
				lab = 1;
			else if (beam_dir)
				lab = 2;
			if (lab)
				A.recs[(size_t)i * 3 + 1] = make_float4(pd.x, pd.y, pd.z, (float)lin); // assign_normal(pt, feature, false)
			if (!P.nms && lin > (double)P.edge_thre_down)
				down = lab;
		}
		else if (pla > (double)P.planar_thre)
		{
			if (fabsf(nd.z) > P.pla_high && z > P.roof_height_min)
				lab = 4;
			else if (fabsf(nd.z) < P.pla_low)
				lab = 3;
			if (lab)
				A.recs[(size_t)i * 3 + 1] = make_float4(nd.x, nd.y, nd.z, (float)pla); // assign_normal(pt, feature, true)
			if (!P.nms && pla > (double)P.planar_thre_down)
				down = lab;
		}
		if (lab == 0 && P.vertex_method == 2 && A.f_curv[i] > (double)P.curvature_thre)
			cand = (uint8_t)(1u | (pillar_dir ? 2u : 0u) | (beam_dir ? 4u : 0u));
	}
	A.lab[i] = lab;
	A.down[i] = down;
	A.cand[i] = cand;
	A.plab[i] = 0;
	A.cstate[i] = (cand & 1u) ? 1 : 0; // 1 undecided, 2 decided: stays, 3 decided: promoted
}

// One round of the promotion loop (:2166-2197).  Upstream walks the points in order and counts, for a candidate, the neighbours that carry
// a label at that moment — labels of the first pass plus those the loop itself gave to candidates of lower index.  A candidate is decided as
// soon as the undecided lower-index candidates in its neighbourhood cannot change its verdict; the lowest undecided one always is.
__global__ __launch_bounds__(256) void k_cl_promote(ClArrays A, ClParams P, uint32_t round)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= P.n)
		return;
	if (A.cstate[i] != 1)
		return;
	const int pt_num = A.f_cnt[i];
	const int listed = pt_num > 3 ? pt_num : 0; // neighbor_indices is only filled when the PCA ran (pca.hpp:396-397)
	const uint8_t *cstate = A.cstate;
	const double thr = (double)P.vertex_ratio_thre;
	for (int look = 0; look < 1; look++) // (more looks per launch only pay with cheap coherent loads: they go to memory here)
	{
		uint32_t lo = 0, hi = 0;
		for (int k = 0; k < listed; k++)
		{
			const uint32_t j = A.nbr[(size_t)k * P.n + i];
			if (A.lab[j])
				lo++, hi++;
			else if (j < i && (A.cand[j] & 6u))
			{
				const uint8_t s = cstate[j];
				if (s == 3)
					lo++, hi++;
				else if (s == 1)
					hi++;
			}
		}
		const bool pass = 1.0 * (double)lo / (double)pt_num > thr, may = 1.0 * (double)hi / (double)pt_num > thr;
		if (pass)
		{
			const float4 pd = A.f_pd[i];
			A.recs[(size_t)i * 3 + 1] = make_float4(pd.x, pd.y, pd.z, (float)(5.0 * A.f_curv[i])); // assign_normal(.., false); normal[3] = 5.0 * curvature
			const uint8_t c = A.cand[i];
			A.plab[i] = (c & 2u) ? 1 : ((c & 4u) ? 2 : 0);
			__threadfence();
			A.cstate[i] = 3;
			return;
		}
		if (!may)
		{
			A.cstate[i] = 2;
			return;
		}
		__threadfence();
	}
	atomicAdd(&A.round_cnt[round], 1u);
}

// encode_stable_points (:1071-1181) into vtx[i] + the eleven masks of the stable compactions:
// 0 pillar (first pass), 1 pillar (promoted), 2 beam, 3 beam (promoted), 4 facade, 5 roof, 6 key point, 7..10 pillar / beam / facade / roof down
__global__ __launch_bounds__(256) void k_cl_encode(ClArrays A, ClParams P)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x, n = P.n;
	if (i >= n)
		return;
	const uint8_t lab = A.lab[i], plab = A.plab[i], down = A.down[i];
	A.mask[0 * (size_t)n + i] = lab == 1;
	A.mask[1 * (size_t)n + i] = plab == 1;
	A.mask[2 * (size_t)n + i] = lab == 2;
	A.mask[3 * (size_t)n + i] = plab == 2;
	A.mask[4 * (size_t)n + i] = lab == 3;
	A.mask[5 * (size_t)n + i] = lab == 4;
	A.mask[7 * (size_t)n + i] = down == 1;
	A.mask[8 * (size_t)n + i] = down == 2;
	A.mask[9 * (size_t)n + i] = down == 3;
	A.mask[10 * (size_t)n + i] = down == 4;
	const int pt_num = A.f_cnt[i];
	const double curvature = A.f_curv[i];
	uint8_t is_key = 0;
	if (pt_num > P.k_min && curvature > (double)P.min_curvature)
	{
		float accu_intensity = 0.0f;
		int cnt[5] = {0, 0, 0, 0, 0}, close_cnt[5] = {0, 0, 0, 0, 0};
		const int total = pt_num > 3 ? pt_num : 0; // neighbor_indices is only filled when the PCA ran
		const unsigned long long bits = A.closebits[i];
		for (int k = 0; k < total; k++)
		{
			const uint32_t j = A.nbr[(size_t)k * n + i];
			const uint8_t lj = A.lab[j] ? A.lab[j] : A.plab[j];
			cnt[lj]++;
			close_cnt[lj] += (int)((bits >> k) & 1ull);
			accu_intensity += A.recs[(size_t)j * 3 + 2].x;
		}
		if (total > 0 && cnt[1] + cnt[2] + cnt[3] + cnt[4] >= P.min_neighbor_feature_pts) // total == 0: upstream divides by zero below
		{
			is_key = 1;
			int d0 = 0, d1 = 0, d2 = 0;
			int mul = 1000000;
#pragma unroll
			for (int l = 1; l <= 4; l++)
			{
				const int far = cnt[l] - close_cnt[l];
				d0 += (100 * cnt[l] / total) * mul;
				d1 += (100 * close_cnt[l] / total) * mul;
				d2 += (100 * far / total) * mul;
				mul /= 100;
			}
			const float4 r0 = A.recs[(size_t)i * 3], r1 = A.recs[(size_t)i * 3 + 1];
			float4 r2 = A.recs[(size_t)i * 3 + 2];
			r2.x = accu_intensity / (float)total; // mean intensity of the neighbourhood
			r2.y = (float)d0;					  // curvature <- descriptor
			A.vtx[(size_t)i * 3] = r0;
			A.vtx[(size_t)i * 3 + 1] = make_float4((float)d1, (float)d2, r1.z, (float)curvature);
			A.vtx[(size_t)i * 3 + 2] = r2;
		}
	}
	A.mask[6 * (size_t)n + i] = is_key;
}

// ---------------------------------------------------------------------------------------------------------------------
// non_max_suppress (:1243-1312) on class clouds already in visiting order: a point is kept unless a kept point of higher priority (earlier
// in the order) lies within the radius.  Two steps over the whole chip instead of upstream's walk:
//   k_cl_nms_lists   every point's earlier neighbours within the radius (all pairs once, LDS-tiled; a class cloud is <= a few 10 k points)
//   k_cl_nms_round   fixed-point rounds: an undecided point with a kept earlier neighbour is suppressed, one whose earlier neighbours are
//                    all suppressed is kept, the others wait — the first undecided point of the order never waits, so every round decides
// state: 0 suppressed, 1 kept, 2 undecided (so that the settled array is the compaction's keep mask)
#define CL_NMS_CAP MULLS_CL_NMS_CAP // list slots; a point with more earlier neighbours scans its predecessors directly in every round
__global__ __launch_bounds__(256) void k_cl_nms_lists(ClNmsArgs a)
{
	// one workgroup per pair of 256-point tiles (bi >= bj) of one class: lane t's point p of tile bi against the points q < p of tile bj
	__shared__ float sx[256], sy[256], sz[256];
	const uint32_t c = blockIdx.y, n = a.n[c], t = threadIdx.x, nb = (n + 255u) / 256u;
	const uint32_t pair = blockIdx.x;
	if (pair >= nb * (nb + 1u) / 2u)
		return;
	uint32_t bi = (uint32_t)((sqrtf(8.f * (float)pair + 1.f) - 1.f) * 0.5f);
	while (bi * (bi + 1u) / 2u > pair)
		bi--;
	while ((bi + 1u) * (bi + 2u) / 2u <= pair)
		bi++;
	const uint32_t bj = pair - bi * (bi + 1u) / 2u;
	const float4 *recs = a.recs[c];
	const uint32_t p = bi * 256u + t, q0 = bj * 256u;
	if (q0 + t < n)
	{
		const float4 r0 = recs[(size_t)(q0 + t) * 3];
		sx[t] = r0.x, sy[t] = r0.y, sz[t] = r0.z;
	}
	__syncthreads();
	if (p >= n)
		return;
	const float4 r0 = recs[(size_t)p * 3];
	const float x = r0.x, y = r0.y, z = r0.z;
	const uint32_t lim = bi == bj ? t : 256u; // q = q0 + j < p
	uint32_t *lst = a.list[c];
	for (uint32_t j = 0; j < lim; j++)
	{
		const float dx = x - sx[j], dy = y - sy[j], dz = z - sz[j];
		if (dx * dx + dy * dy + dz * dz < a.r2)
		{
			const uint32_t slot = atomicAdd(&a.cnt[c][p], 1u);
			if (slot < CL_NMS_CAP)
				lst[(size_t)slot * n + p] = q0 + j;
		}
	}
}
// the points whose earlier neighbours did not fit the fixed list (dozens of points within a quarter of the PCA radius) take cnt entries of a
// shared pool and the tile pairs are walked once more for them; only a cloud that exhausts the pool (pathological: everything within the
// radius of everything) leaves points to the direct scan of k_cl_nms_round
// before the lists: every point undecided, no earlier neighbours counted, the pool empty, the round counters zero (one launch instead of a fill command per array)
__global__ __launch_bounds__(256) void k_cl_nms_init(ClNmsArgs a, uint32_t *round_cnt)
{
	const uint32_t c = blockIdx.y, n = a.n[c], p = blockIdx.x * 256u + threadIdx.x;
	if (c == 0 && blockIdx.x == 0)
	{
		if (threadIdx.x < 64u)
			round_cnt[threadIdx.x] = 0u;
		if (threadIdx.x == 64u)
			*a.pool_used = 0ull;
	}
	if (p < n)
	{
		a.cnt[c][p] = 0u;
		a.keep[c][p] = 2;
	}
}
__global__ __launch_bounds__(256) void k_cl_nms_reserve(ClNmsArgs a)
{
	const uint32_t c = blockIdx.y, n = a.n[c], p = blockIdx.x * 256u + threadIdx.x;
	if (p >= n)
		return;
	const uint32_t cnt = a.cnt[c][p];
	uint32_t off = 0xffffffffu;
	if (cnt > CL_NMS_CAP)
	{
		const unsigned long long o = atomicAdd(a.pool_used, (unsigned long long)cnt); // 64-bit: the requests of a pathological cloud must not wrap back into the pool
		if (o <= (unsigned long long)a.pool_cap && (unsigned long long)cnt <= (unsigned long long)a.pool_cap - o)
			off = (uint32_t)o;
	}
	a.off[c][p] = off;
	a.wcur[c][p] = 0;
}
__global__ __launch_bounds__(256) void k_cl_nms_lists_long(ClNmsArgs a)
{
	__shared__ float sx[256], sy[256], sz[256];
	const uint32_t c = blockIdx.y, n = a.n[c], t = threadIdx.x, nb = (n + 255u) / 256u;
	const uint32_t pair = blockIdx.x;
	if (pair >= nb * (nb + 1u) / 2u)
		return;
	uint32_t bi = (uint32_t)((sqrtf(8.f * (float)pair + 1.f) - 1.f) * 0.5f);
	while (bi * (bi + 1u) / 2u > pair)
		bi--;
	while ((bi + 1u) * (bi + 2u) / 2u <= pair)
		bi++;
	const uint32_t bj = pair - bi * (bi + 1u) / 2u;
	const uint32_t p = bi * 256u + t, q0 = bj * 256u;
	const uint32_t off = p < n ? a.off[c][p] : 0xffffffffu;
	if (!__syncthreads_or(off != 0xffffffffu))
		return; // nobody in this tile has a long list
	const float4 *recs = a.recs[c];
	if (q0 + t < n)
	{
		const float4 r0 = recs[(size_t)(q0 + t) * 3];
		sx[t] = r0.x, sy[t] = r0.y, sz[t] = r0.z;
	}
	__syncthreads();
	if (off == 0xffffffffu)
		return;
	const float4 r0 = recs[(size_t)p * 3];
	const float x = r0.x, y = r0.y, z = r0.z;
	const uint32_t lim = bi == bj ? t : 256u;
	for (uint32_t j = 0; j < lim; j++)
	{
		const float dx = x - sx[j], dy = y - sy[j], dz = z - sz[j];
		if (dx * dx + dy * dy + dz * dz < a.r2)
			a.pool[off + atomicAdd(&a.wcur[c][p], 1u)] = q0 + j;
	}
}
__global__ __launch_bounds__(256) void k_cl_nms_round(ClNmsArgs a, uint32_t *round_cnt)
{
	const uint32_t c = blockIdx.y, n = a.n[c], p = blockIdx.x * 256u + threadIdx.x;
	if (p >= n)
		return;
	uint8_t *state = a.keep[c];
	if (state[p] != 2)
		return;
	const uint32_t cnt = a.cnt[c][p];
	const uint32_t off = cnt > CL_NMS_CAP ? a.off[c][p] : 0u;
	// one look per launch: what the other wavefronts settle meanwhile would have to be read past the L2 (one per XCD), at memory latency
	for (int look = 0; look < 1; look++)
	{
		bool kept_near = false, wait = false;
		if (cnt <= CL_NMS_CAP)
		{
			const uint32_t *lst = a.list[c];
			for (uint32_t k = 0; k < cnt; k++)
			{
				const uint8_t s = state[lst[(size_t)k * n + p]];
				kept_near |= s == 1;
				wait |= s == 2;
			}
		}
		else if (off != 0xffffffffu)
		{
			const uint32_t *lst = a.pool + off;
			for (uint32_t k = 0; k < cnt && !kept_near; k++)
			{
				const uint8_t s = state[lst[k]];
				kept_near |= s == 1;
				wait |= s == 2;
			}
		}
		else
		{
			const float4 *recs = a.recs[c];
			const float4 r0 = recs[(size_t)p * 3];
			for (uint32_t q = 0; q < p && !kept_near; q++)
			{
				const uint8_t s = state[q];
				if (s == 0)
					continue;
				const float4 rq = recs[(size_t)q * 3];
				const float dx = r0.x - rq.x, dy = r0.y - rq.y, dz = r0.z - rq.z;
				if (dx * dx + dy * dy + dz * dz < a.r2)
				{
					kept_near |= s == 1;
					wait |= s == 2;
				}
			}
		}
		if (kept_near)
		{
			state[p] = 0;
			return;
		}
		if (!wait)
		{
			state[p] = 1;
			return;
		}
		__threadfence();
	}
	atomicAdd(round_cnt, 1u);
}

__global__ __launch_bounds__(256) void k_cl_keys(const float4 *__restrict__ recs, uint32_t n, float *__restrict__ keys)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < n)
		keys[i] = recs[(size_t)i * 3 + 1].w;
}
__global__ __launch_bounds__(256) void k_cl_gather(const float4 *__restrict__ in, const uint32_t *__restrict__ perm, float4 *__restrict__ out, uint32_t n)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n)
		return;
	const uint32_t s = perm[i];
	out[(size_t)i * 3] = in[(size_t)s * 3];
	out[(size_t)i * 3 + 1] = in[(size_t)s * 3 + 1];
	out[(size_t)i * 3 + 2] = in[(size_t)s * 3 + 2];
}

// ---------------------------------------------------------------------------------------------------------------------
void launch_cl_grid(hipStream_t st, const ClArrays &A, const ClParams &P)
{
	const uint32_t n = P.n, nb = (n + 255u) / 256u;
	static const uint32_t init_keys[12] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
	(void)hipMemcpyAsync(A.grid, init_keys, 7 * sizeof(uint32_t), hipMemcpyHostToDevice, st);
	hipLaunchKernelGGL(k_cl_bbox, dim3(min(nb, 256u)), dim3(256), 0, st, A.recs, n, A.grid);
	hipLaunchKernelGGL(k_cl_setup, dim3(1), dim3(1), 0, st, A.grid, P.radius);
	(void)hipMemsetAsync(A.cell_start, 0, ((size_t)MULLS_CL_MAX_CELLS + 1u) * sizeof(uint32_t), st);
	hipLaunchKernelGGL(k_cl_count, dim3(nb), dim3(256), 0, st, A.recs, n, A.grid, A.cellof, A.cell_start);
	const uint32_t max_seg = (MULLS_CL_MAX_CELLS + 1u + CL_SEG - 1u) / CL_SEG;
	hipLaunchKernelGGL(k_cl_scan_seg, dim3(max_seg), dim3(256), 0, st, A.cell_start, A.grid, A.seg_sum);
	hipLaunchKernelGGL(k_cl_scan_top, dim3(1), dim3(1024), 0, st, A.seg_sum, A.grid);
	hipLaunchKernelGGL(k_cl_scan_apply, dim3(max_seg), dim3(256), 0, st, A.cell_start, A.grid, A.seg_sum, A.cell_fill);
	hipLaunchKernelGGL(k_cl_scatter, dim3(nb), dim3(256), 0, st, A.recs, n, A.cellof, A.cell_fill, A.sorted);
}
void launch_cl_pca(hipStream_t st, const ClArrays &A, const ClParams &P)
{
	const uint32_t nq = (P.n + (uint32_t)P.down_rate - 1u) / (uint32_t)P.down_rate;
	hipLaunchKernelGGL(k_cl_knn, dim3((nq + 3u) / 4u), dim3(256), 0, st, A, P);
	hipLaunchKernelGGL(k_cl_eig, dim3((nq + 255u) / 256u), dim3(256), 0, st, A, P);
}
void launch_cl_label(hipStream_t st, const ClArrays &A, const ClParams &P)
{
	hipLaunchKernelGGL(k_cl_label, dim3((P.n + 255u) / 256u), dim3(256), 0, st, A, P);
}
void launch_cl_promote_round(hipStream_t st, const ClArrays &A, const ClParams &P, uint32_t round)
{
	hipLaunchKernelGGL(k_cl_promote, dim3((P.n + 255u) / 256u), dim3(256), 0, st, A, P, round);
}
void launch_cl_encode_and_masks(hipStream_t st, const ClArrays &A, const ClParams &P)
{
	hipLaunchKernelGGL(k_cl_encode, dim3((P.n + 255u) / 256u), dim3(256), 0, st, A, P);
}
void launch_cl_nms_lists(hipStream_t st, const ClNmsArgs &a, uint32_t *round_cnt)
{
	uint32_t nmax = 0;
	for (int c = 0; c < 4; c++)
		nmax = max(nmax, a.n[c]);
	hipLaunchKernelGGL(k_cl_nms_init, dim3(nmax ? (nmax + 255u) / 256u : 1u, 4), dim3(256), 0, st, a, round_cnt);
	if (nmax)
	{
		const uint32_t nb = (nmax + 255u) / 256u;
		hipLaunchKernelGGL(k_cl_nms_lists, dim3(nb * (nb + 1u) / 2u, 4), dim3(256), 0, st, a);
		hipLaunchKernelGGL(k_cl_nms_reserve, dim3(nb, 4), dim3(256), 0, st, a);
		hipLaunchKernelGGL(k_cl_nms_lists_long, dim3(nb * (nb + 1u) / 2u, 4), dim3(256), 0, st, a);
	}
}
void launch_cl_nms_round(hipStream_t st, const ClNmsArgs &a, uint32_t *round_cnt)
{
	uint32_t nmax = 0;
	for (int c = 0; c < 4; c++)
		nmax = max(nmax, a.n[c]);
	if (nmax)
		hipLaunchKernelGGL(k_cl_nms_round, dim3((nmax + 255u) / 256u, 4), dim3(256), 0, st, a, round_cnt);
}
void launch_cl_keys(hipStream_t st, const float4 *recs, uint32_t n, float *keys)
{
	if (n)
		hipLaunchKernelGGL(k_cl_keys, dim3((n + 255u) / 256u), dim3(256), 0, st, recs, n, keys);
}
void launch_cl_gather(hipStream_t st, const float4 *in, const uint32_t *perm, float4 *out, uint32_t n)
{
	if (n)
		hipLaunchKernelGGL(k_cl_gather, dim3((n + 255u) / 256u), dim3(256), 0, st, in, perm, out, n);
}
