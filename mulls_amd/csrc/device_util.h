// device_util.h — device helpers shared by the kernel translation units (k_setup / k_grid / k_search / k_reduce .hip).
// Everything here is __device__ __forceinline__ in an unnamed namespace: each translation unit gets its own copy.
//
//
// Reference semantics implemented here (citations relative to the MULLS tree, include/common/):
//   k_clone_src   cloudblock_t::clone_feature + batch_transform_feature_points(initial_guess)   utility.hpp:524-550, cregistration.hpp:1183
//   k_crop        intersection_filter / bbx_filter (stable compaction of all 12 clouds)          cregistration.hpp:2894-2922, cfilter.hpp:950-981
//   k_nn          batch_transform_feature_points(TempTran) fused with the exact 1-NN search      cregistration.hpp:1260, :1740-1747
//   k_filter      duplicate rule, permanent source compaction, distance + direction rejectors    cregistration.hpp:1755-1830
//   k_accum       pt2pl / pt2li / pt2pt normal-equation terms, and the posterior residual pass   cregistration.hpp:1976-2275, :2546-2677
//   k_finish      fixed-order reduction of per-workgroup partials, per-iteration bookkeeping
//   k_step        correspondence-count test, 6x6 solve, step / convergence tests, posterior residual (icp_step.h)  cregistration.hpp:1296-1401
//
// Numerics policy: every float/double operation order follows the reference's C++ expressions; the translation unit
// is compiled with -ffp-contract=off (the reference build has no FMA: CMakeLists.txt:43, no -march), float sqrt and
// division are IEEE-correct (hipcc default), accumulators are double.  No MFMA: this is a search plus a reduction.
//
// Launch geometry: 256-thread workgroups (4 wave64) unless a kernel says otherwise (k_nn_lds: 1024 lanes per class cloud,
// k_accum: 1024 / 512 / 256 by trip length, k_step: one wave per pair).  The search / filter / accumulate kernels share one static job table (one job = 512 consecutive source
// points of one feature class of one pair; the LDS tier's class-level jobs are whole class clouds); dead source points keep
// their slot and are masked by a flag byte instead of being physically compacted, which makes the job tables iteration-
// invariant and the whole batch advance with a handful of launches per ICP iteration (search (+ rejection chain), accumulate,
// finish, step, publish; host-stepped loop: push states, ..., finish (+ pull)).

#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.h"

#pragma clang fp contract(off)

// Workgroups are handed to the 8 XCDs round-robin by blockIdx, and every XCD has its own L2.  Job tables are sorted by
// pair, so consecutive jobs gather from the same target clouds: this bijection gives XCD x the x-th eighth of the table,
// in order, instead of every eighth job — the jobs of one pair then share one L2 (affinity only, never correctness).
__device__ __forceinline__ uint32_t xcd_job(uint32_t b, uint32_t n)
{
	const uint32_t q = n >> 3, r = n & 7u, x = b & 7u;
	return x * q + min(x, r) + (b >> 3);
}

namespace
{

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t f2ord(float f)
{
	uint32_t u = __float_as_uint(f);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t k)
{
	uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
	return __uint_as_float(u);
}

// point i of a staged cloud of n points starting at float4 `off` (CloudDesc::stage_fmt): pos = x y z intensity, nrm = nx ny nz curvature
__device__ __forceinline__ void load_staged(const float4 *__restrict__ stage, uint32_t off, uint32_t fmt, uint32_t n, uint32_t i, float4 &pos, float4 &nrm)
{
	if (fmt == MULLS_STAGE_PACK32)
	{
		pos = stage[(size_t)off + i];
		nrm = stage[(size_t)off + n + i];
	}
	else if (fmt == MULLS_STAGE_PACK28)
	{
		pos = stage[(size_t)off + i];
		const float *f = reinterpret_cast<const float *>(stage + (size_t)off + n) + 3u * (size_t)i;
		nrm = make_float4(f[0], f[1], f[2], 0.0f);
	}
	else
	{
		const float4 *rec = stage + (size_t)off + (size_t)i * 3;
		const float4 a = rec[0], b = rec[1], c = rec[2]; // (x y z _) (nx ny nz _) (intensity curvature _ _)
		pos = make_float4(a.x, a.y, a.z, c.x);
		nrm = make_float4(b.x, b.y, b.z, c.y);
	}
}
// The rigid step of iteration 0 — TempTran is the identity there (cregistration.hpp:1131, :1260) — as pcl::transformPointCloudWithNormals<PointT,double>
// evaluates it: the search kernels' own expression (((T0 x + T1 y) + T2 z) + T3, double math, float store) with the identity's entries.  Not a no-op bit for
// bit (-0.0 becomes +0.0, a non-finite coordinate spreads), but idempotent on finite coordinates (non-finite input is unsupported: DESIGN.md section 2):
// the setup applies it where it writes the cropped source cloud, so that iteration 0 may go straight to the staged search (k_nn_lds `first`) — and a
// kernel that applies it again at iteration 0 changes nothing.
__device__ __forceinline__ void identity_step(float4 &p, float4 &n)
{
	const double x = p.x, y = p.y, z = p.z, nx = n.x, ny = n.y, nz = n.z;
	p.x = (float)(1.0 * x + 0.0 * y + 0.0 * z + 0.0);
	p.y = (float)(0.0 * x + 1.0 * y + 0.0 * z + 0.0);
	p.z = (float)(0.0 * x + 0.0 * y + 1.0 * z + 0.0);
	n.x = (float)(1.0 * nx + 0.0 * ny + 0.0 * nz);
	n.y = (float)(0.0 * nx + 1.0 * ny + 0.0 * nz);
	n.z = (float)(0.0 * nx + 0.0 * ny + 1.0 * nz);
}
__device__ __forceinline__ float4 load_staged_pos(const float4 *__restrict__ stage, uint32_t off, uint32_t fmt, uint32_t i)
{
	return fmt == MULLS_STAGE_AOS48 ? stage[(size_t)off + (size_t)i * 3] : stage[(size_t)off + i]; // (.w: data[3] or the intensity — callers read x y z)
}

// Record m of a cropped target class cloud (position + intensity, direction + curvature): out of the cropped working copy, or — LDS tier with
// rp.tgt_map (k_tgt_grid wrote no copy) — out of the staged cloud through the crop's map.  The same values either way.
__device__ __forceinline__ void tgt_record(const float4 *__restrict__ tgt_stage, const uint16_t *__restrict__ tgt_map, const CloudDesc &d, uint32_t m,
											const float4 *__restrict__ tpos, const float4 *__restrict__ tnrm, float4 &q, float4 &n)
{
	if (tgt_map)
		load_staged(tgt_stage, d.tgt_stage, (d.stage_fmt >> 2) & 3u, d.tgt_n0, tgt_map[d.tgt_off + m], q, n);
	else
		q = tpos[d.tgt_off + m], n = tnrm[d.tgt_off + m];
}
// ... its x y z alone (w unspecified)
__device__ __forceinline__ float4 tgt_point(const float4 *__restrict__ tgt_stage, const uint16_t *__restrict__ tgt_map, const CloudDesc &d, uint32_t m,
											 const float4 *__restrict__ tpos)
{
	return tgt_map ? load_staged_pos(tgt_stage, d.tgt_stage, (d.stage_fmt >> 2) & 3u, tgt_map[d.tgt_off + m]) : tpos[d.tgt_off + m];
}

__device__ __forceinline__ bool class_called(const RunParams &rp, const CloudDesc &d, int cls)
{
	// `if (used[c] && src.size() > 0) determine_corres(...)` (cregistration.hpp:1272-1292) combined with the
	// K_min = 3 early return inside it (:1727-1728, :1832-1833)
	return rp.used[cls] && d.alive_cur >= 3u && d.tgt_n >= 3u;
}

// wave64 + 4-wave workgroup sum of an unsigned count; result valid in every thread
__device__ __forceinline__ uint32_t block_sum_u32(uint32_t v, uint32_t *lds4)
{
	for (int off = 32; off > 0; off >>= 1)
		v += __shfl_down(v, off);
	int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	__syncthreads();
	if (lane == 0)
		lds4[wave] = v;
	__syncthreads();
	return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

} // namespace

// ---------------------------------------------------------------------------------------------------------------
// Exact fixed-radius search tier on a uniform grid.  The reference discards every match farther than 2.5*dis_thre
// (cregistration.hpp:1745), so visiting only the cells that intersect the search ball returns the same nearest
// neighbour as the kd-tree — provided no candidate inside the ball can be skipped.  That holds by construction:
// the cell coordinate is a monotone function of the float coordinate (subtract origin, scale, floor, clamp), so every
// target with |t.x - p.x| <= R lies in a cell between cell(p.x - R) and cell(p.x + R); R carries a relative 1e-4 and
// an absolute 1e-4 m margin over the float rounding of the distance arithmetic.  Ties on the float distance resolve to
// the lowest original target index explicitly (the cell-sorted order is not index order).
namespace
{
__device__ __forceinline__ int grid_cell(float v, float o, float inv_h, uint32_t n)
{
	const float c = floorf((v - o) * inv_h);
	return (int)fminf(fmaxf(c, 0.0f), (float)(n - 1u));
}
__device__ __forceinline__ uint32_t grid_cell_id(const GridDesc &g, float x, float y, float z)
{
	return ((uint32_t)grid_cell(z, g.oz, g.inv_h, g.nz) * g.ny + (uint32_t)grid_cell(y, g.oy, g.inv_h, g.ny)) * g.nx +
		   (uint32_t)grid_cell(x, g.ox, g.inv_h, g.nx);
}
// Evaluate every target in the cells intersecting the cube [p - R, p + R].  The MULLS_GRID_GROUP (= 16) lanes of a
// sub-group share one query.  Rows (fixed cy,cz; cells x0..x1 are one contiguous range of the cell-sorted array) are
// taken 16 at a time: lane j fetches the bounds of row j (occupancy words and ranks, then — only for rows that hold points —
// the two start positions: 16 rows per two memory latencies), then every lane issues its
// first candidate load of 8 rows back to back (coalesced 256-B segments), and only rows holding more than 16
// candidates loop further.  Chunks whose 16 rows are all empty cost one latency and no candidate work.
struct BmGrid
{
	const unsigned long long *bm; // [nw] occupancy words of this cloud
	const uint32_t *pf;			  // [nw] occupied cells before each word
	const uint32_t *cs;			  // [nocc + 1] first sorted position of every occupied cell
};
// sorted candidates [lo, hi) of cells xa..xb of the row whose first word is `rowword` (cells of a row are consecutive in the
// cell-sorted order, occupied or not): ranks of the first and one-past-last occupied cell, then their start positions
__device__ __forceinline__ void bm_row_range(const BmGrid &B, uint32_t rowword, uint32_t xa, uint32_t xb, uint32_t &lo, uint32_t &hi)
{
	const uint32_t wa = rowword + (xa >> 6), wb = rowword + (xb >> 6);
	const unsigned long long A = B.bm[wa], Z = B.bm[wb];
	const uint32_t r0 = B.pf[wa] + (uint32_t)__popcll(A & ((1ull << (xa & 63u)) - 1ull));
	const uint32_t r1 = B.pf[wb] + (uint32_t)__popcll(Z & (~0ull >> (63u - (xb & 63u))));
	lo = hi = 0;
	if (r1 > r0)
	{
		lo = B.cs[r0];
		hi = B.cs[r1];
	}
}
__device__ __forceinline__ void grid_scan_box(const GridDesc &g, const BmGrid &B, const float4 *__restrict__ ts,
											   float px, float py, float pz, float R, uint32_t sub, float &best, int &bi)
{
	const float Rm = R * 1.0001f + 1e-4f;
	const int x0 = grid_cell(px - Rm, g.ox, g.inv_h, g.nx), x1 = grid_cell(px + Rm, g.ox, g.inv_h, g.nx);
	const int y0 = grid_cell(py - Rm, g.oy, g.inv_h, g.ny), y1 = grid_cell(py + Rm, g.oy, g.inv_h, g.ny);
	const int z0 = grid_cell(pz - Rm, g.oz, g.inv_h, g.nz), z1 = grid_cell(pz + Rm, g.oz, g.inv_h, g.nz);
	const int nyc = y1 - y0 + 1, nrows = nyc * (z1 - z0 + 1);
	const uint32_t gshift = (threadIdx.x & 63u) & ~(MULLS_GRID_GROUP - 1u); // first lane of this sub-group inside its wave
	for (int base = 0; base < nrows; base += (int)MULLS_GRID_GROUP)
	{
		const int j = base + (int)sub;
		uint32_t lo = 0, hi = 0;
		if (j < nrows)
		{
			const uint32_t rowword = ((uint32_t)(z0 + j / nyc) * g.ny + (uint32_t)(y0 + j % nyc)) * g.wpr;
			bm_row_range(B, rowword, (uint32_t)x0, (uint32_t)x1, lo, hi);
		}
		const uint32_t nonempty = (uint32_t)(__ballot(hi > lo) >> gshift) & 0xffffu; // per sub-group: which of its 16 rows hold points
		if (!nonempty)
			continue;
#pragma unroll
		for (int half = 0; half < 2; half++) // 8 rows at a time keeps the register footprint at 8 float4 of loads in flight
		{
			if (!((nonempty >> (8 * half)) & 0xffu))
				continue;
			float4 q[8];
			bool ok[8];
#pragma unroll
			for (int jj = 0; jj < 8; jj++)
			{
				const uint32_t t = __shfl(lo, 8 * half + jj, MULLS_GRID_GROUP) + sub;
				ok[jj] = t < __shfl(hi, 8 * half + jj, MULLS_GRID_GROUP);
				if (ok[jj])
					q[jj] = ts[t];
			}
#pragma unroll
			for (int jj = 0; jj < 8; jj++)
				if (ok[jj])
				{
					const float dx = px - q[jj].x, dy = py - q[jj].y, dz = pz - q[jj].z;
					const float dist = (dx * dx + dy * dy) + dz * dz; // L2_Simple<float>, no FMA
					const int idx = __float_as_int(q[jj].w);
					if (dist < best || (dist == best && idx < bi))
					{
						best = dist;
						bi = idx;
					}
				}
		}
		// rows holding more than 16 candidates: four loads in flight per lane and trip
		for (uint32_t rem = nonempty; rem; rem &= rem - 1u)
		{
			const int jj = __ffs((int)rem) - 1;
			const uint32_t lo_j = __shfl(lo, jj, MULLS_GRID_GROUP), hi_j = __shfl(hi, jj, MULLS_GRID_GROUP);
			for (uint32_t t = lo_j + sub + MULLS_GRID_GROUP; t < hi_j; t += 4 * MULLS_GRID_GROUP)
			{
				float4 c[4];
				bool v[4];
#pragma unroll
				for (int w = 0; w < 4; w++)
				{
					v[w] = t + w * MULLS_GRID_GROUP < hi_j;
					if (v[w])
						c[w] = ts[t + w * MULLS_GRID_GROUP];
				}
#pragma unroll
				for (int w = 0; w < 4; w++)
					if (v[w])
					{
						const float dx = px - c[w].x, dy = py - c[w].y, dz = pz - c[w].z;
						const float dist = (dx * dx + dy * dy) + dz * dz;
						const int idx = __float_as_int(c[w].w);
						if (dist < best || (dist == best && idx < bi))
						{
							best = dist;
							bi = idx;
						}
					}
			}
		}
	}
}
// lexicographic (distance, original index) minimum over the lanes of one sub-group; result in every lane
__device__ __forceinline__ void group_min(float &best, int &bi)
{
#pragma unroll
	for (int mask = MULLS_GRID_GROUP / 2; mask > 0; mask >>= 1)
	{
		const float ob = __shfl_xor(best, mask, MULLS_GRID_GROUP);
		const int oi = __shfl_xor(bi, mask, MULLS_GRID_GROUP);
		const bool take = oi >= 0 && (bi < 0 || ob < best || (ob == best && oi < bi));
		best = take ? ob : best;
		bi = take ? oi : bi;
	}
}
} // namespace

// Global-memory tier grid build (clouds too large for LDS).  Fine cells make a dense cell table impossible (a 1 M-point
// map at 0.35 m cells spans ~50 M cells, of which <1 % hold points), so the grid is an occupancy bitmap + ranks:
//   bm  one bit per cell, 64 cells of a row per word          k_bm_mark   (atomicOr per point)
//   pf  per word: occupied cells before it                    k_bm_scan   (one workgroup per cloud)
//   cs  per occupied cell: first sorted position (+ end)      k_bm_count -> k_bm_starts -> k_bm_scatter (counting sort by rank)
// cs / cnt are indexed from cs_off = tgt_off + cloud index (every cloud needs tgt_n + 1 entries).
namespace
{
__device__ __forceinline__ uint32_t bm_bit(const GridDesc &g, float x, float y, float z)
{
	const uint32_t row = (uint32_t)grid_cell(z, g.oz, g.inv_h, g.nz) * g.ny + (uint32_t)grid_cell(y, g.oy, g.inv_h, g.ny);
	return row * (g.wpr * 64u) + (uint32_t)grid_cell(x, g.ox, g.inv_h, g.nx);
}
__device__ __forceinline__ uint32_t bm_rank(const unsigned long long *bm, const uint32_t *pf, uint32_t bit)
{
	return pf[bit >> 6] + (uint32_t)__popcll(bm[bit >> 6] & ((1ull << (bit & 63u)) - 1ull));
}
} // namespace

// exclusive scan of a uint32 sequence produced by `value(i)`, one 1024-lane workgroup, four items per lane and trip
template <typename F, typename G>
__device__ __forceinline__ uint32_t block_scan_1024(uint32_t n, F value, G store)
{
	__shared__ uint32_t wave_tot[16];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t running = 0;
	for (uint32_t base = 0; base < n; base += 4096u)
	{
		const uint32_t i0 = base + threadIdx.x * 4u;
		uint32_t v[4], sum = 0;
		for (int k = 0; k < 4; k++)
		{
			v[k] = (i0 + k < n) ? value(i0 + k) : 0u;
			sum += v[k];
		}
		uint32_t incl = sum;
		for (int off = 1; off < 64; off <<= 1)
		{
			const uint32_t o = __shfl_up(incl, off);
			if (lane >= off)
				incl += o;
		}
		__syncthreads();
		if (lane == 63)
			wave_tot[wave] = incl;
		__syncthreads();
		uint32_t wbase = 0, total = 0;
		for (int w = 0; w < 16; w++)
		{
			if (w < wave)
				wbase += wave_tot[w];
			total += wave_tot[w];
		}
		uint32_t ex = running + wbase + incl - sum;
		for (int k = 0; k < 4; k++)
			if (i0 + k < n)
			{
				store(i0 + k, ex);
				ex += v[k];
			}
		running += total;
	}
	return running;
}
