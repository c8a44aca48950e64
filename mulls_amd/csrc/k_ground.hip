// k_ground.hip — CFilter::fast_ground_filter (include/common/cfilter.hpp:1658-2036, estimate_ground_normal_method 0) on the device:
// the first stage of MULLS's feature extraction.  ONE 512-lane workgroup per scan carries the scan through the
// whole filter (scans are independent: a batch of scans is a batch of workgroups); everything the reference decides with its
// sequential loops is reproduced bit for bit, outputs in the reference's order:
//
//   A  approximate mean height: the sequential float sum of every 100th z (:1689-1698)                     one lane, values staged in LDS
//   B  bounding box (utility.hpp:817-848), grid geometry (:1709-1713, double arithmetic)
//   C  one walk over the points, each wave over its own contiguous range in steps of 64: cell of every point (:1730-1732), per-wave
//      per-cell counts of the ground candidates (z <= mean + max_ground_height), per-cell minimum z and first candidate (LDS atomics)
//   C2 per-cell totals, per-wave bases, prefix sum over the cells: where each cell's list starts in the cell-sorted order
//   D  second walk: the candidates' indices scattered into that order — a STABLE counting sort, so every cell's list is in input
//      order like the reference's point_id vectors; the kept high points (z above the threshold, :1742-1754) counted per wave
//   E  per cell: optional outlier threshold (sequential double sums over the list, :1770-1790), then the 3x3 neighbourhood minimum
//      and reliable-neighbour count (:1795-1812), the cell's verdict and down-sampling rates (:1834-1852)
//   F  per sorted entry: ground / non-ground / dropped (:1853-1907) with its rank in the cell's list as the reference's `j`, then a
//      stable compaction of both kinds into the output clouds; the kept high points first in the non-ground cloud, in input order
//
// The per-cell tables live in global memory (one workgroup = one CU: they stay in its caches; 64 B per cell), up to
// MULLS_GF_MAXCELLS cells — a 64-beam scan's bounding box at the KITTI configuration's 2.5 m cells is 96 x 96.
#include "../../include/mulls_hip.h"
#include "device_util.h"

#define MULLS_GF_BLOCK 512
#define MULLS_GF_WAVES (MULLS_GF_BLOCK / 64)
#define MULLS_GF_MAXCELLS 65536u
#define MULLS_GF_STAGE 4096u // z samples of phase A staged per round

struct GfOut // one per scan, read back by the host
{
	uint32_t n_ground, n_unground, n_high, error; // error: 1 = grid too large
	uint32_t row, col;
	float mean_height;
	uint32_t pad_;
};

namespace
{
// (int)x as the reference's x86 build evaluates it: cvttss2si returns INT_MIN for NaN and for values outside the int range
__device__ __forceinline__ int f2i_x86(float v) { return (v >= -2147483648.0f && v < 2147483648.0f) ? (int)v : (int)0x80000000; }

// the reference's `j % rate == 0` with int operands (rate 0 would be a division by zero upstream: never produced, the rates are ... + 1)
__device__ __forceinline__ bool every(int j, int rate) { return rate != 0 && j % rate == 0; }

struct GfRates
{
	int ground, nonground;
};
// down-sampling rates from a cell's dist2station (:1742-1748, :1836-1850): `distance_weight` is a float variable
__device__ __forceinline__ GfRates gf_rates(const mulls_ground_params &P, float dist2station)
{
	GfRates r = {P.ground_random_down_rate, P.nonground_random_down_rate};
	const float distance_weight = (float)(1.0 * P.standard_distance / (dist2station + 0.0001));
	if (P.distance_weight_downsampling_method == 1)
	{
		r.ground = f2i_x86(distance_weight * P.ground_random_down_rate + 1);
		r.nonground = f2i_x86(distance_weight * P.nonground_random_down_rate + 1);
	}
	else if (P.distance_weight_downsampling_method == 2)
	{
		r.ground = f2i_x86(distance_weight * distance_weight * P.ground_random_down_rate + 1);
		r.nonground = f2i_x86(distance_weight * distance_weight * P.nonground_random_down_rate + 1);
	}
	return r;
}
__device__ __forceinline__ float gf_dist(const float4 p) { return std::sqrt(p.x * p.x + p.y * p.y + p.z * p.z); }

// exclusive prefix sum of one value per lane over the workgroup (512 lanes); total in *total (LDS scratch: MULLS_GF_WAVES + 1 words)
__device__ __forceinline__ uint32_t block_exscan(uint32_t v, uint32_t *scratch, uint32_t *total)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t incl = v;
	for (int off = 1; off < 64; off <<= 1)
	{
		const uint32_t o = __shfl_up(incl, off);
		if (lane >= off)
			incl += o;
	}
	__syncthreads();
	if (lane == 63)
		scratch[wave] = incl;
	__syncthreads();
	uint32_t base = 0, tot = 0;
	for (int w = 0; w < MULLS_GF_WAVES; w++)
	{
		if (w < wave)
			base += scratch[w];
		tot += scratch[w];
	}
	*total = tot;
	return base + incl - v;
}
} // namespace

// pts: the scan as 48-byte records (3 float4 per point).  ids / cellof / code / d3v: scratch of n entries each.  ground / unground:
// output records (capacity n points each).
__global__ __launch_bounds__(MULLS_GF_BLOCK) void k_ground_filter(const float4 *__restrict__ pts, uint32_t n, mulls_ground_params P, uint32_t *__restrict__ ids,
																	uint16_t *__restrict__ cellof, uint8_t *__restrict__ code, float *__restrict__ d3v,
																	float4 *__restrict__ ground, float4 *__restrict__ unground, uint32_t *tables, GfOut *__restrict__ out)
{
	__shared__ float s_stage[MULLS_GF_STAGE];
	__shared__ float s_red[6 * MULLS_GF_WAVES];
	__shared__ uint32_t s_scan[MULLS_GF_WAVES + 1], s_high[MULLS_GF_WAVES + 1];
	__shared__ float s_mean, s_thre;
	__shared__ double s_minx, s_miny;
	__shared__ int s_row, s_col, s_err;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

	// ---- A: approximate mean height ----------------------------------------------------------------------------------------------
	const uint32_t n_samples = (n + 99u) / 100u;
	if (threadIdx.x == 0)
		s_mean = 0.001f; // float sum_height = 0.001
	__syncthreads();
	for (uint32_t s0 = 0; s0 < n_samples; s0 += MULLS_GF_STAGE)
	{
		for (uint32_t s = s0 + threadIdx.x; s < min(n_samples, s0 + MULLS_GF_STAGE); s += MULLS_GF_BLOCK)
			s_stage[s - s0] = pts[(size_t)(s * 100u) * 3].z;
		__syncthreads();
		if (threadIdx.x == 0)
		{
			float sum = s_mean;
			for (uint32_t s = s0; s < min(n_samples, s0 + MULLS_GF_STAGE); s++)
				sum += s_stage[s - s0];
			s_mean = sum;
		}
		__syncthreads();
	}
	// ---- B: bounding box, grid --------------------------------------------------------------------------------------------------
	{
		float mnx = __builtin_inff(), mny = __builtin_inff(), mxx = -__builtin_inff(), mxy = -__builtin_inff();
		for (uint32_t j = threadIdx.x; j < n; j += MULLS_GF_BLOCK)
		{
			const float4 p = pts[(size_t)j * 3];
			mnx = p.x < mnx ? p.x : mnx; // `if (min_x > x) min_x = x`: NaN never taken
			mny = p.y < mny ? p.y : mny;
			mxx = p.x > mxx ? p.x : mxx;
			mxy = p.y > mxy ? p.y : mxy;
		}
		for (int off = 32; off > 0; off >>= 1)
		{
			const float a = __shfl_down(mnx, off), b = __shfl_down(mny, off), c = __shfl_down(mxx, off), d = __shfl_down(mxy, off);
			mnx = a < mnx ? a : mnx;
			mny = b < mny ? b : mny;
			mxx = c > mxx ? c : mxx;
			mxy = d > mxy ? d : mxy;
		}
		if (lane == 0)
		{
			s_red[wave * 4 + 0] = mnx;
			s_red[wave * 4 + 1] = mny;
			s_red[wave * 4 + 2] = mxx;
			s_red[wave * 4 + 3] = mxy;
		}
		__syncthreads();
		if (threadIdx.x == 0)
		{
			float a = s_red[0], b = s_red[1], c = s_red[2], d = s_red[3];
			for (int w = 1; w < MULLS_GF_WAVES; w++)
			{
				a = s_red[w * 4] < a ? s_red[w * 4] : a;
				b = s_red[w * 4 + 1] < b ? s_red[w * 4 + 1] : b;
				c = s_red[w * 4 + 2] > c ? s_red[w * 4 + 2] : c;
				d = s_red[w * 4 + 3] > d ? s_red[w * 4 + 3] : d;
			}
			// bounds_t holds doubles initialised to +-DBL_MAX: an empty side stays there (n >= 1 here)
			const double min_x = a, min_y = b, max_x = c, max_y = d;
			s_minx = min_x;
			s_miny = min_y;
			s_row = (int)ceil((max_y - min_y) / P.grid_resolution);
			s_col = (int)ceil((max_x - min_x) / P.grid_resolution);
			const float appro_mean_height = s_mean / (int)n_samples; // sum_height / count_checkpoint
			s_mean = appro_mean_height;
			s_thre = appro_mean_height + P.max_ground_height; // non_ground_height_thre
			const long ng = (long)s_row * (long)s_col;
			s_err = ng > (long)MULLS_GF_MAXCELLS ? 1 : 0;
		}
		__syncthreads();
	}
	const int row = s_row, col = s_col;
	const int num_grid = (row > 0 && col > 0) ? row * col : 0;
	const float appro_mean_height = s_mean, non_ground_height_thre = s_thre;
	const double min_x = s_minx, min_y = s_miny;
	if (s_err)
	{
		if (threadIdx.x == 0)
		{
			out->error = 1u;
			out->n_ground = out->n_unground = out->n_high = 0u;
			out->row = (uint32_t)row;
			out->col = (uint32_t)col;
			out->mean_height = appro_mean_height;
		}
		return;
	}
	// per-cell tables (global memory, 64 B per cell)
	const uint32_t nc = (uint32_t)num_grid;
	uint32_t *T = tables;								  // [MULLS_GF_WAVES][nc] per-wave counts -> running bases
	uint32_t *minz = T + (size_t)MULLS_GF_WAVES * nc;	  // [nc] ordered key of the minimum candidate z (atomicMin)
	uint32_t *first = minz + nc;						  // [nc] first candidate (input index, atomicMin)
	uint32_t *cstart = first + nc;						  // [nc + 1] start of the cell's list in the sorted order
	uint32_t *ccount = cstart + nc + 1;					  // [nc] candidates of the cell (pts_count)
	float *c_minz = reinterpret_cast<float *>(ccount + nc); // [nc] min_z (after the outlier clamp)
	float *c_nbr = c_minz + nc;							  // [nc] neighbor_min_z
	float *c_out = c_nbr + nc;							  // [nc] min_z_outlier_thre
	uint32_t *c_flag = reinterpret_cast<uint32_t *>(c_out + nc); // [nc] bit0 eligible, bit1 ground cell
	for (uint32_t c = threadIdx.x; c < nc; c += MULLS_GF_BLOCK)
	{
		for (int w = 0; w < MULLS_GF_WAVES; w++)
			T[w * nc + c] = 0;
		minz[c] = f2ord(3.402823466e+38f); // FLT_MAX
		first[c] = 0xffffffffu;
	}
	__threadfence();
	__syncthreads();

	// the cell of a point (:1730-1733): float coordinate minus double bound, divided by the float resolution, in double
	auto cell_of = [&](const float4 p) -> int {
		const int temp_col = (int)floor((p.x - min_x) / P.grid_resolution);
		const int temp_row = (int)floor((p.y - min_y) / P.grid_resolution);
		const int temp_id = temp_row * col + temp_col;
		return (temp_id >= 0 && temp_id < num_grid) ? temp_id : -1;
	};
	// each wave walks its own contiguous range of the scan in steps of 64 points
	const uint32_t per_wave = ((n + MULLS_GF_WAVES - 1) / MULLS_GF_WAVES + 63u) & ~63u;
	const uint32_t w_begin = min(n, (uint32_t)wave * per_wave), w_end = min(n, w_begin + per_wave);

	// ---- C: counts, minimum z and first candidate per cell -------------------------------------------------------------------------
	for (uint32_t j0 = w_begin; j0 < w_end; j0 += 64)
	{
		const uint32_t j = j0 + lane;
		int cell = -1;
		bool cand = false;
		if (j < w_end)
		{
			const float4 p = pts[(size_t)j * 3];
			cell = cell_of(p);
			cand = cell >= 0 && !(p.z > non_ground_height_thre) && p.z > -3.402823466e+38f; // else-branch of :1736, `z > underground_noise_thre`
			if (cand)
			{
				atomicMin(&minz[cell], f2ord(p.z));
				atomicMin(&first[cell], j);
			}
		}
		// lanes of this step in the same cell: the last one adds the group's count to this wave's table (no atomics: one writer per cell)
		uint32_t group = 0;
		bool last = cand;
#pragma unroll 8
		for (int l = 0; l < 64; l++)
		{
			const int c = __builtin_amdgcn_readlane(cand ? cell : -2, l);
			if (cand && c == cell)
			{
				group++;
				if (l > lane)
					last = false;
			}
		}
		if (cand && last)
			T[wave * nc + cell] += group;
	}
	__threadfence();
	__syncthreads();
	// ---- C2: totals, per-wave bases, cell starts ----------------------------------------------------------------------------------
	{
		// each lane owns the cells [c0, c1): their totals in a row, then the workgroup prefix
		const uint32_t per_lane = (nc + MULLS_GF_BLOCK - 1) / MULLS_GF_BLOCK;
		const uint32_t c0 = min(nc, threadIdx.x * per_lane), c1 = min(nc, c0 + per_lane);
		uint32_t mine = 0;
		for (uint32_t c = c0; c < c1; c++)
		{
			uint32_t run = 0;
			for (int w = 0; w < MULLS_GF_WAVES; w++)
			{
				const uint32_t t = T[w * nc + c];
				T[w * nc + c] = run; // candidates of this cell in the waves before w
				run += t;
			}
			ccount[c] = run;
			mine += run;
		}
		uint32_t total;
		uint32_t base = block_exscan(mine, s_scan, &total);
		for (uint32_t c = c0; c < c1; c++)
		{
			cstart[c] = base;
			base += ccount[c];
		}
		if (threadIdx.x == 0)
			cstart[nc] = total;
		__threadfence();
		__syncthreads();
	}
	const uint32_t n_cand = cstart[nc];
	// ---- D: stable scatter of the candidates; kept high points counted per wave ---------------------------------------------------
	uint32_t high_kept = 0;
	auto high_keep = [&](uint32_t j, const float4 p, const float4 q, int cell) -> bool {
		// :1742-1754 — the rates come from the cell's dist2station as it stands when point j is reached: the point's own distance while
		// the cell has no candidate yet (:1734-1737 overwrites it on every such point), the first candidate's afterwards
		float d2s = 0.001f; // grid_t()
		if (P.distance_weight_downsampling_method > 0)
		{
			const uint32_t jc = __hip_atomic_load(&first[cell], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			d2s = j <= jc ? gf_dist(p) : gf_dist(pts[(size_t)jc * 3]);
		}
		const GfRates r = gf_rates(P, d2s);
		return every((int)j, r.nonground) || q.x > P.intensity_thre;
	};
	for (uint32_t j0 = w_begin; j0 < w_end; j0 += 64)
	{
		const uint32_t j = j0 + lane;
		int cell = -1;
		bool cand = false;
		if (j < w_end)
		{
			const float4 p = pts[(size_t)j * 3];
			cell = cell_of(p);
			if (cell >= 0)
			{
				if (p.z > non_ground_height_thre)
					high_kept += high_keep(j, p, pts[(size_t)j * 3 + 2], cell) ? 1u : 0u;
				else
					cand = p.z > -3.402823466e+38f;
			}
		}
		uint32_t group = 0, rank = 0;
		bool last = cand;
#pragma unroll 8
		for (int l = 0; l < 64; l++)
		{
			const int c = __builtin_amdgcn_readlane(cand ? cell : -2, l);
			if (cand && c == cell)
			{
				group++;
				if (l < lane)
					rank++;
				if (l > lane)
					last = false;
			}
		}
		if (cand)
		{
			const uint32_t pos = cstart[cell] + T[wave * nc + cell] + rank;
			ids[pos] = j;
			cellof[pos] = (uint16_t)cell;
		}
		// every lane of the group has read the running base: now the last one advances it (the step is one wave: in order)
		__builtin_amdgcn_wave_barrier();
		if (cand && last)
			T[wave * nc + cell] += group;
		__builtin_amdgcn_wave_barrier();
	}
	for (int off = 32; off > 0; off >>= 1)
		high_kept += __shfl_down(high_kept, off);
	if (lane == 0)
		s_high[wave] = high_kept;
	__threadfence();
	__syncthreads();
	// ---- E: per cell — outlier threshold, neighbourhood, verdict ------------------------------------------------------------------
	for (uint32_t c = threadIdx.x; c < nc; c += MULLS_GF_BLOCK)
	{
		float mz = ord2f(__hip_atomic_load(&minz[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); // FLT_MAX for a cell without candidates
		float thre = -3.402823466e+38f;
		const uint32_t cnt = ccount[c];
		if (P.apply_grid_wise_outlier_filter && (int)cnt >= P.min_grid_pt_num)
		{
			double sum_z = 0, sum_z2 = 0;
			for (uint32_t k = 0; k < cnt; k++)
				sum_z += pts[(size_t)ids[cstart[c] + k] * 3].z;
			const double mean_z = sum_z / (int)cnt;
			for (uint32_t k = 0; k < cnt; k++)
			{
				const float z = pts[(size_t)ids[cstart[c] + k] * 3].z;
				sum_z2 += (z - mean_z) * (z - mean_z);
			}
			const double std_z = std::sqrt(sum_z2 / (int)cnt);
			thre = (float)(mean_z - P.outlier_std_scale * std_z);
			mz = (mz > thre) ? mz : thre; // max_(min_z, min_z_outlier_thre)
		}
		c_minz[c] = mz;
		c_out[c] = thre;
	}
	__threadfence();
	__syncthreads();
	for (uint32_t m = threadIdx.x; m < nc; m += MULLS_GF_BLOCK)
	{
		const int temp_row = (int)m / col, temp_col = (int)m % col;
		float nbr = c_minz[m]; // neighbor_min_z starts as min_z (both FLT_MAX for an empty cell)
		int reliable = 0;
		if (temp_row >= 1 && temp_row <= row - 2 && temp_col >= 1 && temp_col <= col - 2)
			for (int jj = -1; jj <= 1; jj++)
				for (int kk = -1; kk <= 1; kk++)
				{
					const int o = (int)m + jj * col + kk;
					nbr = (nbr < c_minz[o]) ? nbr : c_minz[o];
					if ((int)ccount[o] > P.min_grid_pt_num - 1)
						reliable++;
				}
		c_nbr[m] = nbr;
		uint32_t f = 0;
		if ((int)ccount[m] >= P.min_grid_pt_num && reliable >= P.reliable_neighbor_grid_num_thre)
			f = 1u | ((c_minz[m] - nbr < P.neighbor_height_diff) ? 2u : 0u);
		c_flag[m] = f;
	}
	__threadfence();
	__syncthreads();
	// ---- F: verdict per sorted entry, stable compaction into the outputs ------------------------------------------------------------
	// every lane owns a contiguous range of the sorted order (and, for the high points, of the input order)
	uint32_t n_g = 0, n_u = 0;
	{
		const uint32_t per_lane = (n_cand + MULLS_GF_BLOCK - 1) / MULLS_GF_BLOCK;
		const uint32_t k0 = min(n_cand, threadIdx.x * per_lane), k1 = min(n_cand, k0 + per_lane);
		for (uint32_t k = k0; k < k1; k++)
		{
			const uint32_t c = cellof[k];
			uint8_t verdict = 0;
			float d3 = 0.0f;
			const uint32_t f = c_flag[c];
			if (f & 1u)
			{
				const uint32_t j = ids[k];
				const float4 p = pts[(size_t)j * 3];
				const float inten = pts[(size_t)j * 3 + 2].x;
				const int jr = (int)(k - cstart[c]); // the reference's j: position in the cell's point_id list
				float d2s = 0.001f;
				if (P.distance_weight_downsampling_method > 0)
					d2s = gf_dist(pts[(size_t)__hip_atomic_load(&first[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * 3]); // the cell has candidates: dist2station is its first candidate's distance
				const GfRates r = gf_rates(P, d2s);
				if (f & 2u)
				{
					if (p.z > c_out[c])
					{
						if (p.z - c_minz[c] < P.max_height_difference)
						{
							if (every(jr, r.ground))
								verdict = 1;
						}
						else if (every(jr, r.nonground) || inten > P.intensity_thre)
						{
							verdict = 2;
							d3 = p.z - c_minz[c];
						}
					}
				}
				else if (p.z > c_out[c] && (every(jr, r.nonground) || inten > P.intensity_thre))
				{
					verdict = 2;
					d3 = p.z - c_nbr[c];
				}
			}
			code[k] = verdict;
			d3v[k] = d3;
			n_g += verdict == 1;
			n_u += verdict == 2;
		}
		uint32_t tot_g, tot_u;
		uint32_t bg = block_exscan(n_g, s_scan, &tot_g);
		uint32_t bu = block_exscan(n_u, s_scan, &tot_u);
		uint32_t n_high = 0;
		for (int w = 0; w < MULLS_GF_WAVES; w++)
			n_high += s_high[w];
		for (uint32_t k = k0; k < k1; k++)
		{
			const uint8_t v = code[k];
			if (!v)
				continue;
			const uint32_t j = ids[k];
			float4 a = pts[(size_t)j * 3], b = pts[(size_t)j * 3 + 1];
			const float4 cc = pts[(size_t)j * 3 + 2];
			if (v == 1)
			{
				b.x = 0.0f, b.y = 0.0f, b.z = 1.0f; // estimate_ground_normal_method 0 (:1867-1871)
				ground[(size_t)bg * 3] = a;
				ground[(size_t)bg * 3 + 1] = b;
				ground[(size_t)bg * 3 + 2] = cc;
				bg++;
			}
			else
			{
				a.w = d3v[k]; // data[3]: height above ground
				const size_t o = (size_t)(n_high + bu) * 3;
				unground[o] = a;
				unground[o + 1] = b;
				unground[o + 2] = cc;
				bu++;
			}
		}
		if (threadIdx.x == 0)
		{
			out->n_ground = tot_g;
			out->n_unground = n_high + tot_u;
			out->n_high = n_high;
			out->error = 0u;
			out->row = (uint32_t)row;
			out->col = (uint32_t)col;
			out->mean_height = appro_mean_height;
		}
	}
	// the kept high points, in input order, at the head of the non-ground cloud: each wave re-walks its range from its base
	{
		uint32_t base = 0;
		for (int w = 0; w < wave; w++)
			base += s_high[w];
		for (uint32_t j0 = w_begin; j0 < w_end; j0 += 64)
		{
			const uint32_t j = j0 + lane;
			bool keep = false;
			float4 p = make_float4(0.0f, 0.0f, 0.0f, 0.0f), q = p;
			if (j < w_end)
			{
				p = pts[(size_t)j * 3];
				const int cell = cell_of(p);
				if (cell >= 0 && p.z > non_ground_height_thre)
				{
					q = pts[(size_t)j * 3 + 2];
					keep = high_keep(j, p, q, cell);
				}
			}
			const unsigned long long m = __ballot(keep);
			if (keep)
			{
				const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
				p.w = (float)(p.z - (appro_mean_height - 3.0)); // data[3] = z - (appro_mean_height - 3.0), in double (:1751)
				unground[(size_t)pos * 3] = p;
				unground[(size_t)pos * 3 + 1] = pts[(size_t)j * 3 + 1];
				unground[(size_t)pos * 3 + 2] = q;
			}
			base += (uint32_t)__popcll(m);
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------
size_t ground_filter_table_bytes() { return ((size_t)(MULLS_GF_WAVES + 8) * MULLS_GF_MAXCELLS + 16) * sizeof(uint32_t); }

int launch_ground_filter(hipStream_t st, const float4 *pts, uint32_t n, const mulls_ground_params &P, uint32_t *ids, uint16_t *cellof, uint8_t *code, float *d3v,
						 float4 *ground, float4 *unground, uint32_t *tables, GfOut *out)
{
	hipLaunchKernelGGL(k_ground_filter, dim3(1), dim3(MULLS_GF_BLOCK), 0, st, pts, n, P, ids, cellof, code, d3v, ground, unground, tables, out);
	return 0;
}
