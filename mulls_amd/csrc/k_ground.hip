// k_ground.hip — CFilter::fast_ground_filter (include/common/cfilter.hpp:1658-2036) on the device, every estimate_ground_normal_method:
// the first stage of MULLS's feature extraction.  Everything the reference decides with its sequential loops is
// reproduced bit for bit, outputs in the reference's order; the order-sensitive steps are expressed as order-free ones:
//
//   k_gf_bbox      bounding box (utility.hpp:817-848): ordered-key atomicMin / atomicMax
//   k_gf_setup     approximate mean height = the SEQUENTIAL float sum of every 100th z (:1689-1698; one lane, values staged in LDS);
//                  grid geometry (:1709-1713, double arithmetic); table initialisation
//   k_gf_count     the scan in segments of 1024 points, one wave per segment, 64 points per step: cell of every point (:1730-1732),
//                  per-segment per-cell counts of the ground candidates (z <= mean + max_ground_height), per-cell minimum z and
//                  first candidate (atomicMin: order-free)
//   k_gf_prefix    per cell: candidates in the segments before each segment; prefix sum over the cells = where each cell's list starts
//   k_gf_scatter   the candidates' indices into the cell-sorted order — a STABLE counting sort (rank inside a 64-point step by a
//                  readlane loop, running bases per segment: no atomic decides an order), so every cell's list is in input order like the
//                  reference's point_id vectors; the kept high points (z above the threshold, :1742-1754) counted per segment
//   k_gf_outlier   optional per-cell outlier threshold (:1770-1790): double sums over the cell's list in its order (one wave per cell)
//   k_gf_cells1/2  per cell: the 3x3 neighbourhood minimum, reliable-neighbour count (:1795-1812) and the cell's verdict (:1834, :1853)
//   k_gf_verdict   per sorted entry: ground / non-ground / dropped (:1853-1907), its rank in the cell's list being the reference's `j`;
//                  per-block counts.  Normal method 3: every ground candidate of a ground cell becomes a member of the cell's grid_ground cloud
//   k_gf_ransac    normal method 3 (:1909-1932 -> :2038-2056 -> cprocessing.hpp:67-106): one wave per ground cell runs pcl::SACSegmentation's plane
//                  RANSAC as include/mulls_hip.h defines it (PCL's own sample sequence from a table, its float expressions, the least-squares
//                  refit through Jacobi rotations), keeps every rate-th inlier of the refined plane if abs(normal_z) > 0.8; k_gf_recount
//   k_gf_normals   normal methods 1 / 2 (:1943-1954 -> pca.hpp:66-119, :462-475), k_ground_normals.hip
//   k_gf_scan      prefix sums over the blocks and the segments' high points
//   k_gf_write     stable compaction into the output clouds; the kept high points first (input order) in the non-ground cloud
#include <algorithm>

#include "../../include/mulls_hip.h"
#include "device_util.h"
#include "ground_launch.h"
#include "pca_device.h"

#define MULLS_GF_BLOCK 256
#define MULLS_GF_SEG 1024u // points per segment (one wave walks a segment in 16 steps of 64)
#define MULLS_GF_STAGE 4096u // z samples staged per round (k_gf_setup)

struct GfState // device-resident state of one call; the head (GfOut) is read back by the host
{
	uint32_t n_ground, n_unground, n_high, error; // error: 1 = grid too large
	uint32_t row, col;
	float mean_height;
	uint32_t n_cand;
	// ---- device only
	uint32_t box[4]; // ordered keys: min x, min y, max x, max y
	float thre;		 // non_ground_height_thre
	int num_grid;
	double min_x, min_y;
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

// the cell of a point (:1730-1733): float coordinate minus double bound, divided by the float resolution, in double
__device__ __forceinline__ int gf_cell(const GfState &S, const mulls_ground_params &P, const float4 p)
{
	const int temp_col = (int)floor((p.x - S.min_x) / P.grid_resolution);
	const int temp_row = (int)floor((p.y - S.min_y) / P.grid_resolution);
	const int temp_id = temp_row * (int)S.col + temp_col;
	return (temp_id >= 0 && temp_id < S.num_grid) ? temp_id : -1;
}
// is a high point kept (:1742-1754)?  The rates come from the cell's dist2station as it stands when point j is reached: the point's
// own distance while the cell has no candidate yet (:1734-1737 overwrites it on every such point), the first candidate's afterwards.
__device__ __forceinline__ bool gf_high_keep(const mulls_ground_params &P, const float4 *__restrict__ pts, const uint32_t *__restrict__ first, uint32_t j,
											  const float4 p, float intensity, int cell)
{
	float d2s = 0.001f; // grid_t()
	if (P.distance_weight_downsampling_method > 0)
	{
		const uint32_t jc = first[cell];
		d2s = j <= jc ? gf_dist(p) : gf_dist(pts[(size_t)jc * 3]);
	}
	const GfRates r = gf_rates(P, d2s);
	return every((int)j, r.nonground) || intensity > P.intensity_thre;
}

// per-cell tables inside one arena of uint32 words, `nc` cells, `ns` segments
struct GfTables
{
	uint32_t *T;	  // [ns][nc] per-segment candidate counts -> candidates of the cell in the segments before
	uint32_t *minz;	  // [nc] ordered key of the minimum candidate z
	uint32_t *first;  // [nc] first candidate (input index)
	uint32_t *cstart; // [nc + 1] start of the cell's list in the sorted order
	float *c_minz;	  // [nc] min_z (after the outlier clamp)
	float *c_nbr;	  // [nc] neighbor_min_z
	float *c_out;	  // [nc] min_z_outlier_thre
	uint32_t *c_flag; // [nc] bit0 eligible, bit1 ground cell
};
__device__ __forceinline__ GfTables gf_tables(uint32_t *arena, uint32_t nc, uint32_t ns)
{
	GfTables t;
	t.minz = arena;
	t.first = t.minz + nc;
	t.cstart = t.first + nc;
	t.c_minz = reinterpret_cast<float *>(t.cstart + nc + 1);
	t.c_nbr = t.c_minz + nc;
	t.c_out = t.c_nbr + nc;
	t.c_flag = reinterpret_cast<uint32_t *>(t.c_out + nc);
	t.T = t.c_flag + nc;
	(void)ns;
	return t;
}
} // namespace

__global__ __launch_bounds__(MULLS_GF_BLOCK) void k_gf_bbox(const float4 *__restrict__ pts, uint32_t n, GfState *S)
{
	float mnx = __builtin_inff(), mny = __builtin_inff(), mxx = -__builtin_inff(), mxy = -__builtin_inff();
	for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
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
	__shared__ float s_red[4 * (MULLS_GF_BLOCK / 64)];
	if ((threadIdx.x & 63) == 0)
	{
		s_red[4 * (threadIdx.x >> 6) + 0] = mnx;
		s_red[4 * (threadIdx.x >> 6) + 1] = mny;
		s_red[4 * (threadIdx.x >> 6) + 2] = mxx;
		s_red[4 * (threadIdx.x >> 6) + 3] = mxy;
	}
	__syncthreads();
	if (threadIdx.x < 4)
	{
		float v = s_red[threadIdx.x];
		for (int w = 1; w < MULLS_GF_BLOCK / 64; w++)
		{
			const float o = s_red[4 * w + threadIdx.x];
			v = threadIdx.x < 2 ? (o < v ? o : v) : (o > v ? o : v);
		}
		if (threadIdx.x < 2)
			atomicMin(&S->box[threadIdx.x], f2ord(v)); // one atomic per block and side
		else
			atomicMax(&S->box[threadIdx.x], f2ord(v));
	}
}

__global__ __launch_bounds__(512) void k_gf_setup(const float4 *__restrict__ pts, uint32_t n, mulls_ground_params P, GfState *S, uint32_t max_cells)
{
	__shared__ float s_stage[MULLS_GF_STAGE];
	__shared__ float s_sum;
	const uint32_t n_samples = (n + 99u) / 100u;
	if (threadIdx.x == 0)
		s_sum = 0.001f; // float sum_height = 0.001
	__syncthreads();
	for (uint32_t s0 = 0; s0 < n_samples; s0 += MULLS_GF_STAGE)
	{
		for (uint32_t s = s0 + threadIdx.x; s < min(n_samples, s0 + MULLS_GF_STAGE); s += blockDim.x)
			s_stage[s - s0] = pts[(size_t)(s * 100u) * 3].z;
		__syncthreads();
		if (threadIdx.x == 0)
		{
			float sum = s_sum;
			const uint32_t cnt = min(n_samples, s0 + MULLS_GF_STAGE) - s0;
			uint32_t s = 0;
			for (; s + 16 <= cnt; s += 16) // 16 LDS loads in flight, then the chain of adds (the order of the additions is the reference's)
			{
				float v[16];
#pragma unroll
				for (int u = 0; u < 16; u++)
					v[u] = s_stage[s + u];
#pragma unroll
				for (int u = 0; u < 16; u++)
					sum += v[u];
			}
			for (; s < cnt; s++)
				sum += s_stage[s];
			s_sum = sum;
		}
		__syncthreads();
	}
	if (threadIdx.x == 0)
	{
		// bounds_t holds doubles (n >= 1: every side is a point's coordinate)
		const double min_x = ord2f(S->box[0]), min_y = ord2f(S->box[1]), max_x = ord2f(S->box[2]), max_y = ord2f(S->box[3]);
		const int row = (int)ceil((max_y - min_y) / P.grid_resolution);
		const int col = (int)ceil((max_x - min_x) / P.grid_resolution);
		const float appro_mean_height = s_sum / (int)n_samples; // sum_height / count_checkpoint
		const long ng = (row > 0 && col > 0) ? (long)row * (long)col : 0;
		S->min_x = min_x;
		S->min_y = min_y;
		S->row = (uint32_t)row;
		S->col = (uint32_t)col;
		S->mean_height = appro_mean_height;
		S->thre = appro_mean_height + P.max_ground_height; // non_ground_height_thre
		S->error = ng > (long)max_cells ? 1u : 0u;
		S->num_grid = S->error ? 0 : (int)ng;
		S->n_ground = S->n_unground = S->n_high = S->n_cand = 0u;
	}
}

__global__ __launch_bounds__(MULLS_GF_BLOCK) void k_gf_init(const GfState *S, uint32_t *arena, uint32_t ns)
{
	const uint32_t nc = (uint32_t)S->num_grid;
	const GfTables t = gf_tables(arena, nc, ns);
	const size_t total = (size_t)nc * ns;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
		t.T[i] = 0u;
	for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x)
	{
		t.minz[c] = f2ord(3.402823466e+38f); // FLT_MAX
		t.first[c] = 0xffffffffu;
	}
}

// one wave per segment; mode 0: counts (k_gf_count), mode 1: stable scatter + kept high points per segment (k_gf_scatter)
template <int MODE>
__global__ __launch_bounds__(MULLS_GF_BLOCK) void k_gf_walk(const float4 *__restrict__ pts, uint32_t n, mulls_ground_params P, const GfState *S, uint32_t *arena,
															 uint32_t ns, uint32_t *__restrict__ ids, uint16_t *__restrict__ cellof, uint32_t *__restrict__ seg_high)
{
	const uint32_t nc = (uint32_t)S->num_grid;
	if (!nc)
	{
		if (MODE == 1 && threadIdx.x == 0 && blockIdx.x == 0)
			for (uint32_t s = 0; s < ns; s++)
				seg_high[s] = 0u;
		return;
	}
	const GfTables t = gf_tables(arena, nc, ns);
	const int lane = threadIdx.x & 63;
	const uint32_t seg = blockIdx.x * (MULLS_GF_BLOCK / 64) + (threadIdx.x >> 6);
	if (seg >= ns)
		return;
	uint32_t *Tseg = t.T + (size_t)seg * nc;
	const float thre = S->thre;
	const uint32_t j_begin = seg * MULLS_GF_SEG, j_end = min(n, j_begin + MULLS_GF_SEG);
	uint32_t high_kept = 0;
	for (uint32_t j0 = j_begin; j0 < j_end; j0 += 64)
	{
		const uint32_t j = j0 + lane;
		int cell = -1;
		bool cand = false;
		float pz = 0.0f;
		if (j < j_end)
		{
			const float4 p = pts[(size_t)j * 3];
			cell = gf_cell(*S, P, p);
			if (cell >= 0)
			{
				if (p.z > thre)
				{
					if (MODE == 1)
						high_kept += gf_high_keep(P, pts, t.first, j, p, pts[(size_t)j * 3 + 2].x, cell) ? 1u : 0u;
				}
				else
					cand = p.z > -3.402823466e+38f; // else-branch of :1736: `z > underground_noise_thre`
			}
			pz = p.z;
		}
		// lanes of this step in the same cell: rank by lane order; the last one advances this segment's entry (one writer per cell)
		// and, in the counting pass, carries the group's minimum z and first point to the cell's tables (one atomic pair per group)
		uint32_t group = 0, rank = 0;
		bool last = cand;
		float gmin = pz;
		int gfirst = lane;
#pragma unroll 8
		for (int l = 0; l < 64; l++)
		{
			const int c = __builtin_amdgcn_readlane(cand ? cell : -2, l);
			const float zl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pz), l));
			if (cand && c == cell)
			{
				group++;
				gmin = zl < gmin ? zl : gmin;
				gfirst = l < gfirst ? l : gfirst;
				if (l < lane)
					rank++;
				if (l > lane)
					last = false;
			}
		}
		if (MODE == 0 && cand && last)
		{
			atomicMin(&t.minz[cell], f2ord(gmin));
			atomicMin(&t.first[cell], j0 + (uint32_t)gfirst);
		}
		if (MODE == 1 && cand)
		{
			const uint32_t pos = t.cstart[cell] + Tseg[cell] + rank;
			ids[pos] = j;
			cellof[pos] = (uint16_t)cell;
		}
		__builtin_amdgcn_wave_barrier(); // every lane of the group has read the running base before the last one advances it
		if (cand && last)
			Tseg[cell] += group;
		__builtin_amdgcn_wave_barrier();
	}
	if (MODE == 1)
	{
		for (int off = 32; off > 0; off >>= 1)
			high_kept += __shfl_down(high_kept, off);
		if (lane == 0)
			seg_high[seg] = high_kept;
	}
}

// per cell: counts of the segments -> candidates in the segments before; cell starts.  One workgroup of 1024 lanes.
__global__ __launch_bounds__(1024) void k_gf_prefix(GfState *S, uint32_t *arena, uint32_t ns)
{
	__shared__ uint32_t s_scan[16];
	const uint32_t nc = (uint32_t)S->num_grid;
	const GfTables t = gf_tables(arena, nc, ns);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	// lanes take cells round-robin for the column sums (coalesced rows of T), then contiguous runs for the prefix over the cells
	for (uint32_t c = threadIdx.x; c < nc; c += 1024)
	{
		uint32_t run = 0;
		for (uint32_t s = 0; s < ns; s++)
		{
			const uint32_t v = t.T[(size_t)s * nc + c];
			t.T[(size_t)s * nc + c] = run;
			run += v;
		}
		t.cstart[c] = run; // the cell's total, for now
	}
	__threadfence();
	__syncthreads();
	const uint32_t per_lane = (nc + 1023u) / 1024u;
	const uint32_t c0 = min(nc, threadIdx.x * per_lane), c1 = min(nc, c0 + per_lane);
	uint32_t mine = 0;
	for (uint32_t c = c0; c < c1; c++)
		mine += t.cstart[c];
	uint32_t incl = mine;
	for (int off = 1; off < 64; off <<= 1)
	{
		const uint32_t o = __shfl_up(incl, off);
		if (lane >= off)
			incl += o;
	}
	if (lane == 63)
		s_scan[wave] = incl;
	__syncthreads();
	uint32_t base = 0, total = 0;
	for (int w = 0; w < 16; w++)
	{
		if (w < wave)
			base += s_scan[w];
		total += s_scan[w];
	}
	base += incl - mine;
	for (uint32_t c = c0; c < c1; c++)
	{
		const uint32_t cnt = t.cstart[c];
		t.cstart[c] = base;
		base += cnt;
	}
	if (threadIdx.x == 0)
	{
		t.cstart[nc] = total;
		S->n_cand = total;
	}
}

// per cell: min_z; no outlier threshold yet
__global__ __launch_bounds__(MULLS_GF_BLOCK) void k_gf_cells1(const GfState *S, uint32_t *arena, uint32_t ns)
{
	const uint32_t nc = (uint32_t)S->num_grid;
	const GfTables t = gf_tables(arena, nc, ns);
	const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= nc)
		return;
	t.c_minz[c] = ord2f(t.minz[c]); // FLT_MAX for a cell without candidates
	t.c_out[c] = -3.402823466e+38f;
}

// the optional grid-wise outlier threshold (:1770-1790): mean and standard deviation of the cell's z in double, summed in the list's
// order.  One wave per cell: 64 values are gathered at once, lane 0 adds them one after the other (readlane) — the reference's order.
__global__ __launch_bounds__(MULLS_GF_BLOCK) void k_gf_outlier(const float4 *__restrict__ pts, mulls_ground_params P, const GfState *S, uint32_t *arena, uint32_t ns,
																 const uint32_t *__restrict__ ids)
{
	const uint32_t nc = (uint32_t)S->num_grid;
	const GfTables t = gf_tables(arena, nc, ns);
	const int lane = threadIdx.x & 63;
	const uint32_t c = blockIdx.x * (MULLS_GF_BLOCK / 64) + (threadIdx.x >> 6);
	if (c >= nc)
		return;
	const uint32_t c_begin = t.cstart[c], cnt = t.cstart[c + 1] - c_begin; // pts_count
	if ((int)cnt < P.min_grid_pt_num)
		return;
	auto rl = [](double v, int l) {
		const unsigned long long b = (unsigned long long)__double_as_longlong(v);
		const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), l);
		return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
	};
	double sum_z = 0, sum_z2 = 0;
	for (uint32_t k0 = 0; k0 < cnt; k0 += 64)
	{
		const uint32_t k = k0 + lane;
		const double z = k < cnt ? (double)pts[(size_t)ids[c_begin + k] * 3].z : 0.0;
		const int m = (int)min(64u, cnt - k0);
		for (int l = 0; l < m; l++)
			sum_z += rl(z, l);
	}
	const double mean_z = sum_z / (int)cnt;
	for (uint32_t k0 = 0; k0 < cnt; k0 += 64)
	{
		const uint32_t k = k0 + lane;
		double term = 0.0;
		if (k < cnt)
		{
			const float z = pts[(size_t)ids[c_begin + k] * 3].z;
			term = (z - mean_z) * (z - mean_z);
		}
		const int m = (int)min(64u, cnt - k0);
		for (int l = 0; l < m; l++)
			sum_z2 += rl(term, l);
	}
	if (lane == 0)
	{
		const double std_z = std::sqrt(sum_z2 / (int)cnt);
		const float thre = (float)(mean_z - P.outlier_std_scale * std_z);
		const float mz = t.c_minz[c];
		t.c_minz[c] = (mz > thre) ? mz : thre; // max_(min_z, min_z_outlier_thre)
		t.c_out[c] = thre;
	}
}

// per cell: 3x3 neighbourhood minimum, reliable neighbours, verdict (:1795-1812, :1834, :1853)
__global__ __launch_bounds__(MULLS_GF_BLOCK) void k_gf_cells2(mulls_ground_params P, const GfState *S, uint32_t *arena, uint32_t ns)
{
	const uint32_t nc = (uint32_t)S->num_grid;
	const GfTables t = gf_tables(arena, nc, ns);
	const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
	if (m >= nc)
		return;
	const int row = (int)S->row, col = (int)S->col;
	const int temp_row = (int)m / col, temp_col = (int)m % col;
	float nbr = t.c_minz[m]; // neighbor_min_z starts as min_z (both FLT_MAX for an empty cell)
	int reliable = 0;
	if (temp_row >= 1 && temp_row <= row - 2 && temp_col >= 1 && temp_col <= col - 2)
		for (int jj = -1; jj <= 1; jj++)
			for (int kk = -1; kk <= 1; kk++)
			{
				const int o = (int)m + jj * col + kk;
				nbr = (nbr < t.c_minz[o]) ? nbr : t.c_minz[o];
				if ((int)(t.cstart[o + 1] - t.cstart[o]) > P.min_grid_pt_num - 1)
					reliable++;
			}
	t.c_nbr[m] = nbr;
	uint32_t f = 0;
	if ((int)(t.cstart[m + 1] - t.cstart[m]) >= P.min_grid_pt_num && reliable >= P.reliable_neighbor_grid_num_thre)
		f = 1u | ((t.c_minz[m] - nbr < P.neighbor_height_diff) ? 2u : 0u);
	t.c_flag[m] = f;
}

// per sorted entry: verdict (:1853-1907); per-block counts of ground / non-ground entries
__global__ __launch_bounds__(MULLS_GF_BLOCK) void k_gf_verdict(const float4 *__restrict__ pts, mulls_ground_params P, const GfState *S, uint32_t *arena, uint32_t ns,
																 const uint32_t *__restrict__ ids, const uint16_t *__restrict__ cellof, uint8_t *__restrict__ code,
																 float *__restrict__ d3v, uint32_t *__restrict__ blk_cnt)
{
	__shared__ uint32_t s_cnt[2];
	const uint32_t nc = (uint32_t)S->num_grid;
	const GfTables t = gf_tables(arena, nc, ns);
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (threadIdx.x < 2)
		s_cnt[threadIdx.x] = 0u;
	__syncthreads();
	uint8_t verdict = 0;
	if (k < S->n_cand)
	{
		const uint32_t c = cellof[k];
		const uint32_t f = t.c_flag[c];
		float d3 = 0.0f;
		if (f & 1u)
		{
			const uint32_t j = ids[k];
			const float4 p = pts[(size_t)j * 3];
			const float inten = pts[(size_t)j * 3 + 2].x;
			const int jr = (int)(k - t.cstart[c]); // the reference's j: position in the cell's point_id list
			float d2s = 0.001f;
			if (P.distance_weight_downsampling_method > 0)
				d2s = gf_dist(pts[(size_t)t.first[c] * 3]); // the cell has candidates: dist2station is its first candidate's distance
			const GfRates r = gf_rates(P, d2s);
			const float mz = t.c_minz[c], othre = t.c_out[c];
			if (f & 2u)
			{
				if (p.z > othre)
				{
					if (p.z - mz < P.max_height_difference)
					{
						if (P.estimate_ground_normal_method == 3)
							verdict = 3; // member of the cell's grid_ground cloud (:1860-1861): k_gf_ransac decides
						else if (every(jr, r.ground))
							verdict = 1;
					}
					else if (every(jr, r.nonground) || inten > P.intensity_thre)
					{
						verdict = 2;
						d3 = p.z - mz;
					}
				}
			}
			else if (p.z > othre && (every(jr, r.nonground) || inten > P.intensity_thre))
			{
				verdict = 2;
				d3 = p.z - t.c_nbr[c];
			}
		}
		code[k] = verdict;
		d3v[k] = d3;
	}
	const unsigned long long mg = __ballot(verdict == 1), mu = __ballot(verdict == 2);
	if ((threadIdx.x & 63) == 0)
	{
		atomicAdd(&s_cnt[0], (uint32_t)__popcll(mg));
		atomicAdd(&s_cnt[1], (uint32_t)__popcll(mu));
	}
	__syncthreads();
	if (threadIdx.x < 2)
		blk_cnt[2 * blockIdx.x + threadIdx.x] = s_cnt[threadIdx.x];
}

// ---------------------------------------------------------------------------------------------------------------------------------
// estimate_ground_normal_method 3: pcl::SACSegmentation (SACMODEL_PLANE, SAC_RANSAC, 20 iterations, optimised coefficients) on every ground
// cell's grid_ground cloud, as include/mulls_hip.h defines it and oracle/pcl_restated.h::plane_ransac states it — the same float / double
// operations in the same order, so the same inliers and the same normal bits.  One wave per cell; the cell's members, its shuffled index
// array and its inlier list live in global scratch (a cell of a 64-beam scan holds up to a few thousand candidates).
namespace
{
// dot of a plane with (x, y, z, w) in the order Eigen's 4-float packet reduction adds it up
__device__ __forceinline__ float plane_dot(float c0, float c1, float c2, float c3, float x, float y, float z, float w) { return (c0 * x + c2 * z) + (c1 * y + c3 * w); }
__device__ __forceinline__ float rl_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
} // namespace
__global__ __launch_bounds__(MULLS_GF_BLOCK) void k_gf_ransac(const float4 *__restrict__ pts, mulls_ground_params P, const GfState *S, uint32_t *arena, uint32_t ns,
															   const uint32_t *__restrict__ ids, uint8_t *__restrict__ code, GfRansac X)
{
	__shared__ uint32_t s_sel[MULLS_GF_BLOCK / 64][8];
	const uint32_t nc = (uint32_t)S->num_grid;
	const GfTables t = gf_tables(arena, nc, ns);
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const uint32_t c = blockIdx.x * (MULLS_GF_BLOCK / 64) + wv;
	if (c >= nc || (t.c_flag[c] & 3u) != 3u)
		return; // not a ground cell
	const uint32_t c_begin = t.cstart[c], cnt = t.cstart[c + 1] - c_begin;
	const unsigned long long below = (1ull << lane) - 1ull;
	// grid_ground: the members in the list's order
	uint32_t G = 0;
	for (uint32_t k0 = 0; k0 < cnt; k0 += 64)
	{
		const uint32_t k = k0 + lane;
		const bool member = k < cnt && code[c_begin + k] == 3;
		const unsigned long long bal = __ballot(member);
		if (member)
		{
			const uint32_t pos = G + (uint32_t)__popcll(bal & below);
			X.gg[c_begin + pos] = c_begin + k;
			X.gxyz[c_begin + pos] = pts[(size_t)ids[c_begin + k] * 3];
			X.perm[c_begin + pos] = pos;
			code[c_begin + k] = 0; // dropped unless the plane keeps it
		}
		G += (uint32_t)__popcll(bal);
	}
	if ((int)G < P.min_grid_pt_num || G < 3u) // (:1909; the members of a cell below the count vanish, as upstream; fewer than three points: no model)
		return;
	__threadfence_block();
	const float4 *gx = X.gxyz + c_begin;
	uint32_t *perm = X.perm + c_begin, *inl = X.inl + c_begin;
	const float dist_thre = (float)(0.3 * P.max_height_difference);
	const double threshold = (double)dist_thre;
	int iterations = 0, n_best = -2147483647;
	float b0 = 0, b1 = 0, b2 = 0, b3 = 0;
	bool have = false;
	double p_no_outliers = 0.0;
	const double one_over_indices = 1.0 / (double)G, log_arg = 1.0 - 0.99;
	uint32_t rpos = 0;
	for (;;)
	{
		if (have)
		{
			double pw = 1.0;
			for (int i = 0; i < iterations; i++)
				pw *= p_no_outliers;
			if (!(pw > log_arg))
				break;
		}
		else if (iterations > 0)
			break;
		// getSamples: lane 0 shuffles until isSampleGood (at most 1000 draws)
		if (lane == 0)
		{
			uint32_t good = 0, a0 = 0, a1 = 0, a2 = 0;
			for (int check = 0; check < 1000 && !good; check++)
			{
				for (uint32_t i = 0; i < 3; i++)
				{
					const uint32_t j = i + X.rnd[rpos++] % (G - i);
					const uint32_t ti = perm[i], tj = perm[j];
					perm[i] = tj, perm[j] = ti;
				}
				a0 = perm[0], a1 = perm[1], a2 = perm[2];
				const float4 p0 = gx[a0], p1 = gx[a1], p2 = gx[a2];
				const float d0 = (p1.x - p0.x) / (p2.x - p0.x), d1 = (p1.y - p0.y) / (p2.y - p0.y), d2 = (p1.z - p0.z) / (p2.z - p0.z);
				good = ((d0 != d1) || (d2 != d1)) ? 1u : 0u;
			}
			s_sel[wv][0] = a0, s_sel[wv][1] = a1, s_sel[wv][2] = a2, s_sel[wv][3] = good, s_sel[wv][4] = rpos;
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
		if (!s_sel[wv][3])
			break; // "No samples could be selected!"
		rpos = s_sel[wv][4];
		const float4 p0 = gx[s_sel[wv][0]], p1 = gx[s_sel[wv][1]], p2 = gx[s_sel[wv][2]];
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); // lane 0 rewrites s_sel in the next round
		const float ax = p1.x - p0.x, ay = p1.y - p0.y, az = p1.z - p0.z, bx = p2.x - p0.x, by = p2.y - p0.y, bz = p2.z - p0.z;
		float c0 = ay * bz - az * by, c1 = az * bx - ax * bz, c2 = ax * by - ay * bx, c3 = 0.0f;
		const float zz = (c0 * c0 + c2 * c2) + (c1 * c1 + c3 * c3);
		if (zz > 0.0f)
		{
			const float nrm = sqrtf(zz);
			c0 /= nrm, c1 /= nrm, c2 /= nrm, c3 /= nrm;
		}
		c3 = -1 * plane_dot(c0, c1, c2, c3, p0.x, p0.y, p0.z, p0.w);
		int count = 0;
		for (uint32_t k0 = 0; k0 < G; k0 += 64)
		{
			const uint32_t k = k0 + lane;
			bool in = false;
			if (k < G)
			{
				const float4 q = gx[k];
				in = (double)fabsf(plane_dot(c0, c1, c2, c3, q.x, q.y, q.z, 1.0f)) < threshold;
			}
			count += (int)__popcll(__ballot(in));
		}
		if (count > n_best)
		{
			n_best = count;
			b0 = c0, b1 = c1, b2 = c2, b3 = c3;
			have = true;
			const double w = (double)n_best * one_over_indices;
			p_no_outliers = 1.0 - w * w * w;
			p_no_outliers = p_no_outliers > 2.220446049250313e-16 ? p_no_outliers : 2.220446049250313e-16;
			p_no_outliers = p_no_outliers < 1.0 - 2.220446049250313e-16 ? p_no_outliers : 1.0 - 2.220446049250313e-16;
		}
		++iterations;
		if (iterations > 20)
			break;
	}
	if (!have)
		return; // no model: no ground points from this cell
	// the best model's inliers, in order
	uint32_t I = 0;
	for (uint32_t k0 = 0; k0 < G; k0 += 64)
	{
		const uint32_t k = k0 + lane;
		bool in = false;
		if (k < G)
		{
			const float4 q = gx[k];
			in = (double)fabsf(plane_dot(b0, b1, b2, b3, q.x, q.y, q.z, 1.0f)) < threshold;
		}
		const unsigned long long bal = __ballot(in);
		if (in)
			inl[I + (uint32_t)__popcll(bal & below)] = k;
		I += (uint32_t)__popcll(bal);
	}
	float r0 = b0, r1 = b1, r2 = b2, r3 = b3;
	if (I >= 4u)
	{
		// optimizeModelCoefficients: pcl::computeMeanAndCovarianceMatrix — nine float sums in the inliers' order (64 points gathered at a time,
		// every lane adds them one after the other), then the smallest eigenvector
		__threadfence_block();
		float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, a8 = 0;
		for (uint32_t k0 = 0; k0 < I; k0 += 64)
		{
			const uint32_t k = k0 + lane;
			float4 q = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			if (k < I)
				q = gx[inl[k]];
			const int m = (int)min(64u, I - k0);
			for (int l = 0; l < m; l++)
			{
				const float x = rl_f(q.x, l), y = rl_f(q.y, l), z = rl_f(q.z, l);
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
		const float nI = (float)I;
		a0 /= nI, a1 /= nI, a2 /= nI, a3 /= nI, a4 /= nI, a5 /= nI, a6 /= nI, a7 /= nI, a8 /= nI;
		const float v0 = a0 - a6 * a6, v1 = a1 - a6 * a7, v2 = a2 - a6 * a8, v3 = a3 - a7 * a7, v4 = a4 - a7 * a8, v5 = a5 - a8 * a8;
		mulls_pca::smallest_eigenvector(v0, v1, v2, v3, v4, v5, r0, r1, r2);
		r3 = 0.0f;
		r3 = -1 * plane_dot(r0, r1, r2, r3, a6, a7, a8, 1.0f);
	}
	// the refined plane's inliers are the cell's ground points: every rate-th of them, if the plane is flat enough (:1916-1926)
	float d2s = 0.001f;
	if (P.distance_weight_downsampling_method > 0)
		d2s = gf_dist(pts[(size_t)t.first[c] * 3]);
	const int rate = gf_rates(P, d2s).ground;
	const bool flat = (double)fabsf(r2) > 0.8;
	uint32_t J = 0;
	for (uint32_t k0 = 0; k0 < G; k0 += 64)
	{
		const uint32_t k = k0 + lane;
		bool in = false;
		if (k < G)
		{
			const float4 q = gx[k];
			in = (double)fabsf(plane_dot(r0, r1, r2, r3, q.x, q.y, q.z, 1.0f)) < threshold;
		}
		const unsigned long long bal = __ballot(in);
		if (in && flat && every((int)(J + (uint32_t)__popcll(bal & below)), rate))
			code[X.gg[c_begin + k]] = 1;
		J += (uint32_t)__popcll(bal);
	}
	if (lane == 0)
		X.cell_nrm[c] = make_float4(r0, r1, r2, 0.0f);
}
// ... and the per-block count of ground entries once more (k_gf_verdict counted none of them)
__global__ __launch_bounds__(MULLS_GF_BLOCK) void k_gf_recount(const GfState *S, const uint8_t *__restrict__ code, uint32_t *__restrict__ blk_cnt)
{
	__shared__ uint32_t s_cnt;
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (threadIdx.x == 0)
		s_cnt = 0u;
	__syncthreads();
	const unsigned long long mg = __ballot(k < S->n_cand && code[k] == 1);
	if ((threadIdx.x & 63) == 0)
		atomicAdd(&s_cnt, (uint32_t)__popcll(mg));
	__syncthreads();
	if (threadIdx.x == 0)
		blk_cnt[2 * blockIdx.x] = s_cnt;
}

// exclusive prefix sums: blk_cnt[nblk][2] and seg_high[ns] in place; totals into the state.  One workgroup of 1024 lanes.
__global__ __launch_bounds__(1024) void k_gf_scan(GfState *S, uint32_t *__restrict__ blk_cnt, uint32_t nblk, uint32_t *__restrict__ seg_high, uint32_t ns)
{
	__shared__ uint32_t s_scan[16];
	__shared__ uint32_t s_tot[3];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (int which = 0; which < 3; which++)
	{
		uint32_t *a = which < 2 ? blk_cnt + which : seg_high;
		const uint32_t stride = which < 2 ? 2u : 1u, cnt = which < 2 ? nblk : ns;
		const uint32_t per_lane = (cnt + 1023u) / 1024u;
		const uint32_t i0 = min(cnt, threadIdx.x * per_lane), i1 = min(cnt, i0 + per_lane);
		uint32_t mine = 0;
		for (uint32_t i = i0; i < i1; i++)
			mine += a[(size_t)i * stride];
		uint32_t incl = mine;
		for (int off = 1; off < 64; off <<= 1)
		{
			const uint32_t o = __shfl_up(incl, off);
			if (lane >= off)
				incl += o;
		}
		__syncthreads();
		if (lane == 63)
			s_scan[wave] = incl;
		__syncthreads();
		uint32_t base = 0, total = 0;
		for (int w = 0; w < 16; w++)
		{
			if (w < wave)
				base += s_scan[w];
			total += s_scan[w];
		}
		base += incl - mine;
		for (uint32_t i = i0; i < i1; i++)
		{
			const uint32_t v = a[(size_t)i * stride];
			a[(size_t)i * stride] = base;
			base += v;
		}
		if (threadIdx.x == 0)
			s_tot[which] = total;
	}
	__syncthreads();
	if (threadIdx.x == 0)
	{
		S->n_ground = s_tot[0];
		S->n_high = s_tot[2];
		S->n_unground = s_tot[2] + s_tot[1];
	}
}

// stable compaction of the sorted entries into the two output clouds (positions: block base + rank inside the block)
__global__ __launch_bounds__(MULLS_GF_BLOCK) void k_gf_write(const float4 *__restrict__ pts, const GfState *S, const uint32_t *__restrict__ ids,
															   const uint8_t *__restrict__ code, const float *__restrict__ d3v, const uint32_t *__restrict__ blk_base,
															   float4 *__restrict__ ground, float4 *__restrict__ unground, int normal_method,
															   const uint16_t *__restrict__ cellof, const float4 *__restrict__ cell_nrm)
{
	__shared__ uint32_t s_w[2][MULLS_GF_BLOCK / 64];
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint8_t v = k < S->n_cand ? code[k] : 0;
	const unsigned long long mg = __ballot(v == 1), mu = __ballot(v == 2);
	if (lane == 0)
	{
		s_w[0][wave] = (uint32_t)__popcll(mg);
		s_w[1][wave] = (uint32_t)__popcll(mu);
	}
	__syncthreads();
	if (!v)
		return;
	const unsigned long long below = (1ull << lane) - 1ull;
	uint32_t pos = blk_base[2 * blockIdx.x + (v == 1 ? 0 : 1)] + (uint32_t)__popcll((v == 1 ? mg : mu) & below);
	for (int w = 0; w < wave; w++)
		pos += s_w[v == 1 ? 0 : 1][w];
	const uint32_t j = ids[k];
	float4 a = pts[(size_t)j * 3], b = pts[(size_t)j * 3 + 1];
	const float4 c = pts[(size_t)j * 3 + 2];
	if (v == 1)
	{
		if (normal_method == 0)
			b.x = 0.0f, b.y = 0.0f, b.z = 1.0f; // (:1867-1871)
		else if (normal_method == 3)
		{
			const float4 nn = cell_nrm[cellof[k]]; // the cell's refined plane (:1922-1924)
			b.x = nn.x, b.y = nn.y, b.z = nn.z;
		} // 1 / 2: the input's normal for now; k_gf_normals overwrites it (:1947-1952)
		ground[(size_t)pos * 3] = a;
		ground[(size_t)pos * 3 + 1] = b;
		ground[(size_t)pos * 3 + 2] = c;
	}
	else
	{
		a.w = d3v[k]; // data[3]: height above ground
		const size_t o = (size_t)(S->n_high + pos) * 3;
		unground[o] = a;
		unground[o + 1] = b;
		unground[o + 2] = c;
	}
}

// the kept high points, in input order, at the head of the non-ground cloud: one wave per segment from its base
__global__ __launch_bounds__(MULLS_GF_BLOCK) void k_gf_write_high(const float4 *__restrict__ pts, uint32_t n, mulls_ground_params P, const GfState *S, uint32_t *arena,
																	uint32_t ns, const uint32_t *__restrict__ seg_base, float4 *__restrict__ unground)
{
	const uint32_t nc = (uint32_t)S->num_grid;
	if (!nc)
		return;
	const GfTables t = gf_tables(arena, nc, ns);
	const int lane = threadIdx.x & 63;
	const uint32_t seg = blockIdx.x * (MULLS_GF_BLOCK / 64) + (threadIdx.x >> 6);
	if (seg >= ns)
		return;
	const float thre = S->thre, appro_mean_height = S->mean_height;
	uint32_t base = seg_base[seg];
	const uint32_t j_begin = seg * MULLS_GF_SEG, j_end = min(n, j_begin + MULLS_GF_SEG);
	for (uint32_t j0 = j_begin; j0 < j_end; j0 += 64)
	{
		const uint32_t j = j0 + lane;
		bool keep = false;
		float4 p = make_float4(0.0f, 0.0f, 0.0f, 0.0f), q = p;
		if (j < j_end)
		{
			p = pts[(size_t)j * 3];
			const int cell = gf_cell(*S, P, p);
			if (cell >= 0 && p.z > thre)
			{
				q = pts[(size_t)j * 3 + 2];
				keep = gf_high_keep(P, pts, t.first, j, p, q.x, cell);
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

// ---------------------------------------------------------------------------------------------------------------
// Arena of uint32 words behind the scan-sized arrays: state | seg_high[ns] | blk_cnt[2 * nblk] | tables (8 * MAXCELLS + 1 + ns * MAXCELLS ... bounded below)
static size_t gf_cell_room(uint32_t n)
{
	// the per-segment tables take ns * nc words; nc is only known on the device: room for min(MAXCELLS, 64 M words / ns) cells
	const size_t ns = (n + MULLS_GF_SEG - 1) / MULLS_GF_SEG;
	return std::min<size_t>(MULLS_GF_MAXCELLS, ((size_t)64 << 20) / std::max<size_t>(ns, 1));
}
size_t ground_filter_aux_bytes(uint32_t n)
{
	const size_t ns = (n + MULLS_GF_SEG - 1) / MULLS_GF_SEG, nblk = (n + MULLS_GF_BLOCK - 1) / MULLS_GF_BLOCK;
	const size_t nc_room = gf_cell_room(n);
	return sizeof(GfState) + 64 + (ns + 2 * nblk + 16 + 8 * (size_t)MULLS_GF_MAXCELLS + ns * nc_room) * sizeof(uint32_t);
}

int launch_ground_filter(hipStream_t st, const float4 *pts, uint32_t n, const mulls_ground_params &P, uint32_t *ids, uint16_t *cellof, uint8_t *code, float *d3v,
						 float4 *ground, float4 *unground, void *aux, const GfRansac &R)
{
	const uint32_t ns = (n + MULLS_GF_SEG - 1) / MULLS_GF_SEG, nblk = (n + MULLS_GF_BLOCK - 1) / MULLS_GF_BLOCK;
	GfState *S = static_cast<GfState *>(aux);
	uint32_t *seg_high = reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(aux) + ((sizeof(GfState) + 63) & ~(size_t)63));
	uint32_t *blk_cnt = seg_high + ns;
	uint32_t *arena = blk_cnt + 2 * nblk + 16;
	static const uint32_t box0[4] = {0xffffffffu, 0xffffffffu, 0u, 0u};
	if (hipMemcpyAsync(&S->box[0], box0, sizeof(box0), hipMemcpyHostToDevice, st) != hipSuccess)
		return -1;
	const uint32_t seg_blocks = (ns + MULLS_GF_BLOCK / 64 - 1) / (MULLS_GF_BLOCK / 64);
	hipLaunchKernelGGL(k_gf_bbox, dim3(std::min<uint32_t>(nblk, 64u)), dim3(MULLS_GF_BLOCK), 0, st, pts, n, S);
	hipLaunchKernelGGL(k_gf_setup, dim3(1), dim3(512), 0, st, pts, n, P, S, (uint32_t)gf_cell_room(n));
	hipLaunchKernelGGL(k_gf_init, dim3(512), dim3(MULLS_GF_BLOCK), 0, st, S, arena, ns);
	hipLaunchKernelGGL(k_gf_walk<0>, dim3(seg_blocks), dim3(MULLS_GF_BLOCK), 0, st, pts, n, P, S, arena, ns, ids, cellof, seg_high);
	hipLaunchKernelGGL(k_gf_prefix, dim3(1), dim3(1024), 0, st, S, arena, ns);
	hipLaunchKernelGGL(k_gf_walk<1>, dim3(seg_blocks), dim3(MULLS_GF_BLOCK), 0, st, pts, n, P, S, arena, ns, ids, cellof, seg_high);
	hipLaunchKernelGGL(k_gf_cells1, dim3(MULLS_GF_MAXCELLS / MULLS_GF_BLOCK), dim3(MULLS_GF_BLOCK), 0, st, S, arena, ns);
	if (P.apply_grid_wise_outlier_filter)
		hipLaunchKernelGGL(k_gf_outlier, dim3(MULLS_GF_MAXCELLS / (MULLS_GF_BLOCK / 64)), dim3(MULLS_GF_BLOCK), 0, st, pts, P, S, arena, ns, ids);
	hipLaunchKernelGGL(k_gf_cells2, dim3(MULLS_GF_MAXCELLS / MULLS_GF_BLOCK), dim3(MULLS_GF_BLOCK), 0, st, P, S, arena, ns);
	hipLaunchKernelGGL(k_gf_verdict, dim3(nblk), dim3(MULLS_GF_BLOCK), 0, st, pts, P, S, arena, ns, ids, cellof, code, d3v, blk_cnt);
	if (P.estimate_ground_normal_method == 3)
	{
		hipLaunchKernelGGL(k_gf_ransac, dim3(MULLS_GF_MAXCELLS / (MULLS_GF_BLOCK / 64)), dim3(MULLS_GF_BLOCK), 0, st, pts, P, S, arena, ns, ids, code, R);
		hipLaunchKernelGGL(k_gf_recount, dim3(nblk), dim3(MULLS_GF_BLOCK), 0, st, S, code, blk_cnt);
	}
	hipLaunchKernelGGL(k_gf_scan, dim3(1), dim3(1024), 0, st, S, blk_cnt, nblk, seg_high, ns);
	hipLaunchKernelGGL(k_gf_write, dim3(nblk), dim3(MULLS_GF_BLOCK), 0, st, pts, S, ids, code, d3v, blk_cnt, ground, unground, (int)P.estimate_ground_normal_method,
					   cellof, R.cell_nrm);
	hipLaunchKernelGGL(k_gf_write_high, dim3(seg_blocks), dim3(MULLS_GF_BLOCK), 0, st, pts, n, P, S, arena, ns, seg_high, unground);
	return 0;
}

// The per-point filters ahead of the ground filter as one keep mask (both keep the order, so their sequence is the AND of their tests):
// CFilter::dist_filter(cloud, xy_dist_min, xy_dist_max) (cfilter.hpp:806-832: float range expression widened to double, double limits) and
// CFilter::scanner_filter (cfilter.hpp:914-929: the ego vehicle's ring and the underground ghost points near the scanner).
// mask[i] = 1 keeps point i; the stable compaction is map_kernels.hip's.
__global__ __launch_bounds__(256) void k_raw_mask(const float4 *__restrict__ pts, uint32_t n, RawMaskArgs a, uint8_t *__restrict__ mask)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n)
		return;
	const float4 p = pts[(size_t)i * 3];
	const float dis_square = p.x * p.x + p.y * p.y;
	bool keep = true;
	if (a.dist_on)
		keep = (double)dis_square < a.dist_max_sq && (double)dis_square > a.dist_min_sq;
	if (a.scanner_on)
	{
		bool k2 = false;
		if (dis_square > a.self_radius * a.self_radius && p.z > a.z_min_global)
			k2 = dis_square > a.ghost_radius * a.ghost_radius || p.z > a.z_min_ghost;
		keep = keep && k2;
	}
	mask[i] = keep ? 1 : 0;
}
void launch_raw_mask(hipStream_t st, const float4 *pts, uint32_t n, const RawMaskArgs &a, uint8_t *mask)
{
	if (n)
		hipLaunchKernelGGL(k_raw_mask, dim3((n + 255u) / 256u), dim3(256), 0, st, pts, n, a, mask);
}

// CFilter::voxel_downsample (cfilter.hpp:83-160), device part: pcl::getMinMax3D as ordered keys (box[0..2] min, box[3..5] max, box[6] != 0: a
// non-finite coordinate) and the voxel index of every point (:128-139: float difference times the float inverse size, floor, 64-bit index).
__global__ __launch_bounds__(256) void k_vox_bbox(const float4 *__restrict__ pts, uint32_t n, uint32_t *__restrict__ box)
{
	float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
	uint32_t bad = 0;
	for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < n; j += gridDim.x * 256u)
	{
		const float4 p = pts[(size_t)j * 3];
		const float v[3] = {p.x, p.y, p.z};
		for (int k = 0; k < 3; k++)
		{
			bad |= !(fabsf(v[k]) <= 3.402823466e+38f);
			mn[k] = v[k] < mn[k] ? v[k] : mn[k];
			mx[k] = v[k] > mx[k] ? v[k] : mx[k];
		}
	}
	for (int off = 32; off > 0; off >>= 1)
	{
		for (int k = 0; k < 3; k++)
		{
			const float a = __shfl_down(mn[k], off), b = __shfl_down(mx[k], off);
			mn[k] = a < mn[k] ? a : mn[k];
			mx[k] = b > mx[k] ? b : mx[k];
		}
		bad |= __shfl_down(bad, off);
	}
	if ((threadIdx.x & 63u) == 0)
	{
		for (int k = 0; k < 3; k++)
		{
			atomicMin(&box[k], f2ord(mn[k]));
			atomicMax(&box[3 + k], f2ord(mx[k]));
		}
		if (bad)
			atomicOr(&box[6], 1u);
	}
}
__global__ __launch_bounds__(256) void k_vox_keys(const float4 *__restrict__ pts, uint32_t n, float min_x, float min_y, float min_z, float inverse_voxel_size,
												   unsigned long long mul_vx, unsigned long long mul_vy, unsigned long long *__restrict__ keys)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n)
		return;
	const float4 p = pts[(size_t)i * 3];
	const unsigned long long vx = (unsigned long long)floorf((p.x - min_x) * inverse_voxel_size);
	const unsigned long long vy = (unsigned long long)floorf((p.y - min_y) * inverse_voxel_size);
	const unsigned long long vz = (unsigned long long)floorf((p.z - min_z) * inverse_voxel_size);
	keys[i] = vx * mul_vx + vy * mul_vy + vz;
}
int launch_vox_bbox(hipStream_t st, const float4 *pts, uint32_t n, uint32_t *box)
{
	static const uint32_t box0[7] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u};
	if (hipMemcpyAsync(box, box0, sizeof(box0), hipMemcpyHostToDevice, st) != hipSuccess)
		return -1;
	if (n)
		hipLaunchKernelGGL(k_vox_bbox, dim3(std::min<uint32_t>((n + 255u) / 256u, 128u)), dim3(256), 0, st, pts, n, box);
	return 0;
}
void launch_vox_keys(hipStream_t st, const float4 *pts, uint32_t n, const float min_p[3], float inverse_voxel_size, unsigned long long mul_vx,
					 unsigned long long mul_vy, unsigned long long *keys)
{
	if (n)
		hipLaunchKernelGGL(k_vox_keys, dim3((n + 255u) / 256u), dim3(256), 0, st, pts, n, min_p[0], min_p[1], min_p[2], inverse_voxel_size, mul_vx, mul_vy, keys);
}
