// k_grid.hip — target grids: occupancy-bitmap build of the global-memory tier, in-LDS sort build of the LDS tier
// (gfx950 / CDNA4, wave64; numerics policy and launch geometry: device_util.h)
#include "device_util.h"
#include "crop_grid.h"

// only the words a cloud's grid really uses are cleared (the arena reserves the worst case per cloud)
// (lclouds: the used class clouds on a bitmap grid, pair * MULLS_NC + class each)
__global__ __launch_bounds__(MULLS_BLOCK) void k_bm_clear(const uint32_t *__restrict__ lclouds, const GridDesc *__restrict__ grids, unsigned long long *__restrict__ bm)
{
	const GridDesc g = grids[lclouds[blockIdx.x]];
	for (uint32_t w = blockIdx.y * MULLS_BLOCK + threadIdx.x; w < g.ncell; w += gridDim.y * MULLS_BLOCK)
		bm[g.cell_off + w] = 0ull;
}

// Aggregation of the bitmap kernels' atomics.  A dense map puts hundreds of consecutive scan points into one cell (and thousands into one 64-cell word), a flat
// class cloud fifty points of every workgroup into one word — and atomics on one address are served one after the other past the L2: one atomic per point took
// 316 + 133 + 170 us for a 1 M-point map (profiles/r04_large_steps.txt).  Round 4 let the lanes of a WAVE that share an address be served by one leader (four
// distinct addresses per wave, the rest one by one); round 6 first issued every leader's atomic in one instruction instead of a chain of dependent memory
// operations per group, then moved the aggregation into an LDS hash per WORKGROUP (1024 points, 2048 slots): k_bm_mark ORs the bits of a word there, k_bm_count
// counts a cell's points there and hands out arrival numbers from ONE returning atomic per cell and workgroup (profiles/r06_experiments.txt items 12, 22, 23).
#define MULLS_BM_CH 4 // consecutive 256-point chunks per workgroup of the per-point kernels (k_bm_count explains)
#define MULLS_BM_HASH_BITS 11
#define MULLS_BM_HASH (1u << MULLS_BM_HASH_BITS) // k_bm_mark's LDS table: twice the points of a workgroup
__global__ __launch_bounds__(MULLS_BLOCK) void k_bm_mark(const Job *__restrict__ tjobs, uint32_t ntjobs, const CloudDesc *__restrict__ descs,
														  const GridDesc *__restrict__ grids, const float4 *__restrict__ tpos,
														  unsigned long long *__restrict__ bm)
{
	Job job[MULLS_BM_CH];
	uint32_t ci[MULLS_BM_CH], toff[MULLS_BM_CH], tn[MULLS_BM_CH];
	bool in[MULLS_BM_CH];
#pragma unroll
	for (int u = 0; u < MULLS_BM_CH; u++)
		job[u] = tjobs[min(blockIdx.x * MULLS_BM_CH + (uint32_t)u, ntjobs - 1u)];
#pragma unroll
	for (int u = 0; u < MULLS_BM_CH; u++)
	{
		ci[u] = job[u].pair * MULLS_NC + job[u].cls;
		toff[u] = descs[ci[u]].tgt_off, tn[u] = descs[ci[u]].tgt_n;
	}
	GridDesc g[MULLS_BM_CH];
	float4 p[MULLS_BM_CH];
#pragma unroll
	for (int u = 0; u < MULLS_BM_CH; u++)
	{
		g[u] = grids[ci[u]];
		const uint32_t t = job[u].start + threadIdx.x;
		in[u] = blockIdx.x * MULLS_BM_CH + (uint32_t)u < ntjobs && t < tn[u];
		p[u] = tpos[toff[u] + (in[u] ? t : 0u)]; // (clamped: no control flow between the loads)
	}
	// The workgroup's bits are OR-ed in an LDS table first — an open-addressing hash of (bitmap word -> bits), 2048 slots for at most 1024 points — and every occupied
	// slot goes out as ONE global atomic.  A flat class cloud (the ground class of a 20 000-point local map: 11 500 points in ~200 words of one z-layer) put fifty
	// atomics on every word, and atomics on one address are served one after the other past the L2: 85 us for 64 such clouds (profiles/r06_experiments.txt item 22).
	__shared__ uint32_t hkey[MULLS_BM_HASH];
	__shared__ unsigned long long hval[MULLS_BM_HASH];
	for (uint32_t k = threadIdx.x; k < MULLS_BM_HASH; k += MULLS_BLOCK)
		hkey[k] = 0xffffffffu, hval[k] = 0ull;
	__syncthreads();
#pragma unroll
	for (int u = 0; u < MULLS_BM_CH; u++)
	{
		if (!in[u])
			continue;
		const uint32_t bit = bm_bit(g[u], p[u].x, p[u].y, p[u].z);
		const uint32_t widx = g[u].cell_off + (bit >> 6);
		const unsigned long long b = 1ull << (bit & 63u);
		uint32_t h = (widx * 2654435761u) >> (32 - MULLS_BM_HASH_BITS);
		for (;;)
		{
			const uint32_t old = atomicCAS(&hkey[h], 0xffffffffu, widx);
			if (old == 0xffffffffu || old == widx)
			{
				atomicOr(&hval[h], b);
				break;
			}
			h = (h + 1u) & (MULLS_BM_HASH - 1u);
		}
	}
	__syncthreads();
	for (uint32_t k = threadIdx.x; k < MULLS_BM_HASH; k += MULLS_BLOCK)
		if (hkey[k] != 0xffffffffu)
			atomicOr(&bm[hkey[k]], hval[k]); // (no result wanted: nothing waits for it)
}

__global__ __launch_bounds__(1024) void k_bm_scan(const uint32_t *__restrict__ lclouds, GridDesc *__restrict__ grids, const unsigned long long *__restrict__ bm,
												  uint32_t *__restrict__ pf)
{
	const uint32_t ci = lclouds[blockIdx.x];
	const GridDesc g = grids[ci];
	const unsigned long long *b = bm + g.cell_off;
	uint32_t *p = pf + g.cell_off;
	const uint32_t nocc = block_scan_1024(
		g.ncell, [&](uint32_t w) { return (uint32_t)__popcll(b[w]); }, [&](uint32_t w, uint32_t ex) { p[w] = ex; });
	if (threadIdx.x == 0)
		grids[ci].nocc = nocc;
}

// MULLS_BM_CH consecutive 256-point chunks per workgroup, every stage's loads of all chunks issued before the first is used: a chunk is a chain of five dependent
// round trips (job -> descriptor -> grid -> point -> bitmap word and rank prefix) around one atomic, and with one chunk per workgroup the launch was as long as
// that chain times the rounds of workgroups (30 M map points: 120 000 workgroups, 58 rounds).  The chunks may belong to different class clouds.
__global__ __launch_bounds__(MULLS_BLOCK) void k_bm_count(const Job *__restrict__ tjobs, uint32_t ntjobs, const CloudDesc *__restrict__ descs,
														   const GridDesc *__restrict__ grids, const float4 *__restrict__ tpos,
														   const unsigned long long *__restrict__ bm, const uint32_t *__restrict__ pf,
														   uint32_t *__restrict__ cnt, uint32_t *__restrict__ rk)
{
	Job job[MULLS_BM_CH];
	uint32_t ci[MULLS_BM_CH], toff[MULLS_BM_CH], tn[MULLS_BM_CH], t[MULLS_BM_CH];
	bool in[MULLS_BM_CH];
#pragma unroll
	for (int u = 0; u < MULLS_BM_CH; u++)
		job[u] = tjobs[min(blockIdx.x * MULLS_BM_CH + (uint32_t)u, ntjobs - 1u)];
#pragma unroll
	for (int u = 0; u < MULLS_BM_CH; u++)
	{
		ci[u] = job[u].pair * MULLS_NC + job[u].cls;
		toff[u] = descs[ci[u]].tgt_off, tn[u] = descs[ci[u]].tgt_n;
	}
	GridDesc g[MULLS_BM_CH];
	float4 p[MULLS_BM_CH];
#pragma unroll
	for (int u = 0; u < MULLS_BM_CH; u++)
	{
		g[u] = grids[ci[u]];
		t[u] = job[u].start + threadIdx.x;
		in[u] = blockIdx.x * MULLS_BM_CH + (uint32_t)u < ntjobs && t[u] < tn[u];
		p[u] = tpos[toff[u] + (in[u] ? t[u] : 0u)]; // (clamped: no control flow between the loads)
	}
	uint32_t bit[MULLS_BM_CH], pre[MULLS_BM_CH];
	unsigned long long w[MULLS_BM_CH];
#pragma unroll
	for (int u = 0; u < MULLS_BM_CH; u++)
	{
		bit[u] = in[u] ? bm_bit(g[u], p[u].x, p[u].y, p[u].z) : 0u;
		w[u] = bm[g[u].cell_off + (bit[u] >> 6)];
		pre[u] = pf[g[u].cell_off + (bit[u] >> 6)];
	}
	// The workgroup's points meet in an LDS hash (cell counter -> points of this workgroup in the cell): a point's arrival number is the number its cell's ONE
	// global atomic returns plus its rank inside the workgroup.  (One atomic per group of lanes of a wave that shared a cell before: consecutive points of a scan
	// line share their cell across waves, too — and atomics on one address are served one after the other.)
	__shared__ uint32_t hkey[MULLS_BM_HASH], hcnt[MULLS_BM_HASH];
	for (uint32_t k = threadIdx.x; k < MULLS_BM_HASH; k += MULLS_BLOCK)
		hkey[k] = 0xffffffffu, hcnt[k] = 0u;
	__syncthreads();
	uint32_t r[MULLS_BM_CH], slot[MULLS_BM_CH], lrank[MULLS_BM_CH];
#pragma unroll
	for (int u = 0; u < MULLS_BM_CH; u++)
	{
		r[u] = toff[u] + ci[u] + pre[u] + (uint32_t)__popcll(w[u] & ((1ull << (bit[u] & 63u)) - 1ull)); // counter of this point's cell (bm_rank)
		slot[u] = 0u, lrank[u] = 0u;
		if (in[u])
		{
			uint32_t h = (r[u] * 2654435761u) >> (32 - MULLS_BM_HASH_BITS);
			for (;;)
			{
				const uint32_t old = atomicCAS(&hkey[h], 0xffffffffu, r[u]);
				if (old == 0xffffffffu || old == r[u])
					break;
				h = (h + 1u) & (MULLS_BM_HASH - 1u);
			}
			slot[u] = h;
			lrank[u] = atomicAdd(&hcnt[h], 1u);
		}
	}
	__syncthreads();
	for (uint32_t k = threadIdx.x; k < MULLS_BM_HASH; k += MULLS_BLOCK)
		if (hkey[k] != 0xffffffffu)
			hcnt[k] = atomicAdd(&cnt[hkey[k]], hcnt[k]); // the cell's arrival numbers of this workgroup start here
	__syncthreads();
#pragma unroll
	for (int u = 0; u < MULLS_BM_CH; u++)
		if (in[u]) // k_bm_scatter reads both back: its slot is the cell's start + the arrival number — no second pass of atomics, no second ranking
			reinterpret_cast<uint2 *>(rk)[toff[u] + t[u]] = make_uint2(r[u], hcnt[slot[u]] + lrank[u]);
}

// counts -> start positions; the counters are left at zero so that k_bm_scatter can reuse them as insertion cursors
__global__ __launch_bounds__(1024) void k_bm_starts(const uint32_t *__restrict__ lclouds, const CloudDesc *__restrict__ descs, const GridDesc *__restrict__ grids,
													uint32_t *__restrict__ cnt, uint32_t *__restrict__ cs)
{
	const uint32_t ci = lclouds[blockIdx.x];
	const GridDesc g = grids[ci];
	const uint32_t off = descs[ci].tgt_off + ci;
	uint32_t *c = cnt + off, *s = cs + off;
	const uint32_t total = block_scan_1024(
		g.nocc, [&](uint32_t r) { return c[r]; },
		[&](uint32_t r, uint32_t ex) {
			s[r] = ex;
			c[r] = 0u;
		});
	if (threadIdx.x == 0)
		s[g.nocc] = total;
}

// counting-sort scatter: target positions ordered by cell, original index carried in .w (the order inside a cell is whatever the atomics give: every
// consumer breaks distance ties by original index)
__global__ __launch_bounds__(MULLS_BLOCK) void k_bm_scatter(const Job *__restrict__ tjobs, uint32_t ntjobs, const CloudDesc *__restrict__ descs,
															 const float4 *__restrict__ tpos, const uint32_t *__restrict__ rk, const uint32_t *__restrict__ cs,
															 float4 *__restrict__ tsorted)
{
	Job job[MULLS_BM_CH];
	uint32_t toff[MULLS_BM_CH], tn[MULLS_BM_CH], t[MULLS_BM_CH];
	bool in[MULLS_BM_CH];
#pragma unroll
	for (int u = 0; u < MULLS_BM_CH; u++)
		job[u] = tjobs[min(blockIdx.x * MULLS_BM_CH + (uint32_t)u, ntjobs - 1u)];
#pragma unroll
	for (int u = 0; u < MULLS_BM_CH; u++)
	{
		const uint32_t ci = job[u].pair * MULLS_NC + job[u].cls;
		toff[u] = descs[ci].tgt_off, tn[u] = descs[ci].tgt_n;
	}
	float4 p[MULLS_BM_CH];
	uint2 ra[MULLS_BM_CH];
#pragma unroll
	for (int u = 0; u < MULLS_BM_CH; u++)
	{
		t[u] = job[u].start + threadIdx.x;
		in[u] = blockIdx.x * MULLS_BM_CH + (uint32_t)u < ntjobs && t[u] < tn[u];
		p[u] = tpos[toff[u] + (in[u] ? t[u] : 0u)];
		ra[u] = reinterpret_cast<const uint2 *>(rk)[toff[u] + (in[u] ? t[u] : 0u)]; // (k_bm_count's counter index and arrival number)
	}
	uint32_t start[MULLS_BM_CH];
#pragma unroll
	for (int u = 0; u < MULLS_BM_CH; u++)
		start[u] = in[u] ? cs[ra[u].x] : 0u;
#pragma unroll
	for (int u = 0; u < MULLS_BM_CH; u++)
		if (in[u])
			tsorted[toff[u] + start[u] + ra[u].y] = make_float4(p[u].x, p[u].y, p[u].z, __int_as_float((int)t[u]));
}

// ---------------------------------------------------------------------------------------------------------------
// LDS tier grid build: one workgroup per target class cloud (<= MULLS_LDS_MAXPTS points, <= MULLS_MAXCELLS cells).  Grids of
// fewer than 16384 cells (the usual case) are counting-sorted with LDS atomics; otherwise the
// cloud is sorted by (cell id, original index) with a bitonic network in LDS — packed 32-bit keys, cell id < 2^16,
// index < 2^14 — then the cell table is filled by binary search of every cell id in the sorted keys.  Replaces the
// count / scan / scatter kernels (global atomics on every point, 1.2 ms for 3072 clouds) for clouds that fit; the
// cell-sorted order also becomes deterministic (index order inside a cell).
__global__ __launch_bounds__(MULLS_LDS_BLOCK) void k_grid_build_sort(const CloudDesc *__restrict__ descs, const GridDesc *__restrict__ grids,
																	  RunParams rp, const float4 *__restrict__ tpos,
																	  uint32_t *__restrict__ cell_start, float4 *__restrict__ tsorted)
{
	__shared__ uint32_t K[16384];
	const uint32_t cls = blockIdx.x % MULLS_NC;
	if (!rp.used[cls])
		return;
	const CloudDesc &d = descs[blockIdx.x];
	const GridDesc g = grids[blockIdx.x];
	const uint32_t n = d.tgt_n;
	if (g.ncell == 0 || d.tier != MULLS_TIER_LDS)
		return;
	if (g.ncell < 16384u)
	{
		// few enough cells for an on-chip histogram: counting sort with LDS atomics (a fraction of the bitonic network's passes).
		// The order inside a cell is whatever the atomics give; every consumer breaks distance ties by original index.
		__shared__ uint32_t wave_tot[MULLS_LDS_BLOCK / 64];
		uint32_t *cnt = K; // [ncell + 1]
		for (uint32_t c = threadIdx.x; c <= g.ncell; c += MULLS_LDS_BLOCK)
			cnt[c] = 0u;
		__syncthreads();
		for (uint32_t i = threadIdx.x; i < n; i += MULLS_LDS_BLOCK)
		{
			const float4 p = tpos[d.tgt_off + i];
			atomicAdd(&cnt[grid_cell_id(g, p.x, p.y, p.z)], 1u);
		}
		__syncthreads();
		// exclusive scan over the cells: consecutive cells per lane, wave scan, wave totals
		const uint32_t per = (g.ncell + 1u + MULLS_LDS_BLOCK - 1u) / MULLS_LDS_BLOCK;
		const uint32_t c0 = threadIdx.x * per, c1 = min(g.ncell + 1u, c0 + per);
		uint32_t sum = 0;
		for (uint32_t c = c0; c < c1; c++)
			sum += cnt[c];
		const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
		uint32_t incl = sum;
		for (int off = 1; off < 64; off <<= 1)
		{
			const uint32_t o = __shfl_up(incl, off);
			if (lane >= off)
				incl += o;
		}
		if (lane == 63)
			wave_tot[wave] = incl;
		__syncthreads();
		uint32_t before = incl - sum;
		for (int w = 0; w < wave; w++)
			before += wave_tot[w];
		for (uint32_t c = c0; c < c1; c++)
		{
			const uint32_t v = cnt[c];
			cnt[c] = before; // becomes the insertion cursor of the cell
			reinterpret_cast<uint16_t *>(cell_start)[g.cell_off + c] = (uint16_t)before; // LDS tier: 16-bit table (n <= MULLS_LDS_MAXPTS), half the bytes to stage
			before += v;
		}
		__syncthreads();
		for (uint32_t i = threadIdx.x; i < n; i += MULLS_LDS_BLOCK)
		{
			const float4 p = tpos[d.tgt_off + i];
			const uint32_t pos = atomicAdd(&cnt[grid_cell_id(g, p.x, p.y, p.z)], 1u);
			tsorted[d.tgt_off + pos] = make_float4(p.x, p.y, p.z, __int_as_float((int)i));
		}
		return;
	}
	uint32_t npow = 64;
	while (npow < n)
		npow <<= 1;
	for (uint32_t i = threadIdx.x; i < npow; i += MULLS_LDS_BLOCK)
	{
		uint32_t key = 0xffffffffu;
		if (i < n)
		{
			const float4 p = tpos[d.tgt_off + i];
			key = (grid_cell_id(g, p.x, p.y, p.z) << 14) | i;
		}
		K[i] = key;
	}
	__syncthreads();
	for (uint32_t kk = 2; kk <= npow; kk <<= 1)
		for (uint32_t j = kk >> 1; j > 0; j >>= 1)
		{
			for (uint32_t i = threadIdx.x; i < (npow >> 1); i += MULLS_LDS_BLOCK)
			{
				const uint32_t a = ((i & ~(j - 1u)) << 1) | (i & (j - 1u)), b = a | j;
				const uint32_t x = K[a], y = K[b];
				if ((x > y) == ((a & kk) == 0u))
				{
					K[a] = y;
					K[b] = x;
				}
			}
			__syncthreads();
		}
	for (uint32_t pos = threadIdx.x; pos < n; pos += MULLS_LDS_BLOCK)
	{
		const uint32_t idx = K[pos] & 16383u;
		const float4 p = tpos[d.tgt_off + idx];
		tsorted[d.tgt_off + pos] = make_float4(p.x, p.y, p.z, __int_as_float((int)idx));
	}
	// cell_start[c] = first sorted position whose cell id is >= c (c = ncell gives n)
	for (uint32_t c = threadIdx.x; c <= g.ncell; c += MULLS_LDS_BLOCK)
	{
		uint32_t lo = 0, hi = n;
		while (lo < hi)
		{
			const uint32_t mid = (lo + hi) >> 1;
			if ((K[mid] >> 14) < c)
				lo = mid + 1u;
			else
				hi = mid;
		}
		reinterpret_cast<uint16_t *>(cell_start)[g.cell_off + c] = (uint16_t)lo;
	}
}

// ---------------------------------------------------------------------------------------------------------------
// LDS tier, the setup of one target class cloud in one pass (rp.tgt_map): intersection crop + grid build WITHOUT a working copy of the cloud.
// k_crop + k_grid_build_sort read the staged cloud (32 B per point), write the cropped copy (32 B), read its positions twice more (count,
// scatter) and write the cell-sorted positions (16 B): 112 B per target point, ~30 000 target points per pair.  Here one 512-lane workgroup
// reads the staged POSITIONS once (12 of 16 B; all of a lane's MULLS_TG_TRIPS loads in flight together) and keeps them in registers through the
// crop, the cell count and the scatter; it writes the map rank in the cropped cloud -> staged index (2 B: the consumers gather the few target
// records they need — a correspondence's position and direction when it changes — from the staged cloud through it, tgt_record()) and the
// cell-sorted positions (16 B): 34 B per point.  Same cropped order (stable), grid descriptor, cell table and tsorted records (w = rank in the
// cropped cloud) as the two kernels give.  LDS: the cell counters (two 16-bit counters per word) or the sort keys — 64 KiB; registers capped at 128: two workgroups per CU
// (1.21 ms per 4096 pairs as one 1024-lane workgroup per CU, 0.88 ms like this; the phases by MULLS_DEBUG_STOP 11 - 15: DESIGN.md section 12.3).
#define MULLS_TG_LANES 512u
#define MULLS_TG_WAVES (MULLS_TG_LANES / 64u)
#define MULLS_TG_TRIPS ((MULLS_LDS_MAXPTS + MULLS_TG_LANES - 1u) / MULLS_TG_LANES)
__global__ __launch_bounds__(MULLS_TG_LANES, 4) void k_tgt_grid(CloudDesc *__restrict__ descs, const PairSetup *__restrict__ setup, const uint32_t *__restrict__ bbox,
															   const float4 *__restrict__ stage, RunParams rp, GridDesc *__restrict__ grids, uint16_t *__restrict__ tmap,
															   uint32_t *__restrict__ cell_start, float4 *__restrict__ tsorted)
{
	__shared__ uint32_t K[16384];						  // counting path: cell counters, two per word; sort path: (cell id, rank) keys
	__shared__ uint32_t wcnt[MULLS_TG_TRIPS * MULLS_TG_WAVES + 32u]; // survivors per (trip, wave), then their exclusive prefix: the stable order of the points
	__shared__ float box_red[MULLS_TG_LANES / 64][6];
	__shared__ uint32_t wave_tot[MULLS_TG_LANES / 64];
	__shared__ GridDesc g_sh;
	const uint32_t pair = blockIdx.x / MULLS_NC, cls = blockIdx.x % MULLS_NC;
	CloudDesc &d = descs[blockIdx.x];
	if (d.tier != MULLS_TIER_LDS)
		return; // a cloud of another tier (mixed batch): k_crop writes its cropped copy, the bitmap kernels build its grid
	const uint32_t n0 = d.tgt_n0, fmt = (d.stage_fmt >> 2) & 3u;
	const bool used = rp.used[cls] != 0;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	// diagnostics (MULLS_DEBUG_STOP = 11 .. 15): leave after a phase, the cloud reported empty (timing of the phases with rocprofv3)
#define TG_STOP(k)                                                     \
	if (rp.debug_stop == (k))                                          \
	{                                                                  \
		if (threadIdx.x == 0)                                          \
		{                                                              \
			const float inf3[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, ninf3[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()}; \
			grids[blockIdx.x] = make_grid(inf3, ninf3, 0u, rp, d, cls); \
			d.tgt_n = 0u;                                              \
		}                                                              \
		return;                                                        \
	}
	double lo[3], hi[3];
	if (rp.crop)
		crop_box(pair, bbox, setup, lo, hi);
	float px[MULLS_TG_TRIPS], py[MULLS_TG_TRIPS], pz[MULLS_TG_TRIPS];
	{
		// x y z of load_staged_pos's records, 12-byte loads (ten float4 results in flight would not fit the register budget of a 1024-lane workgroup),
		// all of them unconditional at clamped indices: no control flow between them, every load in flight before the first is consumed.  An empty
		// cloud reads the pair's setup record instead (any mapped address: nothing of it is kept)
		const float4 *base = n0 ? stage + (size_t)d.tgt_stage : reinterpret_cast<const float4 *>(setup + pair);
		const uint32_t last = n0 ? n0 - 1u : 0u, stride = fmt == MULLS_STAGE_AOS48 ? 3u : 1u;
#pragma unroll
		for (uint32_t t = 0; t < MULLS_TG_TRIPS; t++)
		{
			const float *q = reinterpret_cast<const float *>(base + (size_t)min(t * MULLS_TG_LANES + threadIdx.x, last) * stride);
			px[t] = q[0], py[t] = q[1], pz[t] = q[2];
		}
	}
	__builtin_amdgcn_sched_barrier(0); // every load is issued before the first one is consumed
	uint32_t keepmask = 0;
#pragma unroll
	for (uint32_t t = 0; t < MULLS_TG_TRIPS; t++)
	{
		const uint32_t i = t * MULLS_TG_LANES + threadIdx.x;
		__builtin_amdgcn_sched_barrier(0); // one trip at a time: interleaving the ten trips' arithmetic spills
		const bool keep = i < n0 && (!rp.crop || crop_keep(make_float4(px[t], py[t], pz[t], 0.0f), lo, hi));
		const unsigned long long bal = __ballot(keep);
		keepmask |= keep ? (1u << t) : 0u;
		if (lane == 0)
			wcnt[t * MULLS_TG_WAVES + (uint32_t)wave] = (uint32_t)__popcll(bal);
	}
	if (threadIdx.x < 32u)
		wcnt[MULLS_TG_TRIPS * MULLS_TG_WAVES + threadIdx.x] = 0u;
	__syncthreads();
	TG_STOP(11u)
	if (wave == 0)
	{
		// exclusive scan of the (trip, wave) counts, three per lane
		const uint32_t c0 = wcnt[3 * lane], c1 = wcnt[3 * lane + 1], c2 = wcnt[3 * lane + 2];
		const uint32_t sum = c0 + c1 + c2;
		uint32_t incl = sum;
		for (int off = 1; off < 64; off <<= 1)
		{
			const uint32_t o = __shfl_up(incl, off);
			if (lane >= off)
				incl += o;
		}
		const uint32_t before = incl - sum;
		wcnt[3 * lane] = before;
		wcnt[3 * lane + 1] = before + c0;
		wcnt[3 * lane + 2] = before + c0 + c1;
	}
	static_assert(MULLS_TG_TRIPS * MULLS_TG_WAVES + 1u <= 3u * 64u, "one wave scans the (trip, wave) counts, three per lane");
	__syncthreads();
	const uint32_t n = wcnt[MULLS_TG_TRIPS * MULLS_TG_WAVES]; // the prefix of the first padding entry = the number of survivors
	// rank of this lane's trip-t point in the cropped cloud (every lane of the wave calls it: ballot)
	auto rank_of = [&](uint32_t t, bool keep) { return wcnt[t * MULLS_TG_WAVES + (uint32_t)wave] + (uint32_t)__popcll(__ballot(keep) & ((1ull << lane) - 1ull)); };
	float bmin[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, bmax[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
	for (uint32_t t = 0; t < MULLS_TG_TRIPS; t++)
	{
		__builtin_amdgcn_sched_barrier(0);
		const bool keep = (keepmask >> t) & 1u;
		const uint32_t rank = rank_of(t, keep);
		if (keep)
		{
			if (used)
				tmap[d.tgt_off + rank] = (uint16_t)(t * MULLS_TG_LANES + threadIdx.x);
			if (fabsf(px[t]) <= 1.0e18f && fabsf(py[t]) <= 1.0e18f && fabsf(pz[t]) <= 1.0e18f) // the grid covers the finite points; others clamp into its border cells
			{
				bmin[0] = fminf(bmin[0], px[t]), bmin[1] = fminf(bmin[1], py[t]), bmin[2] = fminf(bmin[2], pz[t]);
				bmax[0] = fmaxf(bmax[0], px[t]), bmax[1] = fmaxf(bmax[1], py[t]), bmax[2] = fmaxf(bmax[2], pz[t]);
			}
		}
	}
	for (int k = 0; k < 3; k++)
		for (int off = 32; off > 0; off >>= 1)
		{
			bmin[k] = fminf(bmin[k], __shfl_down(bmin[k], off));
			bmax[k] = fmaxf(bmax[k], __shfl_down(bmax[k], off));
		}
	if (lane == 0)
		for (int k = 0; k < 3; k++)
		{
			box_red[wave][k] = bmin[k];
			box_red[wave][3 + k] = bmax[k];
		}
	__syncthreads();
	TG_STOP(12u)
	if (threadIdx.x == 0)
	{
		float lo3[3], hi3[3];
		for (int k = 0; k < 3; k++)
		{
			lo3[k] = box_red[0][k], hi3[k] = box_red[0][3 + k];
			for (int w = 1; w < MULLS_TG_LANES / 64; w++)
				lo3[k] = fminf(lo3[k], box_red[w][k]), hi3[k] = fmaxf(hi3[k], box_red[w][3 + k]);
		}
		const GridDesc gd = make_grid(lo3, hi3, n, rp, d, cls);
		grids[blockIdx.x] = gd;
		g_sh = gd;
		d.tgt_n = n;
	}
	__syncthreads();
	const GridDesc g = g_sh;
	__syncthreads();
	TG_STOP(13u)
	if (!used || g.ncell == 0)
		return;
	uint16_t *cs16 = reinterpret_cast<uint16_t *>(cell_start) + g.cell_off;
	if (g.ncell < 32768u) // (K holds 32768 16-bit counters; the driver keeps the fused setup's grids below that: run_setup)
	{
		// counting sort with LDS atomics on 16-bit counters packed in pairs (a count never exceeds n <= MULLS_LDS_MAXPTS: no carry into the neighbour)
		uint16_t *cnt16 = reinterpret_cast<uint16_t *>(K);
		for (uint32_t w = threadIdx.x; w <= (g.ncell >> 1); w += MULLS_TG_LANES)
			K[w] = 0u;
		__syncthreads();
#pragma unroll
		for (uint32_t t = 0; t < MULLS_TG_TRIPS; t++)
		{
			__builtin_amdgcn_sched_barrier(0);
			if ((keepmask >> t) & 1u)
			{
				const uint32_t c = grid_cell_id(g, px[t], py[t], pz[t]);
				atomicAdd(&K[c >> 1], (c & 1u) ? 0x10000u : 1u);
			}
		}
		__syncthreads();
		TG_STOP(14u)
		const uint32_t per = (g.ncell + 1u + MULLS_TG_LANES - 1u) / MULLS_TG_LANES;
		const uint32_t c0 = threadIdx.x * per, c1 = min(g.ncell + 1u, c0 + per);
		uint32_t sum = 0;
		for (uint32_t c = c0; c < c1; c++)
			sum += cnt16[c];
		uint32_t incl = sum;
		for (int off = 1; off < 64; off <<= 1)
		{
			const uint32_t o = __shfl_up(incl, off);
			if (lane >= off)
				incl += o;
		}
		if (lane == 63)
			wave_tot[wave] = incl;
		__syncthreads();
		uint32_t before = incl - sum;
		for (int w = 0; w < wave; w++)
			before += wave_tot[w];
		for (uint32_t c = c0; c < c1; c++)
		{
			const uint32_t v = cnt16[c];
			cnt16[c] = (uint16_t)before; // becomes the insertion cursor of the cell
			before += v;
		}
		__syncthreads();
		// the cursors are the cell table (16-bit entries): copied out as whole words, coalesced (the table slot is 32-byte aligned and padded)
		for (uint32_t w = threadIdx.x; w <= (g.ncell >> 1); w += MULLS_TG_LANES)
			reinterpret_cast<uint32_t *>(cs16)[w] = K[w];
		__syncthreads();
		TG_STOP(15u)
#pragma unroll
		for (uint32_t t = 0; t < MULLS_TG_TRIPS; t++)
		{
			__builtin_amdgcn_sched_barrier(0);
			const bool keep = (keepmask >> t) & 1u;
			const uint32_t rank = rank_of(t, keep);
			if (keep)
			{
				const uint32_t c = grid_cell_id(g, px[t], py[t], pz[t]);
				const uint32_t old = atomicAdd(&K[c >> 1], (c & 1u) ? 0x10000u : 1u);
				const uint32_t pos = (c & 1u) ? (old >> 16) : (old & 0xffffu);
				tsorted[d.tgt_off + pos] = make_float4(px[t], py[t], pz[t], __int_as_float((int)rank));
			}
		}
		return;
	}
	// many cells (small clouds only: the cell budget shrinks as the staged cloud grows): sort (cell id, rank) keys with a bitonic network in LDS
	uint32_t npow = 64;
	while (npow < n)
		npow <<= 1;
	for (uint32_t i = n + threadIdx.x; i < npow; i += MULLS_TG_LANES)
		K[i] = 0xffffffffu;
#pragma unroll
	for (uint32_t t = 0; t < MULLS_TG_TRIPS; t++)
	{
		const bool keep = (keepmask >> t) & 1u;
		const uint32_t rank = rank_of(t, keep);
		if (keep)
			K[rank] = (grid_cell_id(g, px[t], py[t], pz[t]) << 14) | rank;
	}
	__threadfence_block(); // the map entries are read back below
	__syncthreads();
	for (uint32_t kk = 2; kk <= npow; kk <<= 1)
		for (uint32_t j = kk >> 1; j > 0; j >>= 1)
		{
			for (uint32_t i = threadIdx.x; i < (npow >> 1); i += MULLS_TG_LANES)
			{
				const uint32_t a = ((i & ~(j - 1u)) << 1) | (i & (j - 1u)), b = a | j;
				const uint32_t x = K[a], y = K[b];
				if ((x > y) == ((a & kk) == 0u))
				{
					K[a] = y;
					K[b] = x;
				}
			}
			__syncthreads();
		}
	for (uint32_t pos = threadIdx.x; pos < n; pos += MULLS_TG_LANES)
	{
		const uint32_t idx = K[pos] & 16383u;
		const float4 q = load_staged_pos(stage, d.tgt_stage, fmt, tmap[d.tgt_off + idx]);
		tsorted[d.tgt_off + pos] = make_float4(q.x, q.y, q.z, __int_as_float((int)idx));
	}
	for (uint32_t c = threadIdx.x; c <= g.ncell; c += MULLS_TG_LANES)
	{
		uint32_t lo_ = 0, hi_ = n;
		while (lo_ < hi_)
		{
			const uint32_t mid = (lo_ + hi_) >> 1;
			if ((K[mid] >> 14) < c)
				lo_ = mid + 1u;
			else
				hi_ = mid;
		}
		cs16[c] = (uint16_t)lo_;
	}
}

// ---------------------------------------------------------------------------------------------------------------
// host-callable launch wrappers (the driver is plain C++ and never sees <<< >>>)
#include "launch.h"
#include <algorithm>

// LDS tier without the fused setup: one workgroup per cropped target class cloud (k_crop wrote the copies)
void launch_grid_build_sort(hipStream_t st, uint32_t npairs, const CloudDesc *descs, GridDesc *grids, const RunParams &rp, const float4 *tpos, uint32_t *cell_start,
							float4 *tsorted)
{
	if (npairs)
		hipLaunchKernelGGL(k_grid_build_sort, dim3(npairs * MULLS_NC), dim3(MULLS_LDS_BLOCK), 0, st, descs, grids, rp, tpos, cell_start, tsorted);
}

// global-memory tier: occupancy bitmap + ranks + counting sort by rank (cs holds the start positions) of the `nl` class clouds lclouds[]; tjobs = their
// 256-point chunks
void launch_bm_build(hipStream_t st, uint32_t nl, const uint32_t *lclouds, uint32_t ntjobs, const Job *tjobs, const CloudDesc *descs, GridDesc *grids, const float4 *tpos,
					 unsigned long long *bm, uint32_t *pf, uint32_t *cnt, uint32_t *cs, float4 *tsorted, uint32_t *rank)
{
	if (!nl)
		return;
	hipLaunchKernelGGL(k_bm_clear, dim3(nl, nl >= 64 ? 4 : 64), dim3(MULLS_BLOCK), 0, st, lclouds, grids, bm);
	if (ntjobs)
		hipLaunchKernelGGL(k_bm_mark, dim3((ntjobs + MULLS_BM_CH - 1) / MULLS_BM_CH), dim3(MULLS_BLOCK), 0, st, tjobs, ntjobs, descs, grids, tpos, bm);
	hipLaunchKernelGGL(k_bm_scan, dim3(nl), dim3(1024), 0, st, lclouds, grids, bm, pf);
	if (ntjobs)
		hipLaunchKernelGGL(k_bm_count, dim3((ntjobs + MULLS_BM_CH - 1) / MULLS_BM_CH), dim3(MULLS_BLOCK), 0, st, tjobs, ntjobs, descs, grids, tpos, bm, pf, cnt, rank);
	hipLaunchKernelGGL(k_bm_starts, dim3(nl), dim3(1024), 0, st, lclouds, descs, grids, cnt, cs);
	if (ntjobs)
		hipLaunchKernelGGL(k_bm_scatter, dim3((ntjobs + MULLS_BM_CH - 1) / MULLS_BM_CH), dim3(MULLS_BLOCK), 0, st, tjobs, ntjobs, descs, tpos, rank, cs, tsorted);
}

int launch_tgt_grid(hipStream_t st, uint32_t npairs, CloudDesc *descs, const PairSetup *setup, const uint32_t *bbox, const float4 *stage, const RunParams &rp,
					GridDesc *grids, uint16_t *tmap, uint32_t *cell_start, float4 *tsorted)
{
	if (npairs)
		hipLaunchKernelGGL(k_tgt_grid, dim3(npairs * MULLS_NC), dim3(MULLS_TG_LANES), 0, st, descs, setup, bbox, stage, rp, grids, tmap, cell_start, tsorted);
	return 0;
}
