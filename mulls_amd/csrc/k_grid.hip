// k_grid.hip — target grids: occupancy-bitmap build of the global-memory tier, in-LDS sort build of the LDS tier
// (gfx950 / CDNA4, wave64; numerics policy and launch geometry: device_util.h)
#include "device_util.h"

// only the words a cloud's grid really uses are cleared (the arena reserves the worst case per cloud)
__global__ __launch_bounds__(MULLS_BLOCK) void k_bm_clear(const GridDesc *__restrict__ grids, RunParams rp, unsigned long long *__restrict__ bm)
{
	if (!rp.used[blockIdx.x % MULLS_NC])
		return;
	const GridDesc g = grids[blockIdx.x];
	for (uint32_t w = blockIdx.y * MULLS_BLOCK + threadIdx.x; w < g.ncell; w += gridDim.y * MULLS_BLOCK)
		bm[g.cell_off + w] = 0ull;
}

__global__ __launch_bounds__(MULLS_BLOCK) void k_bm_mark(const Job *__restrict__ tjobs, const CloudDesc *__restrict__ descs,
														  const GridDesc *__restrict__ grids, const float4 *__restrict__ tpos,
														  unsigned long long *__restrict__ bm)
{
	const Job job = tjobs[blockIdx.x];
	const CloudDesc &d = descs[job.pair * MULLS_NC + job.cls];
	const uint32_t t = job.start + threadIdx.x;
	if (t >= d.tgt_n)
		return;
	const GridDesc g = grids[job.pair * MULLS_NC + job.cls];
	const float4 p = tpos[d.tgt_off + t];
	const uint32_t bit = bm_bit(g, p.x, p.y, p.z);
	unsigned long long *word = &bm[g.cell_off + (bit >> 6)];
	const unsigned long long b = 1ull << (bit & 63u);
	if (!(__builtin_nontemporal_load(word) & b)) // dense maps put tens of points in a cell: most find their bit set already
		atomicOr(word, b);
}

__global__ __launch_bounds__(1024) void k_bm_scan(GridDesc *__restrict__ grids, RunParams rp, const unsigned long long *__restrict__ bm,
												  uint32_t *__restrict__ pf)
{
	const uint32_t cls = blockIdx.x % MULLS_NC;
	if (!rp.used[cls])
		return;
	const GridDesc g = grids[blockIdx.x];
	const unsigned long long *b = bm + g.cell_off;
	uint32_t *p = pf + g.cell_off;
	const uint32_t nocc = block_scan_1024(
		g.ncell, [&](uint32_t w) { return (uint32_t)__popcll(b[w]); }, [&](uint32_t w, uint32_t ex) { p[w] = ex; });
	if (threadIdx.x == 0)
		grids[blockIdx.x].nocc = nocc;
}

__global__ __launch_bounds__(MULLS_BLOCK) void k_bm_count(const Job *__restrict__ tjobs, const CloudDesc *__restrict__ descs,
														   const GridDesc *__restrict__ grids, const float4 *__restrict__ tpos,
														   const unsigned long long *__restrict__ bm, const uint32_t *__restrict__ pf,
														   uint32_t *__restrict__ cnt)
{
	const Job job = tjobs[blockIdx.x];
	const uint32_t ci = job.pair * MULLS_NC + job.cls;
	const CloudDesc &d = descs[ci];
	const uint32_t t = job.start + threadIdx.x;
	if (t >= d.tgt_n)
		return;
	const GridDesc g = grids[ci];
	const float4 p = tpos[d.tgt_off + t];
	atomicAdd(&cnt[d.tgt_off + ci + bm_rank(bm + g.cell_off, pf + g.cell_off, bm_bit(g, p.x, p.y, p.z))], 1u);
}

// counts -> start positions; the counters are left at zero so that k_bm_scatter can reuse them as insertion cursors
__global__ __launch_bounds__(1024) void k_bm_starts(const CloudDesc *__restrict__ descs, const GridDesc *__restrict__ grids, RunParams rp,
													uint32_t *__restrict__ cnt, uint32_t *__restrict__ cs)
{
	const uint32_t cls = blockIdx.x % MULLS_NC;
	if (!rp.used[cls])
		return;
	const GridDesc g = grids[blockIdx.x];
	const uint32_t off = descs[blockIdx.x].tgt_off + blockIdx.x;
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

// counting-sort scatter: target positions ordered by cell, original index carried in .w
__global__ __launch_bounds__(MULLS_BLOCK) void k_bm_scatter(const Job *__restrict__ tjobs, const CloudDesc *__restrict__ descs,
															 const GridDesc *__restrict__ grids, const float4 *__restrict__ tpos,
															 const unsigned long long *__restrict__ bm, const uint32_t *__restrict__ pf,
															 uint32_t *__restrict__ cnt, const uint32_t *__restrict__ cs, float4 *__restrict__ tsorted)
{
	const Job job = tjobs[blockIdx.x];
	const uint32_t ci = job.pair * MULLS_NC + job.cls;
	const CloudDesc &d = descs[ci];
	const uint32_t t = job.start + threadIdx.x;
	if (t >= d.tgt_n)
		return;
	const GridDesc g = grids[ci];
	const float4 p = tpos[d.tgt_off + t];
	const uint32_t r = d.tgt_off + ci + bm_rank(bm + g.cell_off, pf + g.cell_off, bm_bit(g, p.x, p.y, p.z));
	const uint32_t slot = cs[r] + atomicAdd(&cnt[r], 1u);
	tsorted[d.tgt_off + slot] = make_float4(p.x, p.y, p.z, __int_as_float((int)t));
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
	if (g.ncell == 0)
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
// host-callable launch wrappers (the driver is plain C++ and never sees <<< >>>)
#include "launch.h"

void launch_grid_build(hipStream_t st, uint32_t npairs, uint32_t ntjobs, const Job *tjobs, const CloudDesc *descs, GridDesc *grids,
					   const RunParams &rp, const float4 *tpos, unsigned long long *bm, uint32_t *pf, uint32_t *cnt, uint32_t *cell_start,
					   float4 *tsorted, bool lds_tier)
{
	if (!ntjobs || !npairs)
		return;
	if (lds_tier)
	{
		hipLaunchKernelGGL(k_grid_build_sort, dim3(npairs * MULLS_NC), dim3(MULLS_LDS_BLOCK), 0, st, descs, grids, rp, tpos, cell_start, tsorted);
		return;
	}
	// global-memory tier: occupancy bitmap + ranks + counting sort by rank (cell_start holds the start positions)
	hipLaunchKernelGGL(k_bm_clear, dim3(npairs * MULLS_NC, npairs >= 64 ? 4 : 64), dim3(MULLS_BLOCK), 0, st, grids, rp, bm);
	hipLaunchKernelGGL(k_bm_mark, dim3(ntjobs), dim3(MULLS_BLOCK), 0, st, tjobs, descs, grids, tpos, bm);
	hipLaunchKernelGGL(k_bm_scan, dim3(npairs * MULLS_NC), dim3(1024), 0, st, grids, rp, bm, pf);
	hipLaunchKernelGGL(k_bm_count, dim3(ntjobs), dim3(MULLS_BLOCK), 0, st, tjobs, descs, grids, tpos, bm, pf, cnt);
	hipLaunchKernelGGL(k_bm_starts, dim3(npairs * MULLS_NC), dim3(1024), 0, st, descs, grids, rp, cnt, cell_start);
	hipLaunchKernelGGL(k_bm_scatter, dim3(ntjobs), dim3(MULLS_BLOCK), 0, st, tjobs, descs, grids, tpos, bm, pf, cnt, cell_start, tsorted);
}
