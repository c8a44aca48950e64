// k_reduce.hip — normal-equation accumulation, per-pair finish, the loop's per-iteration step on the device (k_step), host exchange of
// states / results (host-stepped loop), stage helpers
// (gfx950 / CDNA4, wave64; numerics policy and launch geometry: device_util.h)
#include "device_util.h"
#include "accum.h"
#include "solve_wave.h"

// Normal-equation accumulation (active pairs) or posterior residual (pairs flagged want_residual), lock-step path.  The job
// table is the one of the search (512 source slots per job); one workgroup per job that starts a trip of MULLS_ACC_LANES slots
// (`leaders`) sums the whole trip in the library's summation order (accum.h) into that job's slot of `partial`; k_finish
// adds the trip partials in order (run-to-run deterministic, unlike atomicAdd(double), and the same bits as k_icp).
// LANES: the workgroup size and the most slots the trip can have.  A trip of 100-400 slots (the roof, beam and pillar clouds of a scan, the tail of
// a facade cloud) in a 1024-lane workgroup occupies a 56-KiB buffer and sixteen waves for the latency of its four barriers; as a 256- or
// 512-lane workgroup it shares the CU with seven or three others.  The sums are the same (accum.h: reduce_terms).
template <int LANES, int MIN_WG>
__global__ __launch_bounds__(LANES, MIN_WG) void k_accum(const uint32_t *__restrict__ leaders, const Job *__restrict__ jobs, const CloudDesc *__restrict__ descs,
														 const PairState *__restrict__ states, RunParams rp, const float4 *__restrict__ spos,
														 const float4 *__restrict__ mq, const uint8_t *__restrict__ flag, float *__restrict__ wd,
														 double *__restrict__ partial)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	__shared__ double part[MULLS_NTERM_PAD];
	const uint32_t job_idx = leaders[xcd_job(blockIdx.x, gridDim.x)]; // a job that starts a trip (index into the batch-wide job table)
	const Job job = jobs[job_idx];
	const PairState &ps = states[job.pair];
	if (!ps.active && !ps.want_residual)
		return;
	const CloudDesc *pd = descs + job.pair * MULLS_NC;
	const bool residual_pass = ps.want_residual != 0;
	int cnt[MULLS_NC];
	for (int c = 0; c < MULLS_NC; c++)
		cnt[c] = (int)(class_called(rp, pd[c], c) ? pd[c].valid_next : pd[c].n_valid);
	const AccumCtx A = accum_ctx(rp, job.cls, ps.iter, residual_pass, class_weight(rp, job.cls, residual_pass, cnt));
	trip_sum<true, LANES>(A, ps.x, pd[job.cls], job.start, spos, mq, flag, wd, lds_raw, part); // two halves of 14 terms: 56 / 28 / 14 KiB of LDS
	if (threadIdx.x < MULLS_NTERM)
		partial[(size_t)job_idx * MULLS_NTERM + threadIdx.x] = part[threadIdx.x];
}

// The same sums by ONE WAVE per trip (round 5) — for launches with enough trips to fill the chip with waves.  The library's summation order reads "lane j of a
// wave adds the values j, j + 64, ..., j + 960 in that order, then a butterfly adds the 64 partial sums" (accum.h).  k_accum realises it by 1024 lanes evaluating one slot
// each, a term buffer in LDS and waves that walk it term by term: 53 B of a slot's records in, 108 B of terms through LDS and back, four barriers per trip and two
// workgroups per CU (56 KiB each) — the launch ran at 3 TB/s of its own bytes.  Here lane j of a wave evaluates the slots j, j + 64, ... one after the other and keeps the 27
// running sums in registers: the additions are the very same ones in the very same order (a running sum starts as -0.0, the one value x + (-0.0) == x holds for bit for bit),
// no LDS, no barrier, sixteen independent waves per CU with the next slot's records in flight while a slot's terms are evaluated.  Trips of point-to-line classes outside
// the faithful combined-system mode (27 terms of doubles: 108 registers of sums and terms) stay with k_accum; launch_accum decides per launch.
namespace
{
template <typename TS, int W0, int WIN, int MODE, int METRIC, int A0>
__device__ __forceinline__ void add_window(const AccumCtx &A, const double *x, const float4 P, const float4 Q, const float4 N, float wi, float &w, double *acc, const float *wpre)
{
	TS t[WIN];
#pragma unroll
	for (int k = 0; k < WIN; k++)
		t[k] = (TS)0;
	point_terms<TS, W0, WIN, MODE, METRIC>(A, x, P, Q, N, wi, w, t, wpre);
#pragma unroll
	for (int k = 0; k < WIN; k++)
		acc[W0 - A0 + k] += (double)t[k];
}
// the terms [W0, W0 + REM) in windows of at most MULLS_ACCW_WIN
#ifndef MULLS_ACCW_WIN
#define MULLS_ACCW_WIN 9 // float terms per window
#endif
#ifndef MULLS_ACCW_WIN_D
#define MULLS_ACCW_WIN_D 12 // double terms per window
#endif
template <typename TS, int W0, int REM, int MODE, int METRIC, int A0>
__device__ __forceinline__ void add_windows(const AccumCtx &A, const double *x, const float4 P, const float4 Q, const float4 N, float wi, float &w, double *acc, const float *wpre)
{
	constexpr int WMAX = sizeof(TS) == 8 ? MULLS_ACCW_WIN_D : MULLS_ACCW_WIN;
	constexpr int WIN = REM < WMAX ? REM : WMAX;
	add_window<TS, W0, WIN, MODE, METRIC, A0>(A, x, P, Q, N, wi, w, acc, wpre);
	if constexpr (REM > WIN)
		add_windows<TS, W0 + WIN, REM - WIN, MODE, METRIC, A0>(A, x, P, Q, N, wi, w, acc, wpre);
}
// T0, NTW: the terms [T0, T0 + NTW) of the NT the form has are this wave's (two waves share a float trip: 14 + 13 running sums instead of 27 leave registers for the
// next slot's records in flight and a fifth wave per SIMD)
template <typename TS, int NT, int MODE, int METRIC, int T0, int NTW>
__device__ __forceinline__ void wave_trip(const AccumCtx &A, const double *x, const CloudDesc &d, uint32_t trip0, const float4 *__restrict__ spos, const float4 *__restrict__ mq,
										   const uint8_t *__restrict__ flag, float *__restrict__ wd, double *__restrict__ out, bool owner)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t src_n = d.src_n, src_off = d.src_off;
	const uint32_t n_here = min(src_n - trip0, (uint32_t)MULLS_ACC_LANES), iters = (n_here + 63u) >> 6; // (uniform)
	const uint32_t last = src_off + src_n - 1u;
	struct Rec
	{
		uint32_t f;
		float4 P, Q;
		float3 N;
		float w;
	};
	// pcl::Correspondence's distance / weight word: the residual pass reads it (the weight the last iteration left, or d^2 for the point-to-point class: SURVEY A.7); a
	// normal-equation pass only writes it — point-to-plane and point-to-line always, point-to-point in the non-faithful mode — so it is not fetched there (the search
	// kernel has just written the squared distance into it: the old "store if the bits differ" test never skipped a store, and cost 4 of the slot's 53 bytes)
	const bool need_w = A.residual_pass, writes_w = !A.residual_pass && (METRIC != 2 || !A.faithful); // (uniform)
	auto load = [&](uint32_t i) {
		Rec r;
		const uint32_t g = min(src_off + trip0 + lane + 64u * i, last); // (slots beyond the cloud re-read its last point: no branch between the loads)
		r.f = flag[g];
		r.P = spos[g], r.Q = mq[2u * g], r.N = *reinterpret_cast<const float3 *>(mq + 2u * g + 1u);
		r.w = need_w ? wd[g] : 0.0f;
		return r;
	};
	double acc[NTW];
#pragma unroll
	for (int k = 0; k < NTW; k++)
		acc[k] = -0.0;
	// (the records are used under the validity test only, and the compiler would sink their loads behind it — flag first, then the rest: two round trips in
	// sequence per slot; values named by a volatile asm have been loaded by then)
#define MULLS_PIN_REC(r) asm volatile("" : "+v"((r).f), "+v"((r).P.x), "+v"((r).Q.x), "+v"((r).N.x), "+v"((r).w))
#if MULLS_ACCW_PREFETCH
	Rec cur = load(0);
#endif
	for (uint32_t i = 0; i < iters; i++)
	{
#if MULLS_ACCW_PREFETCH
		MULLS_PIN_REC(cur);
		Rec nxt = cur;
		if (i + 1u < iters)
			nxt = load(i + 1u); // the next slot's records are in flight while this slot's terms are evaluated
#else
		Rec cur = load(i);
		MULLS_PIN_REC(cur);
#endif
		const uint32_t s = trip0 + lane + 64u * i;
		const bool valid = s < src_n && (cur.f & (MULLS_F_ALIVE | MULLS_F_VALID)) == (MULLS_F_ALIVE | MULLS_F_VALID);
		// A dead or unmatched slot's terms are +0.0: adding them changes a running sum only from -0.0 to +0.0, which k_finish's `0.0 + partial` does anyway (the 512- and
		// 256-lane forms of k_accum leave the slots beyond their lanes out the same way) — so such a slot is skipped altogether.
		// The terms in windows: a window's terms are added to their running sums before the next window's are evaluated (the windows share the row vector through
		// common-subexpression elimination; the weight is handed to them).
		if (valid)
		{
			float w = cur.w;
			const float4 N4 = make_float4(cur.N.x, cur.N.y, cur.N.z, 0.0f);
			// The intensity weight exp(-|i1 - i2| / 255) is evaluated every time.  (Round 5 kept a 16-byte memo of it per slot, when the kernel was bound by its
			// instruction count; with the weights evaluated once per slot — corr_weight — it waits for memory, and the memo's bytes cost more than the exp:
			// 4.0 -> 3.45 ms of accumulation per 4096-pair step, profiles/r06_experiments.txt item 13.)
			const float wi = (A.inten_w && !A.residual_pass) ? point_wi(cur.P, cur.Q) : 1.0f;
			// the correspondence's weight once per slot, in front of the term windows (accum.h: corr_weight — each window would evaluate its own copy)
			const float wpre = A.residual_pass ? 0.0f : corr_weight(A, METRIC, cur.P, cur.Q, N4, wi);
			add_windows<TS, T0, NTW, MODE, METRIC, T0>(A, x, cur.P, cur.Q, N4, wi, w, acc, A.residual_pass ? nullptr : &wpre);
			if (owner && writes_w)
				wd[src_off + s] = w; // pcl::Correspondence::weight (the wave that holds the form's first terms writes it: the weight does not depend on the terms)
		}
#if MULLS_ACCW_PREFETCH
		cur = nxt;
#endif
	}
	// the butterfly of reduce_terms, term by term: the four row sums as there, then (row0 + row1) and (row2 + row3) by row_bcast:15 into rows 1 and 3, and their sum by
	// row_bcast:31 into row 3 — (r2 + r3) + (r0 + r1), the same bits as (r0 + r1) + (r2 + r3) — so that lane 63 holds the trip's sum and stores it (readlanes and
	// a select per term were 900 instructions a trip, an eighth of the kernel)
	if (NTW == NT && lane < MULLS_NTERM) // the whole form in one wave: it also writes the entries the form does not have (0.0; no entry is written twice)
	{
		const bool has = MODE == 1 ? li_slot((int)lane) >= 0 : (int)lane < NT;
		if (!has)
			out[lane] = 0.0;
	}
#pragma unroll
	for (int k = 0; k < NTW; k++)
	{
		double sum = acc[k];
		sum = dpp_add_f64<0xB1>(sum);	// quad_perm [1,0,3,2]
		sum = dpp_add_f64<0x4E>(sum);	// quad_perm [2,3,0,1]
		sum = dpp_add_f64<0x141>(sum); // row_half_mirror
		sum = dpp_add_f64<0x140>(sum); // row_mirror: every lane of a row holds the row's sum
		sum = dpp_add_f64_rows<0x142, 0xa>(sum); // row_bcast:15 -> rows 1, 3
		sum = dpp_add_f64_rows<0x143, 0xc>(sum); // row_bcast:31 -> rows 2, 3
		acc[k] = sum;
	}
	if (lane == 63u)
	{
#pragma unroll
		for (int k = 0; k < NTW; k++)
			out[MODE == 1 ? li_term(T0 + k) : T0 + k] = acc[k];
	}
}
} // namespace

#ifndef MULLS_ACCW_SPLIT
#define MULLS_ACCW_SPLIT 1 // waves that share a float trip (1: one wave holds all 27 running sums; 2: 14 + 13)
#endif
#ifndef MULLS_ACCW_PREFETCH
#define MULLS_ACCW_PREFETCH 0
#endif
#ifndef MULLS_ACCW_OCC
#define MULLS_ACCW_OCC 4
#endif
#define MULLS_ACCW_WAVES 4 // waves per workgroup of k_accum_wave
__global__ __launch_bounds__(64 * MULLS_ACCW_WAVES, MULLS_ACCW_OCC) void k_accum_wave(const uint32_t *__restrict__ leaders, uint32_t n_trips, const Job *__restrict__ jobs,
																					   const CloudDesc *__restrict__ descs, const PairState *__restrict__ states, RunParams rp,
																					   const float4 *__restrict__ spos, const float4 *__restrict__ mq, const uint8_t *__restrict__ flag,
																					   float *__restrict__ wd, double *__restrict__ partial)
{
	const uint32_t wave = threadIdx.x >> 6, part = wave % MULLS_ACCW_SPLIT;
	// (blockIdx in launch order, NOT xcd_job's contiguous eighths: the leaders come longest trip first, so an eighth of the table per XCD gives XCD 0 the sixteen-slot
	// trips and XCD 7 the three-slot ones — the launch then lasts as long as XCD 0; round robin deals every length to every XCD)
	const uint32_t trip = blockIdx.x * (MULLS_ACCW_WAVES / MULLS_ACCW_SPLIT) + wave / MULLS_ACCW_SPLIT;
	if (trip >= n_trips)
		return;
	const uint32_t job_idx = leaders[trip];
	const Job job = jobs[job_idx];
	const PairState &ps = states[job.pair];
	if (!ps.active && !ps.want_residual)
		return;
	const CloudDesc *pd = descs + job.pair * MULLS_NC;
	const bool residual_pass = ps.want_residual != 0;
	int cnt[MULLS_NC];
	for (int c = 0; c < MULLS_NC; c++)
		cnt[c] = (int)(class_called(rp, pd[c], c) ? pd[c].valid_next : pd[c].n_valid);
	const AccumCtx A = accum_ctx(rp, job.cls, ps.iter, residual_pass, class_weight(rp, job.cls, residual_pass, cnt));
	const CloudDesc &d = pd[job.cls];
	double *out = partial + (size_t)job_idx * MULLS_NTERM;
	if (d.src_n <= job.start) // (a trip starts inside its cloud; an emptied descriptor must not turn into a wild index)
	{
		if (part == 0u && (threadIdx.x & 63u) < MULLS_NTERM)
			out[threadIdx.x & 63u] = 0.0;
		return;
	}
	// (the metric is a template argument: one kernel holds every form, and its registers are those of the largest form that is compiled, not of all of point_terms)
	if (residual_pass || A.metric == 1)
	{
		if (part) // two sums, or the twelve of the faithful point-to-line system: one wave
			return;
		if (!residual_pass)
			wave_trip<double, 12, 1, 1, 0, 12>(A, ps.x, d, job.start, spos, mq, flag, wd, out, true); // (li_diag: launch_accum keeps other point-to-line sums away)
		else if (A.metric == 0)
			wave_trip<double, 2, 0, 0, 0, 2>(A, ps.x, d, job.start, spos, mq, flag, wd, out, true);
		else if (A.metric == 1)
			wave_trip<double, 2, 0, 1, 0, 2>(A, ps.x, d, job.start, spos, mq, flag, wd, out, true);
		else
			wave_trip<double, 2, 0, 2, 0, 2>(A, ps.x, d, job.start, spos, mq, flag, wd, out, true);
	}
#if MULLS_ACCW_SPLIT == 2
	else if (A.metric == 0)
	{
		if (part == 0u)
			wave_trip<float, 27, 0, 0, 0, 14>(A, ps.x, d, job.start, spos, mq, flag, wd, out, true);
		else
			wave_trip<float, 27, 0, 0, 14, 13>(A, ps.x, d, job.start, spos, mq, flag, wd, out, false);
	}
	else
	{
		if (part == 0u)
			wave_trip<float, 27, 0, 2, 0, 14>(A, ps.x, d, job.start, spos, mq, flag, wd, out, true);
		else
			wave_trip<float, 27, 0, 2, 14, 13>(A, ps.x, d, job.start, spos, mq, flag, wd, out, false);
	}
#else
	else if (A.metric == 0)
		wave_trip<float, 27, 0, 0, 0, 27>(A, ps.x, d, job.start, spos, mq, flag, wd, out, true);
	else
		wave_trip<float, 27, 0, 2, 0, 27>(A, ps.x, d, job.start, spos, mq, flag, wd, out, true);
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// One workgroup per pair: sum the per-job partials of every class in job order, then roll the per-class counters
// over to the next iteration.
//
namespace
{
// what finish_pair leaves on chip for a step that follows in the same workgroup (k_finish_step): the class rows, the combined row, the record's counter block
struct FinishLds
{
	double sums[MULLS_NC][MULLS_NTERM_PAD], comb[MULLS_NTERM_PAD];
	uint32_t cnt[32]; // n_valid, n_alive, src_n, tgt_n, bbox (6 each)
};
// M (k_finish_step): the rows are combined from LDS and the step reads them there — the record in memory is written all the same (pulled by the host-stepped
// loop, read by nothing else in that launch), but no result of this workgroup makes the round trip through memory
__device__ __forceinline__ void finish_pair(CloudDesc *pd, const PairState &ps, const RunParams &rp, const double *__restrict__ partial,
											 PairOut &o, const uint32_t *__restrict__ pair_bbox, FinishLds *M = nullptr)
{
	if (threadIdx.x >= 192 && threadIdx.x < 198)
	{
		const uint32_t v = pair_bbox[threadIdx.x - 192];
		o.bbox[threadIdx.x - 192] = v;
		if (M)
			M->cnt[24 + threadIdx.x - 192] = v;
	}
	if (threadIdx.x < MULLS_NC * MULLS_NTERM)
	{
		const int c = threadIdx.x / MULLS_NTERM, t = threadIdx.x % MULLS_NTERM;
		if (rp.used[c]) // unused classes contribute nothing: their slots are not even sent over PCIe
		{
			// the trip partials in trip order (the library's one summation order).  A dense class cloud has a hundred of them: sixteen loads in flight, then
			// the chain of adds — one lane's dependent round trips were 41 us of a 236 k-point pair's 170 us iteration (profiles/r04_large_base.txt)
			constexpr uint32_t STEP = MULLS_ACC_LANES / MULLS_SRC_PER_BLOCK; // the jobs that start a trip (k_accum)
			double sum = 0.0;
			uint32_t j = pd[c].job_begin;
			const uint32_t je = pd[c].job_end;
			for (; j + 32u * STEP <= je; j += 32u * STEP) // (thirty-two in flight: a 108 k-point class cloud's 106 partials are four round trips instead of seven)
			{
				double v[32];
#pragma unroll
				for (uint32_t k = 0; k < 32u; k++)
					v[k] = partial[(size_t)(j + k * STEP) * MULLS_NTERM + t];
#pragma unroll
				for (uint32_t k = 0; k < 32u; k++)
					sum += v[k];
			}
			for (; j + 16u * STEP <= je; j += 16u * STEP)
			{
				double v[16];
#pragma unroll
				for (uint32_t k = 0; k < 16u; k++)
					v[k] = partial[(size_t)(j + k * STEP) * MULLS_NTERM + t];
#pragma unroll
				for (uint32_t k = 0; k < 16u; k++)
					sum += v[k];
			}
			for (; j < je; j += STEP)
				sum += partial[(size_t)j * MULLS_NTERM + t];
			o.sums[c][t] = sum;
			if (M)
				M->sums[c][t] = sum;
		}
	}
	__syncthreads();
	if (rp.pull_comb && threadIdx.x >= 64 && threadIdx.x < 64 + MULLS_NTERM)
	{
		if (M)
		{
			combine_rows(rp, ps.want_residual != 0, M->sums, M->comb, (int)threadIdx.x - 64);
			o.comb[threadIdx.x - 64] = M->comb[threadIdx.x - 64];
		}
		else
			combine_rows(rp, ps.want_residual != 0, o.sums, o.comb, (int)threadIdx.x - 64); // only this row crosses PCIe (224 B instead of 224 B per used class)
	}
	if (threadIdx.x < MULLS_NC)
	{
		const int c = threadIdx.x;
		CloudDesc &d = pd[c];
		if (ps.active)
		{
			if (class_called(rp, d, c))
			{
				d.n_valid = d.valid_next;
				d.alive_cur = d.alive_next;
			}
			d.alive_next = 0;
			d.valid_next = 0;
			d.n_matched = 0;
		}
		o.n_valid[c] = d.n_valid;
		o.n_alive[c] = d.alive_cur;
		o.src_n[c] = d.src_n;
		o.tgt_n[c] = d.tgt_n;
		if (M)
			M->cnt[c] = d.n_valid, M->cnt[6 + c] = d.alive_cur, M->cnt[12 + c] = d.src_n, M->cnt[18 + c] = d.tgt_n;
	}
}
// finish_pair by ONE wave (k_sum_step): lane l takes the sums l, l + 64, l + 128 of the 6 x 27 — every trip partial of the three requested before the first
// addition (a sum with more than 16 trips falls back to a loop: dense class clouds do not take this path) — the same additions in the same order.
__device__ __forceinline__ void finish_pair_wave(CloudDesc *pd, const PairState &ps, const RunParams &rp, const double *__restrict__ partial, PairOut &o,
												  const uint32_t *__restrict__ pair_bbox, FinishLds &M)
{
	const uint32_t l = threadIdx.x;
	constexpr uint32_t STEP = MULLS_ACC_LANES / MULLS_SRC_PER_BLOCK;
	if (l >= 32u && l < 38u)
	{
		const uint32_t v = pair_bbox[l - 32u];
		o.bbox[l - 32u] = v;
		M.cnt[24u + l - 32u] = v;
	}
	double sum[3] = {0.0, 0.0, 0.0};
	uint32_t j[3], je[3];
	bool on[3];
#pragma unroll
	for (int u = 0; u < 3; u++)
	{
		const uint32_t idx = l + 64u * (uint32_t)u, c = idx / MULLS_NTERM;
		on[u] = idx < MULLS_NC * MULLS_NTERM && rp.used[c < MULLS_NC ? c : 0u];
		j[u] = on[u] ? pd[c].job_begin : 0u;
		je[u] = on[u] ? pd[c].job_end : 0u;
	}
	double v[3][4];
#pragma unroll
	for (int u = 0; u < 3; u++)
	{
		const uint32_t t = (l + 64u * (uint32_t)u) % MULLS_NTERM;
#pragma unroll
		for (uint32_t k = 0; k < 4u; k++)
			v[u][k] = j[u] + k * STEP < je[u] ? partial[(size_t)(j[u] + k * STEP) * MULLS_NTERM + t] : 0.0;
	}
#pragma unroll
	for (int u = 0; u < 3; u++)
	{
		const uint32_t idx = l + 64u * (uint32_t)u, c = idx / MULLS_NTERM, t = idx % MULLS_NTERM;
#pragma unroll
		for (uint32_t k = 0; k < 4u; k++)
			if (j[u] + k * STEP < je[u])
				sum[u] += v[u][k];
		for (uint32_t jj = j[u] + 4u * STEP; jj < je[u]; jj += STEP)
			sum[u] += partial[(size_t)jj * MULLS_NTERM + t];
		if (on[u])
		{
			o.sums[c][t] = sum[u];
			M.sums[c][t] = sum[u];
		}
	}
	__syncthreads();
	if (rp.pull_comb && l < MULLS_NTERM)
	{
		combine_rows(rp, ps.want_residual != 0, M.sums, M.comb, (int)l);
		o.comb[l] = M.comb[l];
	}
	if (l >= 56u && l < 56u + MULLS_NC)
	{
		const int c = (int)l - 56;
		CloudDesc &d = pd[c];
		if (ps.active)
		{
			if (class_called(rp, d, c))
			{
				d.n_valid = d.valid_next;
				d.alive_cur = d.alive_next;
			}
			d.alive_next = 0;
			d.valid_next = 0;
			d.n_matched = 0;
		}
		o.n_valid[c] = d.n_valid;
		o.n_alive[c] = d.alive_cur;
		o.src_n[c] = d.src_n;
		o.tgt_n[c] = d.tgt_n;
		M.cnt[c] = d.n_valid, M.cnt[6 + c] = d.alive_cur, M.cnt[12 + c] = d.src_n, M.cnt[18 + c] = d.tgt_n;
	}
}
} // namespace

__global__ __launch_bounds__(MULLS_BLOCK) void k_finish(CloudDesc *__restrict__ descs, const PairState *__restrict__ states, RunParams rp,
														 const double *__restrict__ partial, PairOut *__restrict__ out, const uint32_t *__restrict__ bbox,
														 uint32_t pair_base, uint4 *__restrict__ host_words, uint32_t *__restrict__ ticket,
														 volatile uint32_t *host_epoch, uint32_t epoch)
{
	const uint32_t pair = pair_base + blockIdx.x;
	const int active = states[pair].active, want_residual = states[pair].want_residual;
	if (active || want_residual) // uniform per workgroup
		finish_pair(descs + pair * MULLS_NC, states[pair], rp, partial, out[pair], bbox + pair * 6);
	if (!host_words)
		return; // k_pull_outs packs and ships the records
	// Small batches (launch_finish decides): this workgroup ships its pair's record itself — counter block (8 uint4) + combined
	// row (14 uint4), 16-B stores into pinned host memory — and the last workgroup to arrive publishes the epoch the host
	// spins on: one launch less per iteration where the iteration is launch-bound.  (With thousands of pairs the per-workgroup
	// system-scope fences make this slower than the packed k_pull_outs: 91 k vs 104 k registrations/s at 4096 pairs.)
	__syncthreads();
	const uint32_t row_words = MULLS_NTERM_PAD / 2, head_words = 8, wpp = head_words + row_words;
	if (threadIdx.x < wpp)
	{
		const uint4 *dev = reinterpret_cast<const uint4 *>(&out[pair]);
		const uint32_t src = threadIdx.x < head_words ? (MULLS_NC + 1u) * row_words + threadIdx.x : MULLS_NC * row_words + (threadIdx.x - head_words);
		host_words[(size_t)pair * wpp + threadIdx.x] = dev[src];
	}
	__threadfence_system();
	__syncthreads();
	if (threadIdx.x == 0)
	{
		const uint32_t t = atomicAdd(ticket, 1u);
		if (t == gridDim.x - 1u)
		{
			*ticket = 0u; // re-armed for the next launch (stream order: nobody else touches it before)
			__threadfence_system();
			*host_epoch = epoch;
		}
	}
}

static_assert(sizeof(PairOut) == (MULLS_NC + 1) * MULLS_NTERM_PAD * 8 + 128, "PairOut = class rows + combined row + one 128-B counter block");
__global__ __launch_bounds__(MULLS_BLOCK) void k_pull_outs(const uint4 *__restrict__ dev_words, uint4 *__restrict__ host_words, RunParams rp,
															uint32_t pair_base, uint32_t npairs, uint32_t *__restrict__ ticket,
															volatile uint32_t *host_epoch, uint32_t epoch)
{
	const uint32_t row_words = MULLS_NTERM_PAD / 2, head_words = 8, rec_words = sizeof(PairOut) / 16;
	uint32_t n_used = 0;
	for (int c = 0; c < MULLS_NC; c++)
		n_used += rp.used[c] ? 1u : 0u;
	const uint32_t wpp = head_words + row_words * (rp.pull_comb ? 1u : n_used);
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < npairs * wpp)
	{
		const uint32_t p = pair_base + i / wpp, w = i % wpp;
		uint32_t src = (MULLS_NC + 1u) * row_words + w; // counter block
		if (w >= head_words && rp.pull_comb)
			src = MULLS_NC * row_words + (w - head_words); // the combined row
		else if (w >= head_words)
		{
			uint32_t rank = (w - head_words) / row_words, cls = 0;
			for (uint32_t c = 0; c < MULLS_NC; c++)
				if (rp.used[c])
				{
					if (rank == 0)
					{
						cls = c;
						break;
					}
					rank--;
				}
			src = cls * row_words + (w - head_words) % row_words;
		}
		host_words[(size_t)p * wpp + w] = dev_words[(size_t)p * rec_words + src];
	}
	__threadfence_system();
	__syncthreads();
	if (threadIdx.x == 0)
	{
		const uint32_t t = atomicAdd(ticket, 1u);
		if (t == gridDim.x - 1u)
		{
			*ticket = 0u; // re-armed for the next launch (stream order: nobody else touches it before)
			__threadfence_system();
			*host_epoch = epoch;
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------
// Lock-step loop with the O(1) half of the iteration on the device.  k_step_init prepares what the host path keeps in PairHost (pair_iter_init,
// the first PairState, an empty result); k_step, launched behind k_finish, does what driver.cpp's host_step does with the record k_pull_outs
// would have shipped: first-iteration bookkeeping, the correspondence-count test and threshold update (step_counts), the 6x6 solve with its
// step and convergence tests (solve_wave = step_solve), the posterior residual (step_residual), the next PairState — the very functions the
// host calls (icp_step.h), so the two paths produce the same bits — and, once the pair is done, its result record.  One wave per pair (every
// pair of the batch is resident at once: the solve is a latency chain), the pair's state staged in LDS.  Nothing crosses PCIe per iteration
// except one 8-byte word: (epoch << 32 | pairs still iterating), published by k_step_publish.
__global__ __launch_bounds__(64) void k_step_init(uint32_t npairs, const PairSetup *__restrict__ setup, mulls::IcpConst K, mulls::StepState *__restrict__ steps,
												  PairState *__restrict__ states)
{
	const uint32_t pair = blockIdx.x * 64u + threadIdx.x;
	if (pair >= npairs)
		return;
	mulls::StepState &S = steps[pair];
	static_assert(sizeof(mulls::StepState) % 8 == 0, "StepState is cleared / moved in 8-byte words");
	unsigned long long *w = reinterpret_cast<unsigned long long *>(&S);
	for (uint32_t k = 0; k < sizeof(mulls::StepState) / 8; k++)
		w[k] = 0ull;
	mulls::pair_iter_init(S.h, setup[pair].guess, K);
	S.first = 1u;
	PairState &ps = states[pair];
	for (int k = 0; k < 12; k++)
		ps.T[k] = (k % 5 == 0) ? 1.0 : 0.0; // TempTran = identity at i = 0
	for (int k = 0; k < 6; k++)
		ps.x[k] = 0.0;
	for (int c = 0; c < MULLS_NC; c++)
		ps.thr[c] = K.dis_thre_unit;
	ps.iter = 0;
	ps.active = S.h.active;
	ps.want_residual = 0;
	ps.pad_[0] = ps.pad_[1] = ps.pad_[2] = 0;
}

// the step of one pair by the first wave of the workgroup (k_step: a 64-lane workgroup; k_finish_step: the first wave of the 256 lanes that summed
// the pair's partials — every lane reaches the barriers).  Returns (to every lane) whether the pair still iterates or waits for its residual pass.
struct StepLds
{
	mulls::StepState S;
	SolveWs ws;
	double comb[MULLS_NTERM_PAD];
	uint32_t cnt[32]; // the record's counter block: n_valid, n_alive, src_n, tgt_n, bbox (6 each)
	uint32_t jobs[MULLS_NC], n0[MULLS_NC];
	int go, iter, active, resid, done, left;
};
constexpr uint32_t STEP_WORDS = (uint32_t)(sizeof(mulls::StepState) / 8), STEP_WORDS_PER_LANE = (STEP_WORDS + 63u) / 64u;
// PRE (k_finish_step): the workgroup has filled L.comb / L.cnt from its own finish_pair, and the pair's StepState words were requested at the start of the kernel
// (sw: lanes 0..63; nullptr: k_step)
__device__ __forceinline__ bool step_pair(uint32_t pair, const CloudDesc *__restrict__ descs, PairState *__restrict__ states, const RunParams &rp, const mulls::IcpConst &K,
										   const PairOut *__restrict__ out, mulls::StepState *__restrict__ steps, IcpOut *__restrict__ results, int brute, StepLds &L,
										   const unsigned long long *sw = nullptr)
{
	const int l = (int)threadIdx.x;
	const bool PRE = sw != nullptr; // (one function, inlined into both kernels: two instantiations would each call solve_wave out of line — 168 registers instead of 104)
	mulls::StepState &S = L.S;
	if (l == 0)
	{
		L.active = states[pair].active;
		L.resid = states[pair].want_residual;
		L.iter = states[pair].iter;
		L.left = 0;
	}
	__syncthreads();
	if (L.active || L.resid) // uniform
	{
		const PairOut &o = out[pair];
		if (l < 64)
		{
			const unsigned long long *src = reinterpret_cast<const unsigned long long *>(&steps[pair]);
			unsigned long long *dst = reinterpret_cast<unsigned long long *>(&S);
			if (PRE)
			{
#pragma unroll
				for (uint32_t k = 0; k < STEP_WORDS_PER_LANE; k++)
					if ((uint32_t)l + 64u * k < STEP_WORDS)
						dst[(uint32_t)l + 64u * k] = sw[k];
			}
			else
				for (uint32_t w = (uint32_t)l; w < STEP_WORDS; w += 64u)
					dst[w] = src[w];
		}
		if (!PRE)
		{
			if (l < MULLS_NTERM_PAD)
				L.comb[l] = o.comb[l];
			if (l < 30)
				L.cnt[l] = (&o.n_valid[0])[l]; // n_valid, n_alive, src_n, tgt_n, bbox are consecutive
		}
		if (l < MULLS_NC)
		{
			const CloudDesc &d = descs[pair * MULLS_NC + l];
			L.jobs[l] = d.job_end - d.job_begin;
			L.n0[l] = d.src_n0;
		}
		__syncthreads();
		const uint32_t *n_valid = L.cnt, *n_alive = L.cnt + 6, *src_n = L.cnt + 12, *tgt_n = L.cnt + 18, *obox = L.cnt + 24;
		mulls::PairIter &h = S.h;
		if (l == 0)
		{
			L.go = 0;
			if (L.resid)
				mulls::step_residual(h, K, L.comb[0], L.comb[1]); // get_multi_metrics_lls_residual + information matrix (:2518-2544, :1386)
			else
			{
				const int i = L.iter;
				h.iters = i + 1;
				if (S.first)
				{
					// the sizes the reference counts at :1195-1201 (while undistorting: those of the cloned clouds) and the crop box's keys
					int sfc = 0;
					for (int c = 0; c < MULLS_NC; c++)
					{
						const uint32_t n0 = rp.undistort ? L.n0[c] : src_n[c];
						S.nsrc0[c] = n0;
						S.ntgt0[c] = tgt_n[c];
						S.alive_prev[c] = src_n[c];
						if ((c == 1 || c == 2 || c == 3) && rp.used[c])
							sfc += (int)n0;
					}
					h.src_feature_count = sfc;
					for (int k = 0; k < 6; k++)
						S.bbox[k] = obox[k];
					S.first = 0u;
				}
				for (int c = 0; c < MULLS_NC; c++)
				{
					if (rp.used[c] && S.alive_prev[c] >= 3u && tgt_n[c] >= 3u)
					{
						if (brute)
							S.pair_evals += (unsigned long long)S.alive_prev[c] * tgt_n[c];
						S.src_pts += S.alive_prev[c];
						S.tgt_pts += tgt_n[c];
						S.tgt_job_pts += (unsigned long long)tgt_n[c] * L.jobs[c];
					}
					S.alive_prev[c] = n_alive[c];
					S.ncorr[c] = n_valid[c];
					S.corr_pts += n_valid[c];
				}
				L.go = mulls::step_counts(h, K, n_valid) ? 1 : 0; // :1305-1311, then update_corr_dist_thre :1855-1866
			}
		}
		__syncthreads();
		if (L.go && l < 64)
			solve_wave(h, K, L.comb, L.iter, L.ws); // solve :1924-1964, step test :1348-1354, convergence :1357, guess update :1400
		__syncthreads();
		if (l == 0)
		{
			L.done = (!h.active && !h.want_residual) ? 1 : 0;
			L.left = L.done ? 0 : 1;
			if (L.done)
				h.guess = h.temp * h.guess; // :1403 (TempTran is the identity after a failure)
		}
		__syncthreads();
		// the next iteration's PairState
		PairState &ps = states[pair];
		if (l < 12)
			ps.T[l] = h.temp.at(l / 4, l % 4);
		else if (l < 18)
			ps.x[l - 12] = h.x[l - 12];
		else if (l < 24)
			ps.thr[l - 18] = h.thr[l - 18];
		else if (l == 24)
		{
			ps.iter = h.want_residual ? h.iters - 1 : L.iter + 1;
			ps.active = h.active ? 1 : 0;
			ps.want_residual = h.want_residual ? 1 : 0;
		}
		if (l < 64)
		{
			unsigned long long *dst = reinterpret_cast<unsigned long long *>(&steps[pair]);
			const unsigned long long *src = reinterpret_cast<const unsigned long long *>(&S);
			for (uint32_t w = (uint32_t)l; w < sizeof(mulls::StepState) / 8; w += 64u)
				dst[w] = src[w];
		}
		if (L.done) // the result record (the fields k_icp's timers fill stay zero)
		{
			IcpOut &O = results[pair];
			if (l < 16)
				O.T[l] = h.guess.v[l];
			if (l < 36)
				O.info[l] = h.info.v[l];
			if (l >= 40 && l < 46)
			{
				const int c = l - 40;
				O.ncorr[c] = S.ncorr[c];
				O.nsrc0[c] = S.nsrc0[c];
				O.ntgt0[c] = S.ntgt0[c];
				O.bbox[c] = S.bbox[c];
			}
			if (l == 63)
			{
				O.sigma2 = h.sigma2;
				O.ratio = h.ratio;
				O.code = h.code;
				O.iters = h.iters;
				O.singular = h.singular;
				O.trace_len = 0;
				O.src_pts = S.src_pts;
				O.tgt_pts = S.tgt_pts;
				O.corr_pts = S.corr_pts;
				O.tgt_job_pts = S.tgt_job_pts;
				O.pair_evals = S.pair_evals;
			}
		}
	}
	return L.left != 0;
}

__global__ __launch_bounds__(64) void k_step(const CloudDesc *__restrict__ descs, PairState *__restrict__ states, RunParams rp, mulls::IcpConst K,
											 const PairOut *__restrict__ out, mulls::StepState *__restrict__ steps, IcpOut *__restrict__ results, int brute,
											 uint32_t pair_base)
{
	__shared__ StepLds L;
	(void)step_pair(pair_base + blockIdx.x, descs, states, rp, K, out, steps, results, brute, L);
}

// Large batches: k_finish and k_step as ONE launch of one wave per pair — the wave sums its pair's trip partials itself (a KITTI-sized pair has four to eight of
// them per class: three sums per lane, every load in flight at once) and steps the pair.  (k_finish_step — a 256-lane workgroup per pair whose first wave steps —
// was slower than the two launches at 4096 pairs, profiles/r06_experiments.txt item 7: three of its four waves wait through the step.  Here nothing waits.)
__global__ __launch_bounds__(64) void k_sum_step(CloudDesc *__restrict__ descs, PairState *__restrict__ states, RunParams rp, mulls::IcpConst K,
												 const double *__restrict__ partial, PairOut *__restrict__ out, const uint32_t *__restrict__ bbox,
												 mulls::StepState *__restrict__ steps, IcpOut *__restrict__ results, int brute, uint32_t pair_base)
{
	__shared__ StepLds L;
	__shared__ FinishLds F;
	const uint32_t pair = pair_base + blockIdx.x;
	const bool live = states[pair].active || states[pair].want_residual; // uniform
	unsigned long long sw[STEP_WORDS_PER_LANE];
	if (live)
	{
		const unsigned long long *src = reinterpret_cast<const unsigned long long *>(&steps[pair]);
#pragma unroll
		for (uint32_t k = 0; k < STEP_WORDS_PER_LANE; k++)
			sw[k] = src[min(threadIdx.x + 64u * k, STEP_WORDS - 1u)];
		finish_pair_wave(descs + pair * MULLS_NC, states[pair], rp, partial, out[pair], bbox + pair * 6, F);
	}
	__syncthreads(); // F is complete
	if (threadIdx.x < MULLS_NTERM_PAD)
		L.comb[threadIdx.x] = F.comb[threadIdx.x];
	if (threadIdx.x < 30u)
		L.cnt[threadIdx.x] = F.cnt[threadIdx.x];
	(void)step_pair(pair, descs, states, rp, K, out, steps, results, brute, L, sw);
}

// Small batches (launch_finish_step decides): k_finish, k_step and k_step_publish as ONE launch — a workgroup sums its pair's partials, its
// first wave steps the pair, and the last workgroup to arrive publishes the word the host reads.  With a few hundred workgroups the arrival
// ticket (one device-scope fence and two atomics per workgroup) costs less than the two launches it saves; with thousands it does not
// (110 us at 4096 pairs), so large batches keep the three kernels.
__global__ __launch_bounds__(MULLS_BLOCK) void k_finish_step(CloudDesc *__restrict__ descs, PairState *__restrict__ states, RunParams rp, mulls::IcpConst K,
															  const double *__restrict__ partial, PairOut *__restrict__ out, const uint32_t *__restrict__ bbox,
															  mulls::StepState *__restrict__ steps, IcpOut *__restrict__ results, int brute, uint32_t *__restrict__ ticket,
															  volatile unsigned long long *host_word, uint32_t epoch, uint32_t pair_base)
{
	__shared__ StepLds L;
	const uint32_t pair = pair_base + blockIdx.x;
	const bool live = states[pair].active || states[pair].want_residual; // uniform per workgroup
	// the pair's StepState is requested before the partial sums: the step below finds it in registers instead of starting another round trip
	unsigned long long sw[STEP_WORDS_PER_LANE];
	if (live && threadIdx.x < 64u)
	{
		const unsigned long long *src = reinterpret_cast<const unsigned long long *>(&steps[pair]);
#pragma unroll
		for (uint32_t k = 0; k < STEP_WORDS_PER_LANE; k++)
			sw[k] = src[min(threadIdx.x + 64u * k, STEP_WORDS - 1u)];
	}
	__shared__ FinishLds F;
	if (live)
		finish_pair(descs + pair * MULLS_NC, states[pair], rp, partial, out[pair], bbox + pair * 6, &F);
	__syncthreads(); // F is complete
	if (threadIdx.x < MULLS_NTERM_PAD)
		L.comb[threadIdx.x] = F.comb[threadIdx.x];
	else if (threadIdx.x >= 64u && threadIdx.x < 94u)
		L.cnt[threadIdx.x - 64u] = F.cnt[threadIdx.x - 64u];
	const bool left = step_pair(pair, descs, states, rp, K, out, steps, results, brute, L, sw);
	if (threadIdx.x == 0)
	{
		// arrivals (low half) and pairs still iterating (high half) in ONE 64-bit atomic: no fence between two counters, none per workgroup — what the other
		// workgroups wrote is read by the next launch (a kernel boundary) or by the host after the stream has drained, never through this word
		unsigned long long *t64 = reinterpret_cast<unsigned long long *>(ticket);
		const unsigned long long old = atomicAdd(t64, ((unsigned long long)(left ? 1u : 0u) << 32) | 1ull);
		if ((uint32_t)old == gridDim.x - 1u)
		{
			const uint32_t n_left = (uint32_t)(old >> 32) + (left ? 1u : 0u);
			*t64 = 0ull; // re-armed for the next launch (stream order: nobody else touches it before)
			__threadfence_system();
			*host_word = (unsigned long long)epoch << 32 | n_left;
		}
	}
}

// ... and the word the host reads: pairs still iterating after this launch set.  Its own small launch: thousands of single-wave workgroups
// taking a ticket on one address cost more than the whole step (the atomics serialise in the L2).
__global__ __launch_bounds__(1024) void k_step_publish(const PairState *__restrict__ states, uint32_t npairs, volatile unsigned long long *host_word, uint32_t epoch)
{
	int mine = 0;
	for (uint32_t p = threadIdx.x; p < npairs; p += 1024u)
		mine += (states[p].active || states[p].want_residual) ? 1 : 0;
	__shared__ uint32_t s_left;
	if (threadIdx.x == 0)
		s_left = 0u;
	__syncthreads();
	for (int off = 32; off > 0; off >>= 1)
		mine += __shfl_down(mine, off);
	if ((threadIdx.x & 63u) == 0u && mine)
		atomicAdd(&s_left, (uint32_t)mine);
	__syncthreads();
	if (threadIdx.x == 0)
	{
		__threadfence_system();
		*host_word = (unsigned long long)epoch << 32 | s_left;
	}
}

// ---------------------------------------------------------------------------------------------------------------
// Per-iteration input: the host writes the PairState records into pinned memory; this kernel pulls them into HBM with
// coalesced 16-B reads over PCIe (one read of the block instead of one per workgroup of every later kernel).
static_assert(sizeof(PairState) % 16 == 0, "PairState must be a whole number of uint4 words");
__global__ void k_push_states(const uint4 *__restrict__ host_words, uint4 *__restrict__ dev_words, uint32_t nwords)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < nwords)
		dev_words[i] = host_words[i];
}

// ---------------------------------------------------------------------------------------------------------------
// Stage-level helpers for the parity tests (mulls_stage_* in include/mulls_hip.h)
__global__ void k_transform_aos(float4 *__restrict__ recs, uint32_t n, const double *__restrict__ T /* 12 row-major */)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	float4 a = recs[(size_t)i * 3], b = recs[(size_t)i * 3 + 1];
	double x = a.x, y = a.y, z = a.z, nx = b.x, ny = b.y, nz = b.z;
	a.x = (float)(T[0] * x + T[1] * y + T[2] * z + T[3]);
	a.y = (float)(T[4] * x + T[5] * y + T[6] * z + T[7]);
	a.z = (float)(T[8] * x + T[9] * y + T[10] * z + T[11]);
	b.x = (float)(T[0] * nx + T[1] * ny + T[2] * nz);
	b.y = (float)(T[4] * nx + T[5] * ny + T[6] * nz);
	b.z = (float)(T[8] * nx + T[9] * ny + T[10] * nz);
	recs[(size_t)i * 3] = a;
	recs[(size_t)i * 3 + 1] = b;
}

// ... on up to six clouds in one launch (blockIdx.y), the transform travelling with the launch: the local map's update moves the frame's clouds and then the
// whole map (map.cpp) — eleven launches and two uploads otherwise
struct TransformArgs
{
	float4 *recs[6];
	uint32_t n[6];
	double T[12];
};
__global__ void k_transform_clouds(TransformArgs A)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= A.n[blockIdx.y])
		return;
	float4 *__restrict__ recs = A.recs[blockIdx.y];
	const double *T = A.T;
	float4 a = recs[(size_t)i * 3], b = recs[(size_t)i * 3 + 1];
	double x = a.x, y = a.y, z = a.z, nx = b.x, ny = b.y, nz = b.z;
	a.x = (float)(T[0] * x + T[1] * y + T[2] * z + T[3]);
	a.y = (float)(T[4] * x + T[5] * y + T[6] * z + T[7]);
	a.z = (float)(T[8] * x + T[9] * y + T[10] * z + T[11]);
	b.x = (float)(T[0] * nx + T[1] * ny + T[2] * nz);
	b.y = (float)(T[4] * nx + T[5] * ny + T[6] * nz);
	b.z = (float)(T[8] * nx + T[9] * ny + T[10] * nz);
	recs[(size_t)i * 3] = a;
	recs[(size_t)i * 3 + 1] = b;
}

// force a given correspondence list into the flag/match/wd arrays (mulls_stage_accumulate)
__global__ void k_set_corr(uint32_t src_off, const int32_t *__restrict__ cs, const int32_t *__restrict__ ct, const float *__restrict__ cd,
						   uint32_t n, uint8_t *__restrict__ flag, int32_t *__restrict__ match, float *__restrict__ wd, uint32_t tgt_off,
						   const float4 *__restrict__ tpos, const float4 *__restrict__ tnrm, float4 *__restrict__ mq)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	const uint32_t g = src_off + (uint32_t)cs[i];
	flag[g] = MULLS_F_ALIVE | MULLS_F_VALID;
	match[g] = ct[i];
	mq[2u * g] = tpos[tgt_off + (uint32_t)ct[i]];
	mq[2u * g + 1u] = tnrm[tgt_off + (uint32_t)ct[i]];
	wd[g] = cd ? cd[i] : 0.0f;
}

// ---------------------------------------------------------------------------------------------------------------
// host-callable launch wrappers (the driver is plain C++ and never sees <<< >>>)
#include "launch.h"

void launch_accum(hipStream_t st, const uint32_t *leaders, const uint32_t split[4], const Job *jobs, const CloudDesc *descs, const PairState *states, const RunParams &rp,
				  const float4 *spos, const float4 *mq, const uint8_t *flag, float *wd, double *partial, bool single, uint32_t wave_min_trips)
{
	(void)dev_launch<3>([](DevLaunch &) { // per device (launch.h)
		return hipFuncSetAttribute(reinterpret_cast<const void *>(k_accum<1024, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)MULLS_RED_BYTES_HALF) == hipSuccess;
	});
	// One wave per trip once the launch has trips enough to fill the chip with waves (k_accum_wave; below that a trip's sixteen slots per lane in sequence are the launch's
	// latency).  Its point-to-line form is the faithful combined system's (diagonal + right-hand side); other point-to-line sums stay with k_accum.
	const uint32_t n_trips = split[3] - split[0];
	if (wave_min_trips && n_trips >= wave_min_trips && rp.faithful && rp.pull_comb)
	{
		hipLaunchKernelGGL(k_accum_wave, dim3((n_trips + MULLS_ACCW_WAVES / MULLS_ACCW_SPLIT - 1u) / (MULLS_ACCW_WAVES / MULLS_ACCW_SPLIT)), dim3(64 * MULLS_ACCW_WAVES), 0, st, leaders + split[0], n_trips, jobs, descs, states, rp, spos,
						   mq, flag, wd, partial);
		return;
	}
	if (single)
	{
		// small batches: one launch for every trip length (a kernel of a few hundred workgroups takes ~5 us whatever it does: two launches saved
		// are worth more than the short trips' better occupancy)
		if (split[3] > split[0])
			hipLaunchKernelGGL((k_accum<1024, 2>), dim3(split[3] - split[0]), dim3(1024), MULLS_RED_BYTES_HALF, st, leaders + split[0], jobs, descs, states, rp, spos, mq,
							   flag, wd, partial);
		return;
	}
	// leaders[split[0] .. split[1]): trips of more than 512 slots; [split[1] .. split[2]): 257..512; [split[2] .. split[3]): up to 256
	if (split[1] > split[0])
		hipLaunchKernelGGL((k_accum<1024, 2>), dim3(split[1] - split[0]), dim3(1024), MULLS_RED_BYTES_HALF, st, leaders + split[0], jobs, descs, states, rp, spos, mq, flag,
						   wd, partial);
	if (split[2] > split[1])
		hipLaunchKernelGGL((k_accum<512, 4>), dim3(split[2] - split[1]), dim3(512), MULLS_RED_BYTES_HALF / 2, st, leaders + split[1], jobs, descs, states, rp, spos, mq, flag,
						   wd, partial);
	if (split[3] > split[2])
		hipLaunchKernelGGL((k_accum<256, 8>), dim3(split[3] - split[2]), dim3(256), MULLS_RED_BYTES_HALF / 4, st, leaders + split[2], jobs, descs, states, rp, spos, mq, flag,
						   wd, partial);
}

void launch_finish(hipStream_t st, uint32_t npairs, CloudDesc *descs, const PairState *states, const RunParams &rp, const double *partial,
				   PairOut *out, PairOut *out_host, const uint32_t *bbox, uint32_t *ticket, volatile uint32_t *host_epoch, uint32_t epoch,
				   uint32_t pair_base)
{
	if (!npairs)
		return;
	const bool direct = rp.pull_comb && npairs <= 64u; // launch-bound iterations: k_finish ships the records itself
	hipLaunchKernelGGL(k_finish, dim3(npairs), dim3(MULLS_BLOCK), 0, st, descs, states, rp, partial, out, bbox, pair_base,
					   direct ? reinterpret_cast<uint4 *>(out_host) : nullptr, ticket, host_epoch, epoch);
	if (direct)
		return;
	uint32_t n_used = 0;
	for (int c = 0; c < MULLS_NC; c++)
		n_used += rp.used[c] ? 1u : 0u;
	const uint32_t nwords = npairs * (8u + (MULLS_NTERM_PAD / 2u) * (rp.pull_comb ? 1u : n_used));
	hipLaunchKernelGGL(k_pull_outs, dim3((nwords + MULLS_BLOCK - 1) / MULLS_BLOCK), dim3(MULLS_BLOCK), 0, st, reinterpret_cast<const uint4 *>(out),
					   reinterpret_cast<uint4 *>(out_host), rp, pair_base, npairs, ticket, host_epoch, epoch);
}

void launch_step_init(hipStream_t st, uint32_t npairs, const PairSetup *setup, const mulls::IcpConst &K, mulls::StepState *steps, PairState *states)
{
	if (npairs)
		hipLaunchKernelGGL(k_step_init, dim3((npairs + 63u) / 64u), dim3(64), 0, st, npairs, setup, K, steps, states);
}

void launch_finish_step(hipStream_t st, uint32_t pair_base, uint32_t npairs, CloudDesc *descs, PairState *states, const RunParams &rp, const mulls::IcpConst &K, const double *partial,
						PairOut *out, const uint32_t *bbox, mulls::StepState *steps, IcpOut *results, unsigned long long *host_word, uint32_t epoch, int brute,
						uint32_t *ticket, bool sum_step)
{
	if (!npairs)
		return;
	if (ticket) // small batch: one launch
	{
		hipLaunchKernelGGL(k_finish_step, dim3(npairs), dim3(MULLS_BLOCK), 0, st, descs, states, rp, K, partial, out, bbox, steps, results, brute, ticket, host_word, epoch,
						   pair_base);
		return;
	}
	if (sum_step && rp.pull_comb)
		hipLaunchKernelGGL(k_sum_step, dim3(npairs), dim3(64), 0, st, descs, states, rp, K, partial, out, bbox, steps, results, brute, pair_base);
	else
	{
		hipLaunchKernelGGL(k_finish, dim3(npairs), dim3(MULLS_BLOCK), 0, st, descs, states, rp, partial, out, bbox, pair_base, static_cast<uint4 *>(nullptr),
						   static_cast<uint32_t *>(nullptr), static_cast<volatile uint32_t *>(nullptr), 0u);
		hipLaunchKernelGGL(k_step, dim3(npairs), dim3(64), 0, st, descs, states, rp, K, out, steps, results, brute, pair_base);
	}
	hipLaunchKernelGGL(k_step_publish, dim3(1), dim3(1024), 0, st, states + pair_base, npairs, host_word, epoch);
}

void launch_push_states(hipStream_t st, const PairState *host_states, PairState *dev_states, uint32_t npairs)
{
	const uint32_t nwords = npairs * (uint32_t)(sizeof(PairState) / 16);
	if (nwords)
		hipLaunchKernelGGL(k_push_states, dim3((nwords + 255) / 256), dim3(256), 0, st, reinterpret_cast<const uint4 *>(host_states),
						   reinterpret_cast<uint4 *>(dev_states), nwords);
}

void launch_transform_clouds(hipStream_t st, float4 *const recs[], const uint32_t n[], int count, const double T12[12])
{
	TransformArgs a;
	uint32_t most = 0;
	for (int c = 0; c < 6; c++)
	{
		a.recs[c] = c < count ? recs[c] : nullptr;
		a.n[c] = c < count ? n[c] : 0u;
		most = a.n[c] > most ? a.n[c] : most;
	}
	for (int k = 0; k < 12; k++)
		a.T[k] = T12[k];
	if (most)
		hipLaunchKernelGGL(k_transform_clouds, dim3((most + 255) / 256, 6), dim3(256), 0, st, a);
}

void launch_transform_aos(hipStream_t st, float4 *recs, uint32_t n, const double *T12)
{
	if (n)
		hipLaunchKernelGGL(k_transform_aos, dim3((n + 255) / 256), dim3(256), 0, st, recs, n, T12);
}

void launch_set_corr(hipStream_t st, uint32_t src_off, const int32_t *cs, const int32_t *ct, const float *cd, uint32_t n, uint8_t *flag,
					 int32_t *match, float *wd, uint32_t tgt_off, const float4 *tpos, const float4 *tnrm, float4 *mq)
{
	if (n)
		hipLaunchKernelGGL(k_set_corr, dim3((n + 255) / 256), dim3(256), 0, st, src_off, cs, ct, cd, n, flag, match, wd, tgt_off, tpos, tnrm, mq);
}
