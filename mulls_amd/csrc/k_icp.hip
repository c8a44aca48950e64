// k_icp.hip — the device-resident registration loop (gfx950 / CDNA4, wave64): ONE launch runs every ICP iteration of every
// pair of a batch.  A persistent 1024-lane workgroup per CU takes pairs from a queue (most expensive first) and carries each
// through mm_lls_icp's loop (cregistration.hpp:1239-1401) on its own:
//
//   per iteration   for every used class cloud: cert_class (rigid step, certificates, a handful of leftover queries against the
//                   grid in global memory) or, when many points need a search, lds_search_class (target class cloud staged in
//                   LDS) — lds_tier.h, the same device code the lock-step kernels k_cert / k_nn_lds run;
//                   counters rolled over, correspondence-count test, threshold update (icp_step.h: step_counts);
//                   normal equations: accum_point per valid correspondence, fixed-order reduction per class (accum.h), the class
//                   rows combined as cregistration.hpp:1914-1938 does;
//                   6x6 solve, Euler step, step-size and convergence tests (icp_step.h: step_solve — the very functions the
//                   host driver runs for the lock-step path, built on detmath.h: same bits on host and device);
//   at the end      posterior residual pass, sigma, information matrix; one result record per pair.
//
// No host round trip, no launches and no HBM round trip of the per-pair state between iterations: a pair's working set (source
// clouds, correspondence records, hints: ~130 B per source point) stays in the L2 / Infinity Cache of the CU that iterates it.
// Results are bit-identical to the lock-step path (tests/test_gpu_icp.py::test_resident_loop_equals_lock_step).
#include "../../include/mulls_hip.h"
#include "accum.h"
#include "icp_step.h"
#include "solve_wave.h"
#include "lds_tier.h"

#define MULLS_ICP_BLOCK MULLS_LDS_BLOCK
static_assert(MULLS_ICP_BLOCK == MULLS_ACC_LANES, "one source point per lane and trip, in the reduction's lane order");

// ---------------------------------------------------------------------------------------------------------------
// Fused pass of one class cloud's iteration — the common case of the loop: the class is searched this iteration, its source
// cloud has at most MULLS_FUSE_TRIPS x 1024 points and at most MULLS_CERT_SMALL of them fail their certificate.  A lane keeps its
// points (one per trip) in registers from the rigid step to the normal-equation terms: one round trip to memory per point and
// iteration instead of one per stage.  Stage by stage the arithmetic is that of cert_class, class_tail / filter_point and
// trip_sum (the lock-step path's k_cert, k_filter, k_accum): same operations, same summation order, same bits.
//   1  rigid step + certificate (cert_class)                 -> duplicate table, the few leftover queries
//   2  leftovers against the grid in global memory, commit   -> nn_idx / nn_d2 / hint of those points
//   3  duplicate rule + rejection chain (filter_point)       -> flags, match, records, class counters
//   4  terms of the valid correspondences, trip by trip      -> row[0..26]   (skipped when `with_row` is false: the class weight of
//                                                               this class waits for another class's count)
// Returns 0: done, row summed; 1: done, row still to be summed from memory (class_row); 2: too many leftovers — the caller runs
// lds_search_class (nn_idx / nn_d2 of every point are in memory as cert_class leaves them) and sums the row from memory.
#define MULLS_FUSE_TRIPS 2
__device__ __forceinline__ int fused_class(const RunParams &rp, const PairState &ps, const Job &job, CloudDesc *pd, const GridDesc &g, const LdsLayout &Y,
										   unsigned char *lds_raw, bool with_row, float4 *__restrict__ spos, float4 *__restrict__ snrm,
										   const uint32_t *__restrict__ cell_start, const float4 *__restrict__ tsorted, uint8_t *flag, int32_t *__restrict__ nn_idx,
										   float *__restrict__ nn_d2, unsigned long long *__restrict__ winner, const float4 *__restrict__ tnrm,
										   int32_t *__restrict__ match, float *__restrict__ wd, const float4 *__restrict__ tpos, int32_t *__restrict__ nn_hint,
										   float4 *__restrict__ mq, double *row, unsigned long long *tf)
{
#define FST(k)                                       \
	do                                               \
	{                                                \
		if (threadIdx.x == 0)                        \
		{                                            \
			const unsigned long long now_ = wall_clock64(); \
			tf[k] += now_ - tf[6];                   \
			tf[6] = now_;                            \
		}                                            \
	} while (0)
	if (threadIdx.x == 0)
		tf[6] = wall_clock64();

	__shared__ float4 uq[MULLS_CERT_SMALL];
	__shared__ uint32_t us[MULLS_CERT_SMALL];
	__shared__ uint32_t ucount, s_matched, s_alive, s_valid;
	__shared__ double part[MULLS_NTERM_PAD];
	int2 *__restrict__ hint2 = reinterpret_cast<int2 *>(nn_hint);
	uint32_t *W = Y.W;
	const int cls = (int)job.cls;
	CloudDesc &d = pd[cls];
	const uint32_t src_n = d.src_n, tgt_n = d.tgt_n;
	const ClassCtx C = class_ctx(rp, ps, g, cls, d.alive_cur, true);
	const bool have_prev = ps.iter > 0; // hint records of this run exist from its second iteration on
	if (C.dedup)
		for (uint32_t t = threadIdx.x; t < tgt_n; t += MULLS_ICP_BLOCK)
			W[t] = 0xffffffffu;
	if (threadIdx.x == 0)
		ucount = s_matched = s_alive = s_valid = 0u;
	__syncthreads();
	FST(0);

	// ---- stage 1 ------------------------------------------------------------------------------------------------------------------
	uint32_t F[MULLS_FUSE_TRIPS];	 // flag byte as loaded; 0 for a slot beyond the cloud
	float4 Pn[MULLS_FUSE_TRIPS];	 // the point after this iteration's rigid step (w = intensity)
	float4 Nn[MULLS_FUSE_TRIPS];	 // ... its normal / direction
	float4 Q0[MULLS_FUSE_TRIPS], Q1[MULLS_FUSE_TRIPS]; // the point's correspondence record (matched target position, direction)
	int32_t M[MULLS_FUSE_TRIPS], PM[MULLS_FUSE_TRIPS];	// nearest target of this iteration (-1 none, MULLS_NEEDS_SEARCH), match[] as loaded
	float D0[MULLS_FUSE_TRIPS];							// its squared distance (or the sweep radius of a leftover query)
	uint32_t matched_cnt = 0;
#pragma unroll
	for (int k = 0; k < MULLS_FUSE_TRIPS; k++)
	{
		const uint32_t s = threadIdx.x + (uint32_t)k * MULLS_ICP_BLOCK;
		F[k] = 0u;
		M[k] = -1;
		PM[k] = -1;
		D0[k] = 0.0f;
		Pn[k] = Nn[k] = Q0[k] = Q1[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		if (s >= src_n)
			continue;
		const uint32_t gi = d.src_off + s;
		F[k] = flag[gi];
		if (!(F[k] & MULLS_F_ALIVE))
			continue;
		const float4 p = spos[gi], n = snrm[gi];
		const int2 h = hint2[gi];
		PM[k] = match[gi];
		Q0[k] = mq[2u * gi];
		Q1[k] = mq[2u * gi + 1u];
		const uint32_t hv = have_prev ? (uint32_t)h.x : 0xffffu;
		const float lb = have_prev ? __int_as_float(h.y) : 0.0f;
		const int32_t pm = have_prev ? PM[k] : -1;
		// the hinted target's position: for a point whose hint is its standing correspondence it sits in the point's own record
		const uint32_t hj = hv & 0xffffu;
		float4 tj = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		if (hj < tgt_n)
			tj = (int32_t)hj == pm ? Q0[k] : tgt_point(rp.tgt_stage, rp.tgt_map, d, hj, tpos);
		// rigid step (cregistration.hpp:1690-1695): double math, float store, in place
		const double *T = ps.T;
		const double x = p.x, y = p.y, z = p.z, nx = n.x, ny = n.y, nz = n.z;
		float4 out;
		out.x = (float)(T[0] * x + T[1] * y + T[2] * z + T[3]);
		out.y = (float)(T[4] * x + T[5] * y + T[6] * z + T[7]);
		out.z = (float)(T[8] * x + T[9] * y + T[10] * z + T[11]);
		const float onx = (float)(T[0] * nx + T[1] * ny + T[2] * nz);
		const float ony = (float)(T[4] * nx + T[5] * ny + T[6] * nz);
		const float onz = (float)(T[8] * nx + T[9] * ny + T[10] * nz);
		Pn[k] = make_float4(out.x, out.y, out.z, p.w);
		Nn[k] = make_float4(onx, ony, onz, n.w);
		spos[gi] = Pn[k];
		snrm[gi] = Nn[k];
		const float mx = out.x - p.x, my = out.y - p.y, mz = out.z - p.z;
		const float moved = sqrtf((mx * mx + my * my) + mz * mz);
		const float lb_next = lb - moved * 1.00001f;
		out.w = __builtin_inff(); // sweep radius of a search: +inf = no hint
		bool certified = false;
		if (hj < tgt_n)
		{
			const float dx = out.x - tj.x, dy = out.y - tj.y, dz = out.z - tj.z;
			const float d0 = (dx * dx + dy * dy) + dz * dz; // the very expression a search evaluates for this candidate
			if (d0 >= 0.0f)
			{
				const float dh = sqrtf(d0);
				certified = rp.cert != 0u && (dh * 1.00001f + moved * 1.00001f < lb * 0.99999f); // NaN anywhere fails the test
				out.w = dh + fminf(fmaxf(rp.cert_slack_rate * moved, rp.cert_slack_min), rp.cert_slack_max);
				if (certified)
				{
					const bool matched = !((double)d0 > C.max_dist_sqr);
					M[k] = matched ? (int32_t)hj : -1;
					D0[k] = d0;
					hint2[gi] = make_int2((int32_t)hj, __float_as_int(lb_next)); // cost class 0
					if (matched)
					{
						matched_cnt++;
						if (C.dedup)
							atomicMin(&W[hj], s);
						else if (C.gate)
							atomicMin(&winner[d.tgt_off + hj], C.key_hi | (unsigned long long)s);
					}
				}
			}
		}
		if (!certified)
		{
			M[k] = MULLS_NEEDS_SEARCH;
			D0[k] = out.w;
			const uint32_t q = atomicAdd(&ucount, 1u);
			if (q < MULLS_CERT_SMALL)
			{
				uq[q] = out;
				us[q] = s;
			}
		}
	}
	__syncthreads();
	FST(1);
	const uint32_t U = ucount;
	if (U > MULLS_CERT_SMALL)
	{
		// too many for the global-memory walk: leave nn_idx / nn_d2 as cert_class does and let the caller stage the target cloud
#pragma unroll
		for (int k = 0; k < MULLS_FUSE_TRIPS; k++)
		{
			const uint32_t s = threadIdx.x + (uint32_t)k * MULLS_ICP_BLOCK;
			if (s < src_n && (F[k] & MULLS_F_ALIVE))
			{
				nn_idx[d.src_off + s] = M[k];
				nn_d2[d.src_off + s] = D0[k];
			}
		}
		__threadfence_block();
		return 2;
	}
	// ---- stage 2: the few leftovers against the grid where k_grid_build_sort left it (L2-resident): same sweeps, same keys ------------
	if (U)
	{
		const GlobGrid L = {tsorted + d.tgt_off, reinterpret_cast<const uint16_t *>(cell_start) + g.cell_off};
		const uint32_t sub = threadIdx.x & (MULLS_LDS_GROUP - 1u), grp = threadIdx.x / MULLS_LDS_GROUP;
		for (uint32_t i = grp; i < U; i += MULLS_ICP_BLOCK / MULLS_LDS_GROUP)
		{
			nnkey bk;
			float sec, Rfin;
			uint32_t trips;
			CandOut co = {0xffffffffu, 0xffffffffu, 0.0f};
			search_query(g, L, uq[i], C.r, C.m, sub, bk, sec, Rfin, trips, co);
			if (sub == 0 && commit_search(C, d, us[i], bk, sec, Rfin, trips, co, nn_idx, nn_d2, hint2, W, winner))
				matched_cnt++;
		}
	}
	for (int off = 32; off > 0; off >>= 1)
		matched_cnt += __shfl_down(matched_cnt, off);
	if ((threadIdx.x & 63) == 0 && matched_cnt)
		atomicAdd(&s_matched, matched_cnt);
	__threadfence_block(); // the leftovers' nn_idx / nn_d2, read back below by the lanes that own the points
	__syncthreads();
	FST(2);
	// ---- stage 3: duplicate rule (cregistration.hpp:1762-1789) and rejection chain (:1794-1830) ---------------------------------------
	const uint32_t total_matched = s_matched;
	const float thr = ps.thr[cls];
	const float max_sqr = thr * thr; // CorrespondenceRejectorDistance::setMaximumDistance (float)
	const bool any_match = total_matched > 0u, normal_check = cls != 5; // vertex correspondences skip the direction check (:1292)
	bool V[MULLS_FUSE_TRIPS];
	float WD[MULLS_FUSE_TRIPS];
	uint32_t n_alive = 0, n_valid = 0;
#pragma unroll
	for (int k = 0; k < MULLS_FUSE_TRIPS; k++)
	{
		const uint32_t s = threadIdx.x + (uint32_t)k * MULLS_ICP_BLOCK;
		V[k] = false;
		WD[k] = 0.0f;
		if (s >= src_n || !(F[k] & MULLS_F_ALIVE))
			continue;
		const uint32_t gi = d.src_off + s;
		int32_t m = M[k];
		float dist = D0[k];
		if (m == MULLS_NEEDS_SEARCH)
		{
			m = nn_idx[gi];
			dist = nn_d2[gi];
		}
		if (C.dedup && m >= 0 && W[m] != s)
			m = -1; // first source (lowest index) matched to a target keeps it; the others become unmatched
		bool alive = true, valid, fresh = false;
		float4 n2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		if (any_match)
		{
			valid = m >= 0;
			if (C.gate && m < 0)
			{
				alive = false; // unmatched or duplicate-losing source points vanish for good (:1762-1789)
				valid = false;
			}
			if (valid)
			{
				valid = rp.rej_strict ? dist < max_sqr : !(dist > max_sqr); // CorrespondenceRejectorDistance (see mulls_params.rejector_strict)
				if (valid)
				{
					WD[k] = dist; // pcl::Correspondence::distance (shares storage with ::weight)
					if (PM[k] == m)
						n2 = Q1[k];
					else
					{
						match[gi] = m;
						tgt_record(rp.tgt_stage, rp.tgt_map, d, (uint32_t)m, tpos, tnrm, Q0[k], n2);
						Q1[k] = n2;
						mq[2u * gi] = Q0[k];
						mq[2u * gi + 1u] = n2;
					}
					fresh = true;
				}
			}
		}
		else if (C.gate)
		{
			alive = false; // the whole cloud was swapped for an empty one; reference behaviour undefined, see oracle
			valid = false;
		}
		else
			valid = (F[k] & MULLS_F_VALID) != 0; // the previous Corr_f is still in place (SURVEY B-4) and goes through the direction check again
		if (valid && normal_check)
		{
			if (!fresh)
				n2 = Q1[k]; // the standing correspondence's target direction
			const float4 n1 = Nn[k];
			const double dot = (double)n1.x * (double)n2.x + (double)n1.y * (double)n2.y + (double)n1.z * (double)n2.z;
			const float c = (float)fabs(dot);
			if ((double)c < rp.cos_bearing)
				valid = false;
		}
		const uint32_t nf = (alive ? MULLS_F_ALIVE : 0u) | (valid ? MULLS_F_VALID : 0u);
		if (nf != F[k])
			flag[gi] = (uint8_t)nf;
		if (fresh)
			wd[gi] = WD[k];
		else if (valid)
			WD[k] = wd[gi]; // stale correspondence: its distance / weight word as it stands
		V[k] = alive && valid;
		n_alive += alive ? 1u : 0u;
		n_valid += valid ? 1u : 0u;
	}
	for (int off = 32; off > 0; off >>= 1)
	{
		n_alive += __shfl_down(n_alive, off);
		n_valid += __shfl_down(n_valid, off);
	}
	if ((threadIdx.x & 63) == 0)
	{
		if (n_alive)
			atomicAdd(&s_alive, n_alive);
		if (n_valid)
			atomicAdd(&s_valid, n_valid);
	}
	__syncthreads();
	if (threadIdx.x == 0)
	{
		d.n_matched = total_matched;
		d.alive_next = s_alive;
		d.valid_next = s_valid;
		d.n_search = U;
	}
	FST(3);
	if (!with_row)
		return 1;
	// ---- stage 4: this class's row of the normal equations, from the registers ------------------------------------------------------
	int cnt[MULLS_NC];
	for (int c = 0; c < MULLS_NC; c++)
		cnt[c] = c == cls ? (int)s_valid : (int)(class_called(rp, pd[c], c) ? pd[c].valid_next : pd[c].n_valid);
	const AccumCtx A = accum_ctx(rp, cls, ps.iter, false, class_weight(rp, cls, false, cnt));
#pragma unroll
	for (int k = 0; k < MULLS_FUSE_TRIPS; k++)
	{
		if ((uint32_t)k * MULLS_ICP_BLOCK >= src_n && k > 0)
			break;
		float w = WD[k];
		trip_sum_regs(A, ps.x, V[k], Pn[k], Q0[k], Q1[k], w, lds_raw, part);
		if (V[k] && __float_as_uint(w) != __float_as_uint(WD[k]))
			wd[d.src_off + threadIdx.x + (uint32_t)k * MULLS_ICP_BLOCK] = w; // pcl::Correspondence::weight
		if (threadIdx.x < MULLS_NTERM)
			row[threadIdx.x] = k ? row[threadIdx.x] + part[threadIdx.x] : 0.0 + part[threadIdx.x];
		__syncthreads();
	}
	FST(4);
	return 0;
#undef FST
}

// The duplicate table of a class that fused_all hands to lds_search_class, rebuilt at Y.W from what stage 1 left in memory (its own copy
// lies where the target cloud is about to be staged): the certified, matched points are the ones with nn_idx >= 0.  Every lane calls.
__device__ __forceinline__ void rebuild_dup_table(uint32_t *W, const CloudDesc &d, const uint8_t *flag, const int32_t *__restrict__ nn_idx)
{
	for (uint32_t t = threadIdx.x; t < d.tgt_n; t += MULLS_ICP_BLOCK)
		W[t] = 0xffffffffu;
	__syncthreads();
	for (uint32_t s = threadIdx.x; s < d.src_n; s += MULLS_ICP_BLOCK)
	{
		const uint32_t gi = d.src_off + s;
		if (!(flag[gi] & MULLS_F_ALIVE))
			continue;
		const int32_t m = nn_idx[gi];
		if (m >= 0)
			atomicMin(&W[m], s);
	}
	__syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// The fused pass over ALL class clouds of the pair at once (stages 1-3 of fused_class, the same arithmetic point by point): the source
// points of the participating classes are numbered through, a lane takes the points f = lane, lane + 1024, ... of that numbering, and
// every stage runs once per iteration instead of once per class — a class cloud of a few hundred points no longer costs a full round of
// barriers and memory latencies of its own.  Per class: its own duplicate table (all tables side by side in the LDS the staged cloud
// would use), its own leftover count; the leftovers of all classes share one queue.  A class with more than MULLS_FA_SMALL leftovers
// drops out after stage 1 (its nn_idx / nn_d2 in memory as cert_class leaves them): the caller rebuilds its duplicate table at Y.W and
// runs lds_search_class.  Returns the mask of those classes.  Rows: summed afterwards from memory (class_row).
#define MULLS_FA_TRIPS 3
#define MULLS_FA_SMALL 256u // leftovers per class that fused_all searches itself (against the grids in global memory)
#define MULLS_FA_QCAP (MULLS_NC * MULLS_FA_SMALL)
struct FaClass
{
	ClassCtx C;
	uint32_t base, src_n, src_off, tgt_n, tgt_off, w_off, cls;
	float thr;
	uint32_t ucount, matched, alive, valid;
};
__device__ __forceinline__ uint32_t fused_all(const RunParams &rp, const PairState &ps, const Job *jobs, uint32_t njobs, uint32_t part_mask, CloudDesc *pd,
											   const GridDesc *grids, float4 *uq, uint32_t *us, uint32_t *Wall, FaClass *FC, uint32_t *s_q, float4 *__restrict__ spos,
											   float4 *__restrict__ snrm, const uint32_t *__restrict__ cell_start, const float4 *__restrict__ tsorted, uint8_t *flag,
											   int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2, unsigned long long *__restrict__ winner,
											   const float4 *__restrict__ tnrm, int32_t *__restrict__ match, float *__restrict__ wd, const float4 *__restrict__ tpos,
											   int32_t *__restrict__ nn_hint, float4 *__restrict__ mq, unsigned long long *tf)
{
#define FST(k)                                       \
	do                                               \
	{                                                \
		if (threadIdx.x == 0)                        \
		{                                            \
			const unsigned long long now_ = wall_clock64(); \
			tf[k] += now_ - tf[6];                   \
			tf[6] = now_;                            \
		}                                            \
	} while (0)
	if (threadIdx.x == 0)
		tf[6] = wall_clock64();
	int2 *__restrict__ hint2 = reinterpret_cast<int2 *>(nn_hint);
	// uq [MULLS_FA_QCAP]: leftover queries; us: their source index | class slot << 24
	const bool have_prev = ps.iter > 0;
	// class table (lane 0), duplicate tables, counters
	if (threadIdx.x == 0)
	{
		uint32_t base = 0, w_off = 0, k = 0;
		for (uint32_t j = 0; j < njobs; j++)
		{
			const int cls = (int)jobs[j].cls;
			if (!(part_mask >> cls & 1u))
				continue;
			FaClass &F = FC[k++];
			const CloudDesc &d = pd[cls];
			F.C = class_ctx(rp, ps, grids[cls], cls, d.alive_cur, true);
			F.base = base, F.src_n = d.src_n, F.src_off = d.src_off, F.tgt_n = d.tgt_n, F.tgt_off = d.tgt_off, F.cls = (uint32_t)cls;
			F.w_off = w_off;
			F.thr = ps.thr[cls];
			F.ucount = F.matched = F.alive = F.valid = 0u;
			base += d.src_n;
			if (F.C.dedup)
				w_off += d.tgt_n;
		}
		s_q[0] = 0u; // queue length
		s_q[1] = k;	 // classes
		s_q[2] = base;
		s_q[3] = w_off;
	}
	__syncthreads();
	const uint32_t ncls = s_q[1], total = s_q[2], wtot = s_q[3];
	for (uint32_t t = threadIdx.x; t < wtot; t += MULLS_ICP_BLOCK)
		Wall[t] = 0xffffffffu;
	__syncthreads();
	FST(0);

	// ---- stage 1: rigid step + certificate ---------------------------------------------------------------------------------------------
	uint32_t F[MULLS_FA_TRIPS], K[MULLS_FA_TRIPS]; // flag byte as loaded (0: no point); class slot of the point
	float4 Nn[MULLS_FA_TRIPS], Q1[MULLS_FA_TRIPS];
	int32_t M[MULLS_FA_TRIPS], PM[MULLS_FA_TRIPS];
	float D0[MULLS_FA_TRIPS];
	uint32_t S[MULLS_FA_TRIPS];
#pragma unroll
	for (int k = 0; k < MULLS_FA_TRIPS; k++)
	{
		const uint32_t f = threadIdx.x + (uint32_t)k * MULLS_ICP_BLOCK;
		F[k] = 0u, K[k] = 0u, M[k] = -1, PM[k] = -1, D0[k] = 0.0f, S[k] = 0u;
		Nn[k] = Q1[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		bool matched = false;
		uint32_t kc = 0;
		if (f < total)
		{
			while (kc + 1u < ncls && f >= FC[kc + 1u].base)
				kc++;
			const FaClass &A = FC[kc];
			const uint32_t s = f - A.base, gi = A.src_off + s, tgt_n = A.tgt_n;
			K[k] = kc, S[k] = s;
			// every record of the point is requested before the flag is looked at: one memory round trip, not two
			F[k] = flag[gi];
			const float4 p = spos[gi], n = snrm[gi];
			const int2 h = hint2[gi];
			const int32_t pm0 = match[gi];
			const float4 q0 = mq[2u * gi], q1 = mq[2u * gi + 1u];
			if (F[k] & MULLS_F_ALIVE)
			{
				PM[k] = pm0;
				Q1[k] = q1;
				const uint32_t hv = have_prev ? (uint32_t)h.x : 0xffffu;
				const float lb = have_prev ? __int_as_float(h.y) : 0.0f;
				const int32_t pm = have_prev ? PM[k] : -1;
				const uint32_t hj = hv & 0xffffu;
				float4 tj = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
				if (hj < tgt_n)
					tj = (int32_t)hj == pm ? q0 : tgt_point(rp.tgt_stage, rp.tgt_map, pd[A.cls], hj, tpos);
				// rigid step (cregistration.hpp:1690-1695): double math, float store, in place
				const double *T = ps.T;
				const double x = p.x, y = p.y, z = p.z, nx = n.x, ny = n.y, nz = n.z;
				float4 out;
				out.x = (float)(T[0] * x + T[1] * y + T[2] * z + T[3]);
				out.y = (float)(T[4] * x + T[5] * y + T[6] * z + T[7]);
				out.z = (float)(T[8] * x + T[9] * y + T[10] * z + T[11]);
				const float onx = (float)(T[0] * nx + T[1] * ny + T[2] * nz);
				const float ony = (float)(T[4] * nx + T[5] * ny + T[6] * nz);
				const float onz = (float)(T[8] * nx + T[9] * ny + T[10] * nz);
				Nn[k] = make_float4(onx, ony, onz, n.w);
				spos[gi] = make_float4(out.x, out.y, out.z, p.w);
				snrm[gi] = Nn[k];
				const float mx = out.x - p.x, my = out.y - p.y, mz = out.z - p.z;
				const float moved = sqrtf((mx * mx + my * my) + mz * mz);
				const float lb_next = lb - moved * 1.00001f;
				out.w = __builtin_inff(); // sweep radius of a search: +inf = no hint
				bool certified = false;
				if (hj < tgt_n)
				{
					const float dx = out.x - tj.x, dy = out.y - tj.y, dz = out.z - tj.z;
					const float d0 = (dx * dx + dy * dy) + dz * dz; // the very expression a search evaluates for this candidate
					if (d0 >= 0.0f)
					{
						const float dh = sqrtf(d0);
						certified = rp.cert != 0u && (dh * 1.00001f + moved * 1.00001f < lb * 0.99999f); // NaN anywhere fails the test
						out.w = dh + fminf(fmaxf(rp.cert_slack_rate * moved, rp.cert_slack_min), rp.cert_slack_max);
						if (certified)
						{
							matched = !((double)d0 > A.C.max_dist_sqr);
							M[k] = matched ? (int32_t)hj : -1;
							D0[k] = d0;
							hint2[gi] = make_int2((int32_t)hj, __float_as_int(lb_next)); // cost class 0
							if (matched)
							{
								if (A.C.dedup)
									atomicMin(&Wall[A.w_off + hj], s);
								else if (A.C.gate)
									atomicMin(&winner[A.tgt_off + hj], A.C.key_hi | (unsigned long long)s);
							}
						}
					}
				}
				if (!certified)
				{
					M[k] = MULLS_NEEDS_SEARCH;
					D0[k] = out.w;
					const uint32_t qc = atomicAdd(&FC[kc].ucount, 1u);
					if (qc < MULLS_FA_SMALL)
					{
						const uint32_t q = atomicAdd(&s_q[0], 1u);
						uq[q] = out;
						us[q] = s | (kc << 24);
					}
				}
			}
		}
		// matches certified here, per class (a wave's lanes belong to at most a few classes)
		for (uint32_t c = 0; c < ncls; c++)
		{
			const unsigned long long b = __ballot(matched && kc == c);
			if ((threadIdx.x & 63u) == 0u && b)
				atomicAdd(&FC[c].matched, (uint32_t)__popcll(b));
		}
	}
	__syncthreads();
	FST(1);
	// classes with too many leftovers drop out: their nn_idx / nn_d2 as cert_class leaves them
	uint32_t over = 0u;
	for (uint32_t c = 0; c < ncls; c++)
		if (FC[c].ucount > MULLS_FA_SMALL)
			over |= 1u << c;
	if (over)
	{
#pragma unroll
		for (int k = 0; k < MULLS_FA_TRIPS; k++)
			if ((F[k] & MULLS_F_ALIVE) && (over >> K[k] & 1u))
			{
				const uint32_t gi = FC[K[k]].src_off + S[k];
				nn_idx[gi] = M[k];
				nn_d2[gi] = D0[k];
			}
	}
	// ---- stage 2: the leftovers against the grids in global memory ----------------------------------------------------------------------
	{
		const uint32_t U = s_q[0];
		const uint32_t sub = threadIdx.x & (MULLS_LDS_GROUP - 1u), grp = threadIdx.x / MULLS_LDS_GROUP;
		for (uint32_t i = grp; i < U; i += MULLS_ICP_BLOCK / MULLS_LDS_GROUP)
		{
			const uint32_t kc = us[i] >> 24, s = us[i] & 0xffffffu;
			if (over >> kc & 1u)
				continue;
			const FaClass &A = FC[kc];
			const GridDesc &g = grids[A.cls];
			const GlobGrid L = {tsorted + A.tgt_off, reinterpret_cast<const uint16_t *>(cell_start) + g.cell_off};
			nnkey bk;
			float sec, Rfin;
			uint32_t trips;
			CandOut co = {0xffffffffu, 0xffffffffu, 0.0f};
			search_query(g, L, uq[i], A.C.r, A.C.m, sub, bk, sec, Rfin, trips, co);
			if (sub == 0 && commit_search(A.C, pd[A.cls], s, bk, sec, Rfin, trips, co, nn_idx, nn_d2, hint2, Wall + A.w_off, winner))
				atomicAdd(&FC[kc].matched, 1u);
		}
	}
	__threadfence_block(); // the leftovers' nn_idx / nn_d2, read back below by the lanes that own the points
	__syncthreads();
	FST(2);
	// ---- stage 3: duplicate rule (cregistration.hpp:1762-1789) and rejection chain (:1794-1830) ---------------------------------------
#pragma unroll
	for (int k = 0; k < MULLS_FA_TRIPS; k++)
	{
		const uint32_t kc = K[k];
		bool alive = false, valid = false;
		const bool mine = (F[k] & MULLS_F_ALIVE) && !(over >> kc & 1u);
		if (mine)
		{
			const FaClass &A = FC[kc];
			const uint32_t s = S[k], gi = A.src_off + s;
			const float max_sqr = A.thr * A.thr; // CorrespondenceRejectorDistance::setMaximumDistance (float)
			const bool any_match = A.matched > 0u, normal_check = A.cls != 5u; // vertex correspondences skip the direction check (:1292)
			int32_t m = M[k];
			float dist = D0[k];
			if (m == MULLS_NEEDS_SEARCH)
			{
				m = nn_idx[gi];
				dist = nn_d2[gi];
			}
			if (A.C.dedup && m >= 0 && Wall[A.w_off + m] != s)
				m = -1; // first source (lowest index) matched to a target keeps it; the others become unmatched
			bool fresh = false;
			alive = true;
			float4 n2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			float wdist = 0.0f;
			if (any_match)
			{
				valid = m >= 0;
				if (A.C.gate && m < 0)
				{
					alive = false; // unmatched or duplicate-losing source points vanish for good (:1762-1789)
					valid = false;
				}
				if (valid)
				{
					valid = rp.rej_strict ? dist < max_sqr : !(dist > max_sqr); // CorrespondenceRejectorDistance (see mulls_params.rejector_strict)
					if (valid)
					{
						wdist = dist; // pcl::Correspondence::distance (shares storage with ::weight)
						if (PM[k] == m)
							n2 = Q1[k];
						else
						{
							match[gi] = m;
							float4 q2;
							tgt_record(rp.tgt_stage, rp.tgt_map, pd[A.cls], (uint32_t)m, tpos, tnrm, q2, n2);
							mq[2u * gi] = q2;
							mq[2u * gi + 1u] = n2;
						}
						fresh = true;
					}
				}
			}
			else if (A.C.gate)
			{
				alive = false; // the whole cloud was swapped for an empty one; reference behaviour undefined, see oracle
				valid = false;
			}
			else
				valid = (F[k] & MULLS_F_VALID) != 0; // the previous Corr_f is still in place (SURVEY B-4) and goes through the direction check again
			if (valid && normal_check)
			{
				if (!fresh)
					n2 = Q1[k]; // the standing correspondence's target direction
				const float4 n1 = Nn[k];
				const double dot = (double)n1.x * (double)n2.x + (double)n1.y * (double)n2.y + (double)n1.z * (double)n2.z;
				const float c = (float)fabs(dot);
				if ((double)c < rp.cos_bearing)
					valid = false;
			}
			const uint32_t nf = (alive ? MULLS_F_ALIVE : 0u) | (valid ? MULLS_F_VALID : 0u);
			if (nf != F[k])
				flag[gi] = (uint8_t)nf;
			if (fresh)
				wd[gi] = wdist;
		}
		for (uint32_t c = 0; c < ncls; c++)
		{
			const unsigned long long ba = __ballot(mine && alive && kc == c), bv = __ballot(mine && valid && kc == c);
			if ((threadIdx.x & 63u) == 0u)
			{
				if (ba)
					atomicAdd(&FC[c].alive, (uint32_t)__popcll(ba));
				if (bv)
					atomicAdd(&FC[c].valid, (uint32_t)__popcll(bv));
			}
		}
	}
	__syncthreads();
	FST(3);
	if (threadIdx.x < ncls && !(over >> threadIdx.x & 1u))
	{
		const FaClass &A = FC[threadIdx.x];
		CloudDesc &d = pd[A.cls];
		d.n_matched = A.matched;
		d.alive_next = A.alive;
		d.valid_next = A.valid;
		d.n_search = A.ucount;
	}
	// class slots -> classes
	uint32_t over_cls = 0u;
	for (uint32_t c = 0; c < ncls; c++)
		if (over >> c & 1u)
			over_cls |= 1u << FC[c].cls;
	return over_cls;
#undef FST
}

__global__ __launch_bounds__(MULLS_ICP_BLOCK) void k_icp(const Job *__restrict__ rjobs, const uint32_t *__restrict__ pair_rjob, const uint32_t *__restrict__ order,
														   uint32_t npairs, uint32_t pair_base, uint32_t *__restrict__ queue, CloudDesc *__restrict__ descs,
														   const PairSetup *__restrict__ setup, RunParams rp, mulls::IcpConst K, float4 *__restrict__ spos,
														   float4 *__restrict__ snrm, const GridDesc *__restrict__ grids, const uint32_t *__restrict__ cell_start,
														   const float4 *__restrict__ tsorted, uint8_t *flag, int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2,
														   unsigned long long *__restrict__ winner, const float4 *__restrict__ tnrm, int32_t *__restrict__ match,
														   float *__restrict__ wd, const float4 *__restrict__ tpos, int32_t *__restrict__ nn_hint, float4 *__restrict__ mq,
														   const uint32_t *__restrict__ bbox, uint32_t cap, IcpOut *__restrict__ outs, mulls_iter_trace *__restrict__ trace,
														   uint32_t trace_cap, uint32_t lds_bytes)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	const LdsLayout Y = lds_layout(lds_raw, cap, rp.grid_maxcells);
	const bool fuse = rp.debug_stop != 7u; // diagnostics (MULLS_DEBUG_STOP=7): every class through the unfused passes
	void *R = lds_raw; // the term buffer of the row sums: the staged cloud is dead while rows are summed
	__shared__ PairState ps;
	__shared__ mulls::PairIter h;
	__shared__ double rows[MULLS_NC][MULLS_NTERM_PAD], comb[MULLS_NTERM_PAD];
	__shared__ uint32_t s_ticket, s_nvalid[MULLS_NC], s_nalive[MULLS_NC];
	__shared__ CloudDesc s_desc[MULLS_NC]; // the pair's class descriptors and grids live in LDS while the pair iterates: the per-class
	__shared__ GridDesc s_grid[MULLS_NC];
	__shared__ Job s_jobs[MULLS_NC];
	__shared__ unsigned long long s_tf[8];
	__shared__ SolveWs s_ws;  // counters are read and rolled over many times per iteration (a scalar load from memory each time otherwise)
	__shared__ int s_go;
	__shared__ unsigned long long s_src_pts, s_tgt_pts, s_corr_pts, s_t[6], s_mark;
	__shared__ uint32_t s_q[4], s_dup[MULLS_NC]; // s_dup: fused_all kept a duplicate table for the class
	__shared__ CertLds<MULLS_CERT_SMALL> s_cert;  // cert_class's leftover queries and reduction scratch
	// fused_all's scratch, in the LDS the staged target cloud uses otherwise: leftover queue, class table, then the duplicate tables
	float4 *fa_uq = reinterpret_cast<float4 *>(Y.P);
	uint32_t *fa_us = reinterpret_cast<uint32_t *>(fa_uq + MULLS_FA_QCAP);
	FaClass *s_fc = reinterpret_cast<FaClass *>(fa_us + MULLS_FA_QCAP);
	uint32_t *Wall = reinterpret_cast<uint32_t *>(s_fc + MULLS_NC);
	const uint32_t wall_cap = lds_bytes > (uint32_t)(reinterpret_cast<unsigned char *>(Wall) - lds_raw) ? (lds_bytes - (uint32_t)(reinterpret_cast<unsigned char *>(Wall) - lds_raw)) / 4u : 0u;
// phase clock: lane 0 charges the time since the last mark to phase k
#define PHASE(k)                                    \
	do                                              \
	{                                               \
		if (threadIdx.x == 0)                       \
		{                                           \
			const unsigned long long now_ = wall_clock64(); \
			s_t[k] += now_ - s_mark;                \
			s_mark = now_;                          \
		}                                           \
	} while (0)

	for (;;)
	{
		__syncthreads();
		if (threadIdx.x == 0)
			s_ticket = atomicAdd(queue, 1u);
		__syncthreads();
		if (s_ticket >= npairs)
			return;
		const uint32_t pair = pair_base + order[s_ticket];
		CloudDesc *pd = s_desc;
		const uint32_t j0 = pair_rjob[pair], j1 = pair_rjob[pair + 1u];
		if (threadIdx.x < MULLS_NC)
		{
			s_desc[threadIdx.x] = descs[(size_t)pair * MULLS_NC + threadIdx.x];
			s_grid[threadIdx.x] = grids[(size_t)pair * MULLS_NC + threadIdx.x];
			if (j0 + threadIdx.x < j1)
				s_jobs[threadIdx.x] = rjobs[j0 + threadIdx.x];
		}
		__syncthreads();
		IcpOut &O = outs[pair];
		if (threadIdx.x == 0)
		{
			mulls::pair_iter_init(h, setup[pair].guess, K);
			for (int k = 0; k < 12; k++)
				ps.T[k] = (k % 5 == 0) ? 1.0 : 0.0; // TempTran = identity at i = 0
			for (int k = 0; k < 6; k++)
				ps.x[k] = 0.0;
			for (int c = 0; c < MULLS_NC; c++)
				ps.thr[c] = K.dis_thre_unit;
			ps.iter = 0;
			ps.active = 1;
			ps.want_residual = 0;
			// source_feature_points_count (cregistration.hpp:1195-1201); while undistorting, the sizes the reference counts are those
			// of the cloned clouds, before the five non-vertex clouds are regenerated from block2->pc_*_down inside the loop
			int sfc = 0;
			for (int c = 0; c < MULLS_NC; c++)
			{
				const uint32_t n0 = rp.undistort ? pd[c].src_n0 : pd[c].src_n;
				O.nsrc0[c] = n0;
				O.ntgt0[c] = pd[c].tgt_n;
				O.ncorr[c] = 0u;
				if ((c == 1 || c == 2 || c == 3) && rp.used[c])
					sfc += (int)n0;
			}
			h.src_feature_count = sfc;
			for (int k = 0; k < 6; k++)
				O.bbox[k] = bbox[(size_t)pair * 6 + k];
			s_src_pts = s_tgt_pts = s_corr_pts = 0ull;
			O.trace_len = 0;
			for (int k = 0; k < 6; k++)
				s_t[k] = s_tf[k] = 0ull;
			s_mark = wall_clock64();
			s_t[5] = s_mark;
		}
		__syncthreads();

		for (int i = 0; i < K.max_iter_num; i++)
		{
			// --- correspondences of every used class cloud (cregistration.hpp:1272-1292) ---------------------------------------------
			if (threadIdx.x == 0)
				for (int c = 0; c < MULLS_NC; c++)
					if (rp.used[c] && pd[c].alive_cur >= 3u && pd[c].tgt_n >= 3u)
					{
						s_src_pts += pd[c].alive_cur;
						s_tgt_pts += pd[c].tgt_n;
					}
			if (threadIdx.x < MULLS_NC * MULLS_NTERM_PAD)
				rows[threadIdx.x / MULLS_NTERM_PAD][threadIdx.x % MULLS_NTERM_PAD] = 0.0;
			// Flattened pass (fused_all) when every called class cloud fits: at most MULLS_FA_TRIPS x 1024 source points in all, the
			// duplicate tables side by side in the LDS of the staged cloud.  Otherwise class by class as before.
			uint32_t part = 0u, fa_src = 0u, fa_w = 0u;
			for (uint32_t j = j0; j < j1; j++)
			{
				const int cls = (int)s_jobs[j - j0].cls;
				const CloudDesc &d = pd[cls];
				if (class_called(rp, d, cls))
				{
					part |= 1u << cls;
					fa_src += d.src_n;
					if (rp.lds_dedup != 0u && d.alive_cur >= 500u)
						fa_w += d.tgt_n;
				}
			}
			const bool flat = fuse && rp.debug_stop != 8u && i > 0 && part != 0u && fa_src <= MULLS_FA_TRIPS * MULLS_ICP_BLOCK && fa_w <= wall_cap;
			uint32_t pend = 0u;
			bool has_ground = false;
			for (uint32_t j = j0; j < j1; j++)
				has_ground |= s_jobs[j - j0].cls == 0u;
			if (flat)
			{
				__syncthreads(); // the previous iteration's LDS contents have been consumed
				const uint32_t over = fused_all(rp, ps, s_jobs, j1 - j0, part, pd, s_grid, fa_uq, fa_us, Wall, s_fc, s_q, spos, snrm, cell_start, tsorted, flag, nn_idx, nn_d2, winner,
												tnrm, match, wd, tpos, nn_hint, mq, s_tf);
				if (threadIdx.x < s_q[1]) // (the class table lies where lds_search_class stages the target cloud)
					s_dup[s_fc[threadIdx.x].cls] = s_fc[threadIdx.x].C.dedup ? 1u : 0u;
				for (uint32_t j = j0; j < j1; j++)
				{
					const Job job = s_jobs[j - j0];
					const int cls = (int)job.cls;
					pend |= 1u << cls;
					if (part >> cls & 1u)
					{
						if (!(over >> cls & 1u))
							continue;
						// too many leftovers: the class's duplicate table moves to where lds_search_class expects it, the target cloud is staged
						__syncthreads();
						if (s_dup[cls])
							rebuild_dup_table(Y.W, pd[cls], flag, nn_idx);
						lds_search_class(rp, ps, job, pd[cls], s_grid[cls], Y, lds_raw, spos, snrm, cell_start, tsorted, flag, nn_idx, nn_d2, winner, tnrm, match, wd, tpos, nn_hint, mq);
						continue;
					}
					// a class that sits this iteration out: its points still move (cert_class)
					__syncthreads();
					(void)cert_class<MULLS_ICP_BLOCK>(s_cert, rp, ps, job, pd[cls], s_grid[cls], Y.W, spos, snrm, cell_start, tsorted, flag, nn_idx, nn_d2, winner, tnrm, match, wd, tpos,
													  nn_hint, mq);
				}
			}
			else
			{
			// Class order: the classes whose weight is 1 first, then roof and ground (their weight needs the other classes' counts,
			// cregistration.hpp:1886-1894).  `pend`: classes whose row is summed from memory after the count test (uniform).
			for (int pass = 0; pass < 3; pass++)
				for (uint32_t j = j0; j < j1; j++)
				{
					const Job job = s_jobs[j - j0];
					const int cls = (int)job.cls;
					if (pass == 0 ? (cls == 0 || cls == 4) : (pass == 1 ? cls != 4 : cls != 0))
						continue;
					CloudDesc &d = pd[cls];
					const GridDesc g = s_grid[cls];
					__syncthreads(); // the previous class cloud's LDS contents (duplicate table, staged cloud, term buffer) have been consumed
					if (fuse && class_called(rp, d, cls) && d.src_n <= MULLS_FUSE_TRIPS * MULLS_ICP_BLOCK)
					{
						int r = fused_class(rp, ps, job, pd, g, Y, lds_raw, !(cls == 4 && has_ground), spos, snrm, cell_start, tsorted, flag, nn_idx, nn_d2, winner, tnrm,
											match, wd, tpos, nn_hint, mq, rows[cls], s_tf);
						if (r == 2)
						{
							__syncthreads();
							lds_search_class(rp, ps, job, d, g, Y, lds_raw, spos, snrm, cell_start, tsorted, flag, nn_idx, nn_d2, winner, tnrm, match, wd, tpos, nn_hint, mq);
						}
						if (r)
							pend |= 1u << cls;
						continue;
					}
					if (!cert_class<MULLS_ICP_BLOCK>(s_cert, rp, ps, job, d, g, Y.W, spos, snrm, cell_start, tsorted, flag, nn_idx, nn_d2, winner, tnrm, match, wd, tpos,
													 nn_hint, mq))
					{
						__syncthreads();
						lds_search_class(rp, ps, job, d, g, Y, lds_raw, spos, snrm, cell_start, tsorted, flag, nn_idx, nn_d2, winner, tnrm, match, wd, tpos, nn_hint, mq);
					}
					pend |= 1u << cls;
				}
			}
			__threadfence_block();
			__syncthreads();
			if (threadIdx.x == 0 && i < 24)
				O.t_search_it[i] = (uint32_t)(wall_clock64() - s_mark);
			PHASE(0);
			// --- counters rolled over (k_finish), count test, threshold update ------------------------------------------------------------
			if (threadIdx.x < MULLS_NC)
			{
				const int c = threadIdx.x;
				CloudDesc &d = pd[c];
				if (class_called(rp, d, c))
				{
					d.n_valid = d.valid_next;
					d.alive_cur = d.alive_next;
				}
				d.alive_next = 0;
				d.valid_next = 0;
				d.n_matched = 0;
				s_nvalid[c] = d.n_valid;
				s_nalive[c] = d.alive_cur;
			}
			__syncthreads();
			if (threadIdx.x == 0)
			{
				h.iters = i + 1;
				mulls_iter_trace *tr = nullptr;
				if (trace && (uint32_t)O.trace_len < trace_cap)
				{
					tr = &trace[(size_t)pair * trace_cap + (uint32_t)O.trace_len];
					tr->iter = i;
					for (int c = 0; c < MULLS_NC; c++)
					{
						tr->ncorr[c] = s_nvalid[c];
						tr->nsrc[c] = s_nalive[c];
						tr->thr[c] = h.thr[c];
					}
					for (int k = 0; k < 36; k++)
						tr->atpa[k] = 0.0;
					for (int k = 0; k < 6; k++)
						tr->atpb[k] = tr->x[k] = 0.0;
				}
				for (int c = 0; c < MULLS_NC; c++)
				{
					O.ncorr[c] = s_nvalid[c];
					s_corr_pts += s_nvalid[c];
				}
				s_go = mulls::step_counts(h, K, s_nvalid) ? 1 : 0;
				if (!s_go && tr)
					O.trace_len++;
			}
			__syncthreads();
			if (!s_go)
				break; // process code -2
			PHASE(1);
			// --- normal equations (cregistration.hpp:1869-1938) -------------------------------------------------------------------------------
			int cnt[MULLS_NC];
			for (int c = 0; c < MULLS_NC; c++)
				cnt[c] = (int)s_nvalid[c];
			for (uint32_t j = j0; j < j1; j++)
			{
				const int cls = (int)s_jobs[j - j0].cls;
				if (!(pend >> cls & 1u))
					continue; // the fused pass summed this row from its registers
				const AccumCtx A = accum_ctx(rp, cls, i, false, class_weight(rp, cls, false, cnt));
				class_row(A, ps.x, pd[cls], spos, mq, flag, wd, R, rows[cls]);
			}
			if (threadIdx.x < MULLS_NTERM)
				combine_rows(rp, false, rows, comb, (int)threadIdx.x);
			__syncthreads();
			PHASE(2);
			// --- solve, step and convergence tests (:1924-1964, :1333-1357) --------------------------------------------------------------------
			if (threadIdx.x < 64)
				solve_wave(h, K, comb, i, s_ws);
			if (threadIdx.x == 0)
			{
				if (trace && (uint32_t)O.trace_len < trace_cap)
				{
					mulls_iter_trace *tr = &trace[(size_t)pair * trace_cap + (uint32_t)O.trace_len];
					for (int k = 0; k < 36; k++)
						tr->atpa[k] = s_ws.N[k];
					for (int k = 0; k < 6; k++)
					{
						tr->atpb[k] = s_ws.b[k];
						tr->x[k] = h.x[k];
					}
					O.trace_len++;
				}
				for (int k = 0; k < 6; k++)
					ps.x[k] = h.x[k];
				s_go = h.done ? 0 : (h.want_residual ? 2 : 1);
				if (s_go == 1)
				{
					for (int r = 0; r < 3; r++)
						for (int c = 0; c < 4; c++)
							ps.T[r * 4 + c] = h.temp.at(r, c);
					for (int c = 0; c < MULLS_NC; c++)
						ps.thr[c] = h.thr[c];
					ps.iter = i + 1;
				}
			}
			__syncthreads();
			PHASE(3);
			if (s_go == 1)
				continue;
			if (s_go == 2)
			{
				// --- posterior residual pass with the final step and the last correspondences (:2518-2677) --------------------------------
				if (threadIdx.x < MULLS_NC * MULLS_NTERM_PAD)
					rows[threadIdx.x / MULLS_NTERM_PAD][threadIdx.x % MULLS_NTERM_PAD] = 0.0;
				__syncthreads();
				for (uint32_t j = j0; j < j1; j++)
				{
					const int cls = (int)s_jobs[j - j0].cls;
					const AccumCtx A = accum_ctx(rp, cls, i, true, class_weight(rp, cls, true, cnt));
					class_row(A, ps.x, pd[cls], spos, mq, flag, wd, R, rows[cls]);
				}
				if (threadIdx.x < MULLS_NTERM)
					combine_rows(rp, true, rows, comb, (int)threadIdx.x);
				__syncthreads();
				if (threadIdx.x == 0)
					mulls::step_residual(h, K, comb[0], comb[1]);
				__syncthreads();
				PHASE(4);
			}
			break;
		}
		if (threadIdx.x == 0)
		{
			h.guess = h.temp * h.guess; // :1403 (TempTran is the identity after a failure)
			for (int k = 0; k < 16; k++)
				O.T[k] = h.guess.v[k];
			for (int k = 0; k < 36; k++)
				O.info[k] = h.info.v[k];
			O.sigma2 = h.sigma2;
			O.ratio = h.ratio;
			O.code = h.code;
			O.iters = h.iters;
			O.singular = h.singular;
			O.src_pts = s_src_pts;
			O.tgt_pts = s_tgt_pts;
			O.corr_pts = s_corr_pts;
			for (int k = 0; k < 5; k++)
				O.t_phase[k] = s_t[k];
			O.t_phase[5] = wall_clock64() - s_t[5];
			for (int k = 0; k < 6; k++)
				O.t_fused[k] = s_tf[k];
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------
#include "launch.h"

int launch_icp(hipStream_t st, uint32_t npairs, uint32_t pair_base, const Job *rjobs, const uint32_t *pair_rjob, const uint32_t *order, uint32_t *queue,
			   CloudDesc *descs, const PairSetup *setup, const RunParams &rp, const mulls::IcpConst &K, float4 *spos, float4 *snrm, const GridDesc *grids,
			   const uint32_t *cell_start, const float4 *tsorted, uint8_t *flag, int32_t *nn_idx, float *nn_d2, unsigned long long *winner, const float4 *tnrm,
			   int32_t *match, float *wd, const float4 *tpos, int32_t *nn_hint, float4 *mq, const uint32_t *bbox, uint32_t cap, uint32_t maxcells, IcpOut *outs,
			   mulls_iter_trace *trace, uint32_t trace_cap)
{
	const DevLaunch D = dev_launch<2>([](DevLaunch &) { // per device (launch.h)
		if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_icp), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - MULLS_ICP_STATIC_LDS) != hipSuccess)
			return false;
		hipFuncAttributes fa;
		return hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_icp)) == hipSuccess && fa.sharedSizeBytes <= (size_t)MULLS_ICP_STATIC_LDS;
	});
	if (!D.ok)
		return -1;
	const uint32_t n_cu = D.n_cu;
	if (!npairs)
		return 0;
	size_t lds = nn_lds_bytes(cap, maxcells, true);
	const size_t red = MULLS_RED_BYTES;
	if (lds < red)
		lds = red;
	hipLaunchKernelGGL(k_icp, dim3(npairs < n_cu ? npairs : n_cu), dim3(MULLS_ICP_BLOCK), lds, st, rjobs, pair_rjob, order, npairs, pair_base, queue, descs, setup, rp, K,
					   spos, snrm, grids, cell_start, tsorted, flag, nn_idx, nn_d2, winner, tnrm, match, wd, tpos, nn_hint, mq, bbox, cap, outs,
					   trace, trace_cap, (uint32_t)lds);
	return 0;
}
