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
#include "lds_tier.h"

#define MULLS_ICP_BLOCK MULLS_LDS_BLOCK
static_assert(MULLS_ICP_BLOCK == MULLS_ACC_LANES, "one source point per lane and trip, in the reduction's lane order");

__global__ __launch_bounds__(MULLS_ICP_BLOCK) void k_icp(const Job *__restrict__ rjobs, const uint32_t *__restrict__ pair_rjob, const uint32_t *__restrict__ order,
														   uint32_t npairs, uint32_t pair_base, uint32_t *__restrict__ queue, CloudDesc *__restrict__ descs,
														   const PairSetup *__restrict__ setup, RunParams rp, mulls::IcpConst K, float4 *__restrict__ spos,
														   float4 *__restrict__ snrm, const GridDesc *__restrict__ grids, const uint32_t *__restrict__ cell_start,
														   const float4 *__restrict__ tsorted, uint8_t *flag, int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2,
														   unsigned long long *__restrict__ winner, const float4 *__restrict__ tnrm, int32_t *__restrict__ match,
														   float *__restrict__ wd, const float4 *__restrict__ tpos, int32_t *__restrict__ nn_hint, float4 *__restrict__ mq,
														   const uint32_t *__restrict__ bbox, uint32_t cap, IcpOut *__restrict__ outs, mulls_iter_trace *__restrict__ trace,
														   uint32_t trace_cap)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	const LdsLayout Y = lds_layout(lds_raw, cap, rp.grid_maxcells);
	double *R = reinterpret_cast<double *>(lds_raw); // the reduction's transposition buffer: the staged cloud is dead while rows are summed
	__shared__ PairState ps;
	__shared__ mulls::PairIter h;
	__shared__ double rows[MULLS_NC][MULLS_NTERM_PAD], comb[MULLS_NTERM_PAD];
	__shared__ uint32_t s_ticket, s_nvalid[MULLS_NC], s_nalive[MULLS_NC];
	__shared__ int s_go;
	__shared__ unsigned long long s_src_pts, s_tgt_pts, s_corr_pts;

	for (;;)
	{
		__syncthreads();
		if (threadIdx.x == 0)
			s_ticket = atomicAdd(queue, 1u);
		__syncthreads();
		if (s_ticket >= npairs)
			return;
		const uint32_t pair = pair_base + order[s_ticket];
		CloudDesc *pd = descs + (size_t)pair * MULLS_NC;
		const uint32_t j0 = pair_rjob[pair], j1 = pair_rjob[pair + 1u];
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
			for (uint32_t j = j0; j < j1; j++)
			{
				const Job job = rjobs[j];
				CloudDesc &d = pd[job.cls];
				const GridDesc g = grids[(size_t)pair * MULLS_NC + job.cls];
				__syncthreads(); // the previous class cloud's LDS contents (duplicate table, staged cloud) have been consumed
				if (!cert_class<MULLS_ICP_BLOCK>(rp, ps, job, d, g, Y.W, spos, snrm, cell_start, tsorted, flag, nn_idx, nn_d2, winner, tnrm, match, wd, tpos, nn_hint,
												 mq))
				{
					__syncthreads();
					lds_search_class(rp, ps, job, d, g, Y, lds_raw, spos, snrm, cell_start, tsorted, flag, nn_idx, nn_d2, winner, tnrm, match, wd, tpos, nn_hint, mq);
				}
			}
			__threadfence_block();
			__syncthreads();
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
			// --- normal equations (cregistration.hpp:1869-1938) -------------------------------------------------------------------------------
			int cnt[MULLS_NC];
			for (int c = 0; c < MULLS_NC; c++)
				cnt[c] = (int)s_nvalid[c];
			if (threadIdx.x < MULLS_NC * MULLS_NTERM_PAD)
				rows[threadIdx.x / MULLS_NTERM_PAD][threadIdx.x % MULLS_NTERM_PAD] = 0.0;
			__syncthreads();
			for (uint32_t j = j0; j < j1; j++)
			{
				const int cls = (int)rjobs[j].cls;
				const AccumCtx A = accum_ctx(rp, cls, i, false, class_weight(rp, cls, false, cnt));
				class_row(A, ps.x, pd[cls], spos, mq, flag, wd, R, rows[cls]);
			}
			if (threadIdx.x < MULLS_NTERM)
				combine_rows(rp, false, rows, comb, (int)threadIdx.x);
			__syncthreads();
			// --- solve, step and convergence tests (:1924-1964, :1333-1357) --------------------------------------------------------------------
			if (threadIdx.x == 0)
			{
				mulls::Mat6 N;
				double b[6];
				mulls::normal_from_row(comb, N, b);
				mulls::step_solve(h, K, N, b, i);
				if (trace && (uint32_t)O.trace_len < trace_cap)
				{
					mulls_iter_trace *tr = &trace[(size_t)pair * trace_cap + (uint32_t)O.trace_len];
					for (int k = 0; k < 36; k++)
						tr->atpa[k] = N.v[k];
					for (int k = 0; k < 6; k++)
					{
						tr->atpb[k] = b[k];
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
					const int cls = (int)rjobs[j].cls;
					const AccumCtx A = accum_ctx(rp, cls, i, true, class_weight(rp, cls, true, cnt));
					class_row(A, ps.x, pd[cls], spos, mq, flag, wd, R, rows[cls]);
				}
				if (threadIdx.x < MULLS_NTERM)
					combine_rows(rp, true, rows, comb, (int)threadIdx.x);
				__syncthreads();
				if (threadIdx.x == 0)
					mulls::step_residual(h, K, comb[0], comb[1]);
				__syncthreads();
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
	static bool attr_set = false;
	static uint32_t n_cu = 256;
	if (!attr_set)
	{
		if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_icp), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 6144) != hipSuccess)
			return -1;
		int dev = 0, cus = 0;
		if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
			n_cu = (uint32_t)cus;
		attr_set = true;
	}
	if (!npairs)
		return 0;
	size_t lds = nn_lds_bytes(cap, maxcells, true);
	const size_t red = (size_t)MULLS_RED_TERMS * MULLS_ICP_BLOCK * sizeof(double);
	if (lds < red)
		lds = red;
	hipLaunchKernelGGL(k_icp, dim3(npairs < n_cu ? npairs : n_cu), dim3(MULLS_ICP_BLOCK), lds, st, rjobs, pair_rjob, order, npairs, pair_base, queue, descs, setup, rp, K,
					   spos, snrm, grids, cell_start, tsorted, flag, nn_idx, nn_d2, winner, tnrm, match, wd, tpos, nn_hint, mq, bbox, cap, outs,
					   trace, trace_cap);
	return 0;
}
