// loop.cpp — mulls_batch_run: one registration run of a device-resident batch (reference: CRegistration::mm_lls_icp, cregistration.hpp:1114-1440).
// Set-up launches (clone + initial guess, intersection crop, keep-less thinning, target grids), then one of three loops with identical results:
//   run_resident      the device-resident loop (k_icp: one launch iterates every pair to the end)
//   run_device_step   the lock-step loop stepped on the device: four or seven launches per iteration, one 8-byte word back per launch set
//   run_host_step     the lock-step loop stepped by the host (per-iteration traces)
#include "batch.h"

using namespace mulls_drv;

namespace
{
// what the phases of one run share
struct Run
{
	mulls_ctx *ctx;
	mulls_batch *B;
	const mulls_params *P;
	mulls_result *results;
	int n;
	hipStream_t st;
	std::chrono::steady_clock::time_point wall0;
	EvTimer evt;
	RunParams rp;
	mulls::IcpConst K;
	uint32_t lds_cap = 0;
	uint32_t max_big_tgt = 0; // the largest target class cloud on the global-memory tier (launch_search: how long a small batch keeps chunk-level jobs)
	int tier = 0;
	bool use_grid = false, dstep = false, resident = false;
};
#define RUN_ALIASES \
	mulls_ctx *ctx = R.ctx; \
	mulls_batch *B = R.B; \
	const mulls_params *P = R.P; \
	mulls_result *results = R.results; \
	const int n = R.n; \
	hipStream_t st = R.st; \
	RunParams &rp = R.rp; \
	const mulls::IcpConst &K = R.K; \
	EvTimer &evt = R.evt; \
	const uint32_t lds_cap = R.lds_cap; \
	const int tier = R.tier; \
	const bool use_grid = R.use_grid; \
	(void)P, (void)results, (void)n, (void)st, (void)rp, (void)K, (void)evt, (void)lds_cap, (void)tier, (void)use_grid;


// run parameters, job tables, tier; then the set-up launches: clone + initial guess + intersection filter (cregistration.hpp:1180-1188), keep-less
// thinning, the target grids
int run_setup(Run &R)
{
	mulls_ctx *ctx = R.ctx;
	mulls_batch *B = R.B;
	const mulls_params *P = R.P;
	mulls_result *results = R.results;
	const int n = R.n;
	hipStream_t st = R.st;
	EvTimer &evt = R.evt;
	int rc;

	RunParams &rp = R.rp;
	std::memset(&rp, 0, sizeof(rp));
	rp.pull_comb = 1; // the host only needs the assembled system (or VTPV and the observation count) of each pair
	for (int c = 0; c < MULLS_NC; c++)
		rp.used[c] = P->used_feature_type[c] == '1';
	rp.w_balance = P->weight_strategy[0] == '1';
	rp.w_resid = P->weight_strategy[1] == '1';
	rp.w_dist = P->weight_strategy[2] == '1';
	rp.w_inten = P->weight_strategy[3] == '1';
	rp.normal_shooting = P->normal_shooting_on != 0;
	rp.undistort = P->apply_motion_undistortion != 0;
	rp.crop = P->apply_intersection_filter != 0 && !rp.undistort; // cregistration.hpp:1186
	rp.faithful = P->faithful != 0;
	rp.rej_strict = P->rejector_strict != 0;
	rp.z_xy_ratio = P->z_xy_balanced_ratio;
	rp.win_pt = P->pt2pt_residual_window;
	rp.win_pl = P->pt2pl_residual_window;
	rp.win_li = P->pt2li_residual_window;
	rp.cos_bearing = std::cos(P->normal_bearing / 180.0 * M_PI);
	rp.resid_from_iter = 2;
	init_cert(ctx, rp);
	if ((rc = take_epochs(ctx, B, (uint32_t)std::max(P->max_iter_num, 0) + 2u, rp)) != MULLS_OK)
		return rc;
	rp.debug_stop = (uint32_t)ctx->opt[MULLS_OPT_DEBUG_STOP];

	// Lock-step tiers: the O(1) half of every iteration (count test, 6x6 solve, convergence tests, residual) runs on the device behind the
	// accumulation (k_finish_step) unless the caller wants per-iteration traces, which the host half collects (MULLS_HOST_STEP=1: diagnostics).
	// Nothing but one 8-byte word crosses PCIe per iteration then, and there is no host work to hide behind a second sub-batch.
	bool dstep = P->max_iter_num > 0;
	for (int p = 0; p < n && dstep; p++)
		dstep = !(results[p].trace && results[p].trace_cap > 0);
	dstep = dstep && ctx->opt[MULLS_OPT_HOST_STEP] == 0.0;
	uint32_t lds_cap = 0;
	int tier = 0;
	bool resident = false;
	// device-stepped loop: one sub-batch, or two on two streams for the batch sizes whose kernels leave most of the chip idle (not while profiling:
	// the event sets are laid out for one)
	const int dstep_nsub = (ctx->profiling == 0 && n >= (int)ctx->opt[MULLS_OPT_SPLIT_MIN_PAIRS] && n <= (int)ctx->opt[MULLS_OPT_SPLIT_MAX_PAIRS]) ? 2 : 1;
	rc = prepare_run(ctx, B, P, rp, &lds_cap, &tier, &resident, dstep ? dstep_nsub : 0, true);
	if (rc != MULLS_OK)
		return rc;
	const bool use_grid = tier != 0;
	R.lds_cap = lds_cap, R.tier = tier, R.use_grid = use_grid, R.dstep = dstep, R.resident = resident;
	R.max_big_tgt = 0;
	if (tier == 3)
		for (const CloudDesc &d : B->descs_h)
			if (d.tier == MULLS_TIER_BM)
				R.max_big_tgt = std::max(R.max_big_tgt, d.tgt_n0);

	// LDS tier: the target clouds are cropped and their grids built in one pass that writes no cropped copy (k_tgt_grid) — unless something needs the
	// copy (the keep-less thinning, the normal-shooting search) or a cloud of a class the run does not read is larger than the kernel's lanes cover
	// (a mixed batch — tier 3 — exists only with the fused setup: its LDS-tier clouds are the ones that fit, choose_tier)
	bool fused_tgt = tier == 2 && ctx->opt[MULLS_OPT_FUSED_TGT_SETUP] != 0.0 && !P->keep_less_source_points && !rp.normal_shooting && B->n_big_tgt == 0;
	for (size_t k = 0; k < B->descs_h.size() && fused_tgt; k++)
		fused_tgt = B->descs_h[k].tgt_n0 <= MULLS_LDS_MAXPTS;
	fused_tgt = fused_tgt || tier == 3;
	if (fused_tgt)
	{
		rp.tgt_stage = B->stage;
		rp.tgt_map = B->tmap;
		// k_tgt_grid counting-sorts grids of fewer than 32768 cells in LDS; beyond, it falls back to a bitonic sort of (cell, rank) keys that takes four times
		// as long (real scans' 2 000 - 5 000-point class clouds leave room for 30 000+ cells: 3.7 ms of setup per 4096 pairs against 0.9, profiles/r06_experiments.txt)
		rp.grid_maxcells = std::min(rp.grid_maxcells, 32767u);
	}

	if (rp.debug_stop == 20u || rp.debug_stop == 21u)
	{
		if (!B->dbg && dmalloc(ctx, &B->dbg, 16) != MULLS_OK)
			return MULLS_E_HIP;
		HIPCHK(ctx, hipMemsetAsync(B->dbg, 0, 16 * sizeof(unsigned long long), st));
		rp.dbg_ticks = B->dbg;
	}

	// setup: clone + initial guess + intersection filter (cregistration.hpp:1180-1188), then the target grids
	evt.begin(&ctx->prof.ms_setup);
	launch_clone_src(st, (uint32_t)B->setup_jobs_h.size(), B->setup_jobs, B->descs, B->setup, B->stage, B->tmp_pos, B->tmp_nrm, B->bbox, rp);
	launch_crop(st, (uint32_t)n, B->descs, B->setup, B->bbox, B->stage, B->tmp_pos, B->tmp_nrm, B->spos, B->snrm, B->tpos, B->tnrm, B->flag,
				B->match, B->wd, rp, B->grids, (uint32_t)B->big_segs_h.size(), B->big_segs, (uint32_t)B->big_clouds_h.size(), B->big_clouds, B->seg_cnt,
				B->big_box);
	if (P->keep_less_source_points && !rp.undistort)
	{
		// keep_less_source_pts (cregistration.hpp:2866-2892): needs the post-filter sizes, so this (map-to-map only) option
		// costs one extra device round trip per run
		std::vector<CloudDesc> back(B->descs_h.size());
		HIPCHK(ctx, hipMemcpyAsync(back.data(), B->descs, sizeof(CloudDesc) * back.size(), hipMemcpyDeviceToHost, st));
		HIPCHK(ctx, hipStreamSynchronize(st));
		std::vector<uint8_t> skeep(std::max<size_t>(B->n_src, 1), 1), tkeep(std::max<size_t>(B->n_tgt, 1), 1);
		for (int p = 0; p < n; p++)
		{
			const CloudDesc *pd = &back[(size_t)p * MULLS_NC];
			auto T = [&](int c, int keep) { return thin_mask(tkeep.data() + pd[c].tgt_off, pd[c].tgt_n, keep, P->rng_seed, 0 * 6 + c); };
			auto S = [&](int c, int keep) { return thin_mask(skeep.data() + pd[c].src_off, pd[c].src_n, keep, P->rng_seed, 1 * 6 + c); };
			const uint32_t tg = T(MULLS_GROUND, (int)(pd[MULLS_GROUND].tgt_n / 2));
			const uint32_t tf = T(MULLS_FACADE, (int)(pd[MULLS_FACADE].tgt_n / 2));
			S(MULLS_GROUND, (int)(tg / 4));
			S(MULLS_FACADE, (int)(tf / 2));
			S(MULLS_PILLAR, (int)pd[MULLS_PILLAR].tgt_n);
			S(MULLS_BEAM, (int)pd[MULLS_BEAM].tgt_n);
			S(MULLS_ROOF, (int)pd[MULLS_ROOF].tgt_n);
			S(MULLS_VERTEX, (int)pd[MULLS_VERTEX].tgt_n);
		}
		uint8_t *d_sk = nullptr, *d_tk = nullptr;
		if (dmalloc(ctx, &d_sk, skeep.size()) != MULLS_OK || dmalloc(ctx, &d_tk, tkeep.size()) != MULLS_OK)
			return MULLS_E_HIP;
		hipError_t e = hipMemcpyAsync(d_sk, skeep.data(), skeep.size(), hipMemcpyHostToDevice, st);
		if (e == hipSuccess)
			e = hipMemcpyAsync(d_tk, tkeep.data(), tkeep.size(), hipMemcpyHostToDevice, st);
		if (e == hipSuccess)
		{
			launch_thin(st, (uint32_t)n, B->descs, d_sk, d_tk, B->spos, B->snrm, B->tpos, B->tnrm);
			e = hipStreamSynchronize(st); // the masks are freed right below
		}
		(void)hipFree(d_sk);
		(void)hipFree(d_tk);
		if (e != hipSuccess)
		{
			ctx->err = std::string("keep_less_source_points: ") + hipGetErrorString(e);
			return MULLS_E_HIP;
		}
	}
	if (fused_tgt)
		(void)launch_tgt_grid(st, (uint32_t)n, B->descs, B->setup, B->bbox, B->stage, rp, B->grids, B->tmap, B->cell_start, B->tsorted);
	else if (tier == 2)
		launch_grid_build_sort(st, (uint32_t)n, B->descs, B->grids, rp, B->tpos, B->cell_start, B->tsorted);
	launch_bm_build(st, (uint32_t)B->lclouds_h.size(), B->lclouds, (uint32_t)B->tjobs_h.size(), B->tjobs, B->descs, B->grids, B->tpos, B->bm, B->pf, B->cell_cnt, B->bm_cs,
					B->tsorted, B->bm_rank);
	evt.end();

	R.K = icp_const(P);
	return MULLS_OK;
}

// a sub-batch's slices of the batch's job tables
struct Slice
{
	uint32_t job_lo = 0, job_n = 0, cjob_lo = 0, cjob_n = 0, bjob_lo = 0, bjob_n = 0, fjob_lo = 0, fjob_n = 0, ejob_lo = 0, ejob_n = 0;
	const uint32_t *ajob_split = nullptr;
};
Slice slice_of(const mulls_batch *B, int lo, int hi, int k)
{
	auto first_of = [](const std::vector<Job> &v, uint32_t pair) {
		return (uint32_t)(std::lower_bound(v.begin(), v.end(), pair, [](const Job &j, uint32_t q) { return j.pair < q; }) - v.begin());
	};
	Slice L;
	L.job_lo = first_of(B->jobs_h, (uint32_t)lo);
	L.job_n = first_of(B->jobs_h, (uint32_t)hi) - L.job_lo;
	L.cjob_lo = first_of(B->cjobs_h, (uint32_t)lo);
	L.cjob_n = first_of(B->cjobs_h, (uint32_t)hi) - L.cjob_lo;
	L.bjob_lo = first_of(B->bjobs_h, (uint32_t)lo);
	L.bjob_n = first_of(B->bjobs_h, (uint32_t)hi) - L.bjob_lo;
	L.fjob_lo = first_of(B->fjobs_h, (uint32_t)lo);
	L.fjob_n = first_of(B->fjobs_h, (uint32_t)hi) - L.fjob_lo;
	L.ejob_lo = first_of(B->ejobs_h, (uint32_t)lo);
	L.ejob_n = first_of(B->ejobs_h, (uint32_t)hi) - L.ejob_lo;
	L.ajob_split = B->ajob_split[k];
	return L;
}

// One iteration's correspondence search + rejection chain of a sub-batch: the class clouds of each tier by that tier's kernels (a mixed batch launches both
// families; a batch on one tier only its own), then k_filter for the clouds whose correspondences are spread over several workgroups
int launch_search(Run &R, hipStream_t sst, const Slice &L, uint32_t *wl, uint32_t *wl_ctr, uint32_t parity, EvTimer &ev, int iter)
{
RUN_ALIASES
	const Job *jobs = B->jobs + L.job_lo;
	ev.begin(&ctx->prof.ms_nn);
	if (tier == 0)
		launch_nn(sst, L.job_n, jobs, B->descs, B->states, rp, B->spos, B->snrm, B->tpos, B->flag, B->nn_idx, B->nn_d2, B->winner);
	// mixed batch, first iterations: chunk-level jobs for every cloud of the global-memory tier (MULLS_OPT_BIG_EARLY_SETS)
	// ... and every iteration of a small one: a handful of class-level workgroups cannot search a dense map's leftovers fast enough (a 1 M-point map leaves most
	// points uncertified for ten iterations: its second-nearest targets are millimetres behind the nearest), while k_filter costs such a batch 6 us
	// (round 6: only against maps of more than 100 000 points per class — a 20 000-point local map's points certify within five iterations like everybody else's,
	// and from then on the class-level jobs save the k_filter launch: 1 / 4 / 16 scans against such maps 0.97 / 1.14 / 1.21 -> 0.91 / 1.04 / 1.09 ms)
	const bool early = L.ejob_n && (iter < (int)ctx->opt[MULLS_OPT_BIG_EARLY_SETS] || (L.bjob_n - L.fjob_n < 64u && R.max_big_tgt > 100000u));
	// (resident workgroups of k_cert_big: two per CU; four rounds of them while the chunk-level jobs still search)
	const uint32_t max_wgs = iter < std::max(3, (int)ctx->opt[MULLS_OPT_BIG_EARLY_SETS]) ? 2048u : 512u;
	// a small mixed batch: both tiers' class clouds in one launch (in the first two iterations, whose chunk-level jobs search, in up to three rounds of workgroups)
#ifndef MULLS_MIXED_WIDE_ITERS
#define MULLS_MIXED_WIDE_ITERS 2
#endif
#ifndef MULLS_MIXED_ROUNDS
#define MULLS_MIXED_ROUNDS 3u
#endif
	const uint32_t mixed_rounds = early && iter < MULLS_MIXED_WIDE_ITERS ? MULLS_MIXED_ROUNDS : 1u;
	const uint32_t big_n = early ? L.ejob_n : L.bjob_n;
	const Job *big_jobs = early ? B->ejobs + L.ejob_lo : B->bjobs + L.bjob_lo;
	// iteration 0 of the LDS tier's class-level jobs: the setup has applied the rigid step (identity_step), no point has a hint — every called class cloud
	// goes straight to the staged search, no light pass (k_search.hip: first_goes_direct)
	const bool first = iter == 0 && rp.lds_dedup != 0u && !rp.normal_shooting && ctx->opt[MULLS_OPT_FIRST_DIRECT] != 0.0;
	const bool together = tier == 3 && L.cjob_n && big_n &&
						  launch_cert_mixed(sst, L.cjob_n, B->cjobs + L.cjob_lo, big_n, big_jobs, max_wgs, mixed_rounds, B->descs, B->states, rp, B->spos, B->snrm, B->grids, B->cell_start, B->bm,
											B->pf, B->bm_cs, B->tsorted, B->flag, B->nn_idx, B->nn_d2, B->winner, B->tnrm, B->match, B->wd, B->tpos, B->nn_hint, B->mq, lds_cap,
											rp.grid_maxcells, first) != 0;
	if (!together && L.cjob_n &&
		launch_nn_lds(sst, L.cjob_n, B->cjobs + L.cjob_lo, B->descs, B->states, rp, B->spos, B->snrm, B->grids, B->cell_start, B->tsorted, B->flag, B->nn_idx, B->nn_d2, B->winner,
					  B->tnrm, B->match, B->wd, B->tpos, B->nn_hint, B->mq, lds_cap, rp.grid_maxcells, wl, wl_ctr, parity, first) != 0)
	{
		ctx->err = "could not raise the dynamic LDS limit of k_nn_lds";
		return MULLS_E_HIP;
	}
	if (together)
		;
	else if (early)
		launch_cert_big(sst, L.ejob_n, B->ejobs + L.ejob_lo, max_wgs, B->descs, B->states, rp, B->spos, B->snrm, B->grids, B->bm, B->pf, B->bm_cs, B->tsorted, B->flag, B->nn_idx,
						B->nn_d2, B->winner, B->tpos, B->tnrm, B->nn_hint, B->match, B->wd, B->mq);
	else if (L.bjob_n)
		launch_cert_big(sst, L.bjob_n, B->bjobs + L.bjob_lo, max_wgs, B->descs, B->states, rp, B->spos, B->snrm, B->grids, B->bm, B->pf, B->bm_cs, B->tsorted, B->flag, B->nn_idx,
						B->nn_d2, B->winner, B->tpos, B->tnrm, B->nn_hint, B->match, B->wd, B->mq);
	if (rp.normal_shooting)
		launch_nn_shoot(sst, L.job_n, jobs, B->descs, B->states, rp, B->spos, B->snrm, B->tpos, B->flag, B->nn_idx, B->nn_d2, B->winner);
	ev.end();
	ev.begin(&ctx->prof.ms_filter);
	if (tier == 0 || (tier == 2 && !rp.lds_dedup)) // (else the LDS tier's kernels ran the rejection chain themselves)
		launch_filter(sst, L.job_n, jobs, B->descs, B->states, rp, B->snrm, B->tnrm, B->flag, B->nn_idx, B->nn_d2, B->match, B->wd, B->winner, B->tpos, B->mq);
	else if (early)
		launch_filter(sst, L.ejob_n, B->ejobs + L.ejob_lo, B->descs, B->states, rp, B->snrm, B->tnrm, B->flag, B->nn_idx, B->nn_d2, B->match, B->wd, B->winner, B->tpos, B->mq, true);
	else if (L.fjob_n)
		launch_filter(sst, L.fjob_n, B->fjobs + L.fjob_lo, B->descs, B->states, rp, B->snrm, B->tnrm, B->flag, B->nn_idx, B->nn_d2, B->match, B->wd, B->winner, B->tpos, B->mq, true);
	ev.end();
	const hipError_t e = hipGetLastError();
	if (e != hipSuccess)
	{
		ctx->err = std::string("correspondence search launch: ") + hipGetErrorString(e);
		return MULLS_E_HIP;
	}
	return MULLS_OK;
}

// results of the loops that end on the device (k_icp; k_finish_step): IcpOut records -> mulls_result, profile counters
int results_from_device(Run &R, uint32_t trace_cap, bool from_icp)
{
RUN_ALIASES
	const auto wall0 = R.wall0;

		// (pinned: a copy into pageable memory goes through the runtime's staging buffers and holds the calling thread)
		if (grow_pinned(ctx, &B->icp_outs_pin, &B->cap_icp_pin, (size_t)n, hipHostMallocDefault) != MULLS_OK)
			return MULLS_E_NOMEM;
		HIPCHK(ctx, hipMemcpyAsync(B->icp_outs_pin, B->icp_outs, sizeof(IcpOut) * (size_t)n, hipMemcpyDeviceToHost, st));
		if (trace_cap)
		{
			B->trace_h.resize((size_t)n * trace_cap);
			HIPCHK(ctx, hipMemcpyAsync(B->trace_h.data(), B->trace_dev, sizeof(mulls_iter_trace) * (size_t)n * trace_cap, hipMemcpyDeviceToHost, st));
		}
		HIPCHK(ctx, hipStreamSynchronize(st));
		evt.collect();
		const double wall_ms = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count() * 1e3;
		int max_it = 0;
		for (int p = 0; p < n; p++)
		{
			const IcpOut &o = B->icp_outs_pin[p];
			mulls_result &R = results[p];
			R.code = o.code;
			R.iters = o.iters;
			std::memcpy(R.T, o.T, sizeof(R.T));
			std::memcpy(R.info, o.info, sizeof(R.info));
			R.sigma = (float)std::sqrt(o.sigma2);
			R.confidence = o.ratio;
			R.singular = o.singular;
			R.ms_total = (float)(wall_ms / n);
			for (int c = 0; c < MULLS_NC; c++)
			{
				R.ncorr[c] = o.ncorr[c];
				R.nsrc0[c] = o.nsrc0[c];
				R.ntgt0[c] = o.ntgt0[c];
			}
			R.cropped = 0;
			std::memset(R.crop_box, 0, sizeof(R.crop_box));
			fill_crop_box(rp, B->setup_h[p].tgt_bound, o.bbox, R);
			R.trace_len = 0;
			if (trace_cap && R.trace && R.trace_cap > 0)
			{
				R.trace_len = std::min(o.trace_len, R.trace_cap);
				std::memcpy(R.trace, &B->trace_h[(size_t)p * trace_cap], sizeof(mulls_iter_trace) * (size_t)R.trace_len);
			}
			ctx->prof.nn_src_pts += o.src_pts;
			ctx->prof.nn_tgt_unique += o.tgt_pts;
			ctx->prof.nn_tgt_pts += from_icp ? o.tgt_pts : o.tgt_job_pts;
			ctx->prof.nn_corr_pts += o.corr_pts;
			if (!from_icp)
				ctx->prof.nn_pair_evals += o.pair_evals;
			if (from_icp)
			{
				for (int k = 0; k < 6; k++)
					ctx->prof.icp_phase_ms[k] += (double)o.t_phase[k] * 1e-5; // 10-ns ticks -> ms (summed over the pairs)
				for (int k = 0; k < 6; k++)
					ctx->prof.icp_fused_ms[k] += (double)o.t_fused[k] * 1e-5;
				for (int k = 0; k < 24 && k < o.iters; k++)
					ctx->prof.icp_search_ms[k] += (double)o.t_search_it[k] * 1e-5;
			}
			max_it = std::max(max_it, o.iters);
		}
		ctx->prof.iterations = max_it;
		if (rp.dbg_ticks) // diagnostics: k_cert's phase clocks, summed over its workgroups, in the resident loop's slots ([5] = workgroups)
		{
			unsigned long long t[16];
			HIPCHK(ctx, hipMemcpy(t, rp.dbg_ticks, sizeof(t), hipMemcpyDeviceToHost));
			if (rp.debug_stop == 21u) // the global-memory tier's leftover search: queries, workgroup time (summed over the workgroups)
			{
				ctx->prof.icp_fused_ms[0] += (double)t[13], ctx->prof.icp_fused_ms[1] += (double)t[7] * 1e-5;
				return MULLS_OK;
			}
			for (int k = 0; k < 5; k++)
				ctx->prof.icp_fused_ms[k] += (double)t[k] * 1e-5;
			ctx->prof.icp_fused_ms[5] += (double)t[6];
			// the k-candidate certificates of the one-pass walk, in the (otherwise unused) per-iteration slots: points the plain certificate left over, points
			// that got the second chance, points it certified, points searched
			ctx->prof.icp_search_ms[0] += (double)t[13], ctx->prof.icp_search_ms[1] += (double)t[14], ctx->prof.icp_search_ms[2] += (double)t[15], ctx->prof.icp_search_ms[3] += (double)t[7];
			for (int k = 0; k < 4; k++) // ... and the heavy pass's (k_nn_lds), in the phase slots ([4] = class clouds)
				ctx->prof.icp_phase_ms[k] += (double)t[8 + k] * 1e-5;
			ctx->prof.icp_phase_ms[4] += (double)t[12];
		}
		return MULLS_OK;
}

// ---- device-resident loop: ONE launch iterates every pair to the end (k_icp.hip) ------------------------------------------------------
int run_resident(Run &R)
{
RUN_ALIASES

	// ---- device-resident loop: ONE launch iterates every pair to the end (k_icp.hip) ------------------------------------------
	uint32_t trace_cap = 0;
	for (int p = 0; p < n; p++)
		if (results[p].trace && results[p].trace_cap > 0)
			trace_cap = std::max(trace_cap, (uint32_t)results[p].trace_cap);
	if (trace_cap)
	{
		trace_cap = std::min(trace_cap, (uint32_t)std::max(P->max_iter_num, 1));
		if (grow(ctx, &B->trace_dev, &B->cap_icp[4], (size_t)n * trace_cap) != MULLS_OK)
			return MULLS_E_HIP;
	}
	evt.begin(&ctx->prof.ms_nn);
	if (launch_icp(st, (uint32_t)n, 0u, B->rjobs, B->pair_rjob, B->order, B->icp_queue, B->descs, B->setup, rp, K, B->spos, B->snrm, B->grids, B->cell_start,
				   B->tsorted, B->flag, B->nn_idx, B->nn_d2, B->winner, B->tnrm, B->match, B->wd, B->tpos, B->nn_hint, B->mq, B->bbox, lds_cap, rp.grid_maxcells,
				   B->icp_outs, trace_cap ? B->trace_dev : nullptr, trace_cap) != 0)
	{
		ctx->err = "could not raise the dynamic LDS limit of k_icp";
		return MULLS_E_HIP;
	}
	evt.end();
	const int rcr = results_from_device(R, trace_cap, true);
	if (rcr != MULLS_OK)
		return rcr;
	ctx->prof.launches_nn = 1;
	return MULLS_OK;
}

// ---- lock-step loop with the O(1) half of the iteration on the device (k_reduce.hip: k_finish / k_step / k_step_publish, or k_finish_step) -----
int run_device_step(Run &R)
{
RUN_ALIASES
	int rc;
	// One launch set per iteration: search (+ filter), accumulation, finish + step + publication.  The host keeps two sets queued and reads one
	// 8-byte word per set — (epoch << 32 | pairs still iterating) — to know when to stop queueing; a set queued behind the last useful one finds no
	// active pair and falls through.  Mid-size batches (R.nsub == 2) run as two sub-batches on two streams, each with its own sets, word and
	// ticket: their kernels are a few hundred workgroups each and bound by latency, so the chip runs one sub-batch's kernel beside the other's.
	if (grow(ctx, &B->steps, &B->cap_steps, (size_t)n) != MULLS_OK || grow(ctx, &B->icp_outs, &B->cap_icp[3], (size_t)n) != MULLS_OK)
		return MULLS_E_HIP;
	HIPCHK(ctx, hipMemsetAsync(B->icp_outs, 0, sizeof(IcpOut) * (size_t)n, st));
	launch_step_init(st, (uint32_t)n, B->setup, K, B->steps, B->states);
	// Small batches are bound by the NUMBER of launches (a kernel of a few hundred workgroups takes ~5 us whatever it does; one pair is bound by
	// the host's ~4 us per launch): the three accumulation launches become one, finish + step + publication one (k_finish_step)
	const bool few_launches = n <= (int)ctx->opt[MULLS_OPT_FEW_LAUNCHES_MAX_PAIRS];
	// one wave per pair sums and steps (k_sum_step) when no class cloud has more trips than its lanes keep in flight at once (the jobs are 512-slot blocks, a trip is
	// two of them: KITTI-sized source clouds have one or two trips per class)
	bool sum_step = ctx->opt[MULLS_OPT_SUM_STEP] != 0.0;
	for (size_t k = 0; sum_step && k < B->descs_h.size(); k++)
		sum_step = B->descs_h[k].job_end - B->descs_h[k].job_begin <= 4u * (1024u / MULLS_SRC_PER_BLOCK); // (1024 = MULLS_ACC_LANES, accum.h)
	struct Sub
	{
		int lo = 0, hi = 0;
		Slice L;
		hipStream_t st = nullptr;
		volatile unsigned long long *word = nullptr;
		unsigned long long *word_dev = nullptr;
		uint32_t *epoch_ctr = nullptr, *ticket = nullptr, *wl = nullptr, *wl_ctr = nullptr;
		uint32_t epoch0 = 0, nn_launches = 0, left = 0;
		int s = 0;
		bool done = false;
		EvTimer ev2[2] = {EvTimer{nullptr}, EvTimer{nullptr}};
	};
	const int nsub = B->nsub;
	Sub subs[2];
	for (int k = 0; k < nsub; k++)
	{
		Sub &S = subs[k];
		S.lo = (int)((long)n * k / nsub);
		S.hi = (int)((long)n * (k + 1) / nsub);
		S.L = slice_of(B, S.lo, S.hi, k);
		S.st = k == 0 ? ctx->stream : ctx->stream2;
		S.word = reinterpret_cast<volatile unsigned long long *>(B->epoch_h + 32 + 16 * k);
		S.word_dev = reinterpret_cast<unsigned long long *>(B->epoch_dev + 32 + 16 * k);
		S.epoch_ctr = k == 0 ? &B->epoch2 : &B->epoch3;
		S.epoch0 = *S.epoch_ctr;
		S.ticket = B->ticket + 2 + 4 * k;
		S.wl = B->wl + S.L.cjob_lo;
		S.wl_ctr = B->wl_ctr + 8 * k;
		S.left = (uint32_t)(S.hi - S.lo);
		for (int e = 0; e < 2; e++)
		{
			S.ev2[e].ctx = ctx;
			S.ev2[e].base = 10 * e;
		}
	}
	subs[0].ev2[0].used = evt.used; // the setup events were recorded on the first set (profiling: one sub-batch)
	for (int k = 0; k < 5; k++)
		subs[0].ev2[0].slot[k] = evt.slot[k];
	evt.used = 0;
	if (nsub == 2)
	{
		HIPCHK(ctx, hipEventRecord(ctx->ev_setup, ctx->stream));
		HIPCHK(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_setup, 0));
	}
	struct DrainOnError // an error from here on leaves kernels in flight that still write the pinned words
	{
		mulls_ctx *ctx;
		bool armed = true;
		~DrainOnError()
		{
			if (armed)
			{
				(void)hipStreamSynchronize(ctx->stream);
				(void)hipStreamSynchronize(ctx->stream2);
			}
		}
	} drain{ctx};
	// wait until launch set `set` of a sub-batch has published; S.left = its pairs still iterating after the newest published set
	auto wait_set = [&](Sub &S, int set) -> int {
		const uint32_t want = S.epoch0 + (uint32_t)set + 1u;
		const auto t0 = std::chrono::steady_clock::now();
		bool synced = false;
		for (uint64_t spins = 0;; spins++)
		{
			const unsigned long long w = *S.word;
			if ((int32_t)((uint32_t)(w >> 32) - want) >= 0)
			{
				std::atomic_thread_fence(std::memory_order_acquire);
				S.left = (uint32_t)w;
				return MULLS_OK;
			}
			if (synced)
				break;
			if ((spins & 0xfff) == 0xfff && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0)
			{
				HIPCHK(ctx, hipStreamSynchronize(S.st)); // a stalled device, or an asynchronous error: surfaces here
				synced = true;
			}
		}
		ctx->err = "device did not publish the iteration epoch";
		return MULLS_E_HIP;
	};
	// queue launch set S.s of a sub-batch (after its set S.s - 2 has published); marks the sub-batch done when nothing is left to queue
	auto advance = [&](Sub &S) -> int {
		const int s = S.s;
		if (s > P->max_iter_num) // max_iter_num iterations and the residual pass of the last pairs to finish have been queued
		{
			S.done = true;
			return MULLS_OK;
		}
		if (s >= 2)
		{
			const auto t_wait0 = std::chrono::steady_clock::now();
			int rcw;
			if ((rcw = wait_set(S, s - 2)) != MULLS_OK)
				return rcw;
			ctx->prof.ms_host_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wait0).count() * 1e3;
			S.ev2[s & 1].collect();
			if (S.left == 0)
			{
				S.done = true;
				return MULLS_OK;
			}
		}
		hipStream_t sst = S.st;
		EvTimer &ev = S.ev2[s & 1];
		ev.stream = sst;
		const bool search = s < P->max_iter_num; // the last set can only hold residual passes
		if (search)
		{
			int rcs;
			if ((rcs = launch_search(R, sst, S.L, S.wl, S.wl_ctr, S.nn_launches++, ev, s)) != MULLS_OK)
			{
				(void)hipMemsetAsync(S.ticket, 0, 2 * sizeof(uint32_t), sst); // (a failed launch set must not leave the next run an armed ticket)
				return rcs;
			}
			if (&S == &subs[0])
				ctx->prof.launches_nn++;
		}
		// a set without a search holds only posterior-residual passes (every pair ran its last iteration in the set before): that is the
		// residual kernel time; a set of a converging batch mixes both kinds of pairs and is charged to the accumulation
		ev.begin(search ? &ctx->prof.ms_accum : &ctx->prof.ms_residual);
		launch_accum(sst, B->ajobs, S.L.ajob_split, B->jobs, B->descs, B->states, rp, B->spos, B->mq, B->flag, B->wd, B->partial, few_launches, (uint32_t)ctx->opt[MULLS_OPT_ACCUM_WAVE_MIN_TRIPS]);
		launch_finish_step(sst, (uint32_t)S.lo, (uint32_t)(S.hi - S.lo), B->descs, B->states, rp, K, B->partial, B->outs, B->bbox, B->steps, B->icp_outs, S.word_dev,
						   ++*S.epoch_ctr, use_grid ? 0 : 1, (few_launches || n <= (int)ctx->opt[MULLS_OPT_STEP_LAUNCH_MAX_PAIRS]) ? S.ticket : nullptr, sum_step);
		ev.end();
		{
			const hipError_t e = hipGetLastError(); // a rejected launch (dynamic LDS size, ...) would otherwise only show as an epoch that never arrives
			if (e != hipSuccess)
			{
				ctx->err = std::string("iteration launch set: ") + hipGetErrorString(e);
				(void)hipMemsetAsync(S.ticket, 0, 2 * sizeof(uint32_t), sst);
				return MULLS_E_HIP;
			}
		}
		S.s++;
		return MULLS_OK;
	};
	const auto t_loop0 = std::chrono::steady_clock::now();
	for (;;)
	{
		bool any = false;
		for (int k = 0; k < nsub; k++)
			if (!subs[k].done)
			{
				if ((rc = advance(subs[k])) != MULLS_OK)
					return rc;
				any = true;
			}
		if (!any)
			break;
	}
	ctx->prof.ms_host_launch += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop0).count() * 1e3 - ctx->prof.ms_host_wait;
	// the result records' download is stream-ordered behind the last launch set: ONE wait for the run's end instead of a stream synchronisation, then the copy, then
	// another one (a blocked host thread takes tens of microseconds to come back: 5 % of a single registration)
	if (nsub == 2)
	{
		HIPCHK(ctx, hipEventRecord(ctx->ev_setup, ctx->stream2));
		HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_setup, 0));
	}
	rc = results_from_device(R, 0, false); // (synchronises ctx->stream, and with it — through the event — the second sub-batch's stream)
	if (rc != MULLS_OK)
		return rc;
	drain.armed = false;
	for (int k = 0; k < nsub; k++)
	{
		subs[k].ev2[0].collect();
		subs[k].ev2[1].collect();
	}
	return MULLS_OK;
}

// ---- lock-step loop stepped by the host: per-iteration traces -----------------------------------------------------------------------------------
int run_host_step(Run &R)
{
RUN_ALIASES
	const auto wall0 = R.wall0;
	int rc;

	std::vector<PairHost> H(n);
	for (int p = 0; p < n; p++)
	{
		PairHost &h = H[p];
		mulls::pair_iter_init(h, B->setup_h[p].guess, K);
		results[p].trace_len = 0;
		std::memset(results[p].ncorr, 0, sizeof(results[p].ncorr));
		std::memset(results[p].nsrc0, 0, sizeof(results[p].nsrc0));
		std::memset(results[p].ntgt0, 0, sizeof(results[p].ntgt0));
		results[p].cropped = 0;
		std::memset(results[p].crop_box, 0, sizeof(results[p].crop_box));
	}

	if (P->max_iter_num <= 0)
	{
		// the iteration loop never runs (process code 0): still report the post-filter cloud sizes
		std::vector<CloudDesc> back(B->descs_h.size());
		std::vector<uint32_t> keys((size_t)n * 6);
		HIPCHK(ctx, hipMemcpyAsync(back.data(), B->descs, sizeof(CloudDesc) * back.size(), hipMemcpyDeviceToHost, st));
		HIPCHK(ctx, hipMemcpyAsync(keys.data(), B->bbox, sizeof(uint32_t) * keys.size(), hipMemcpyDeviceToHost, st));
		HIPCHK(ctx, hipStreamSynchronize(st));
		evt.collect();
		for (int p = 0; p < n; p++)
		{
			for (int c = 0; c < MULLS_NC; c++)
			{
				results[p].nsrc0[c] = back[p * MULLS_NC + c].src_n;
				results[p].ntgt0[c] = back[p * MULLS_NC + c].tgt_n;
			}
			fill_crop_box(rp, B->setup_h[p].tgt_bound, &keys[(size_t)p * 6], results[p]);
		}
	}

	// Two sub-batches share the stream: while the device runs one sub-batch's iteration the host solves the other's 6x6
	// systems and queues its next launch set behind it, so neither side idles (a single sub-batch when the batch is small).
	// Each sub-batch has its own iteration counter, arrival ticket, epoch word and contiguous slice of the job tables.
	struct Sub
	{
		int lo = 0, hi = 0;
		Slice L;
		int iter = 0;
		bool inflight = false;
		uint32_t nn_launches = 0; // parity of the LDS tier's queue counters
		uint64_t seq = 0;
		uint32_t *epoch_ctr = nullptr;
		volatile uint32_t *word = nullptr;
		uint32_t *word_dev = nullptr, *ticket = nullptr;
		hipStream_t st = nullptr;
		EvTimer evt{nullptr};
	};
	const int nsub = subbatch_count(ctx, n);
	Sub subs[2];
	for (int k = 0; k < nsub; k++)
	{
		Sub &S = subs[k];
		S.lo = (int)((long)n * k / nsub);
		S.hi = (int)((long)n * (k + 1) / nsub);
		S.L = slice_of(B, S.lo, S.hi, k);
		S.epoch_ctr = k == 0 ? &B->epoch : &B->epoch1;
		S.word = B->epoch_h + 16 * k;
		S.word_dev = B->epoch_dev + 16 * k;
		S.ticket = B->ticket + 16 * k;
		S.evt.ctx = ctx;
		S.evt.base = 10 * k;
		S.st = ctx->stream;
	}
	// two streams: the second sub-batch's filter / accumulate kernels (latency-bound, few registers and no LDS to speak of)
	// run under the first one's search (issue-bound, one workgroup per CU) and vice versa
	// Opt-in (MULLS_TWO_STREAMS=1): measured +4 % registrations/s at 4096 pairs, but the two searches then share the CUs and
	// every kernel's own duration doubles, which would blur the per-kernel accounting bench.py and the profiles report.
	const bool two_streams = nsub == 2 && ctx->opt[MULLS_OPT_TWO_STREAMS] != 0.0;
	if (two_streams)
	{
		subs[1].st = ctx->stream2;
		subs[1].evt.stream = ctx->stream2;
		HIPCHK(ctx, hipEventRecord(ctx->ev_setup, ctx->stream));
		HIPCHK(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_setup, 0));
	}
	// the setup events were recorded on sub-batch 0's set
	subs[0].evt.used = evt.used;
	for (int k = 0; k < 5; k++)
		subs[0].evt.slot[k] = evt.slot[k];
	evt.used = 0;
	uint64_t launch_seq = 0;

	// queue one iteration (search, filter, accumulation, publication) of a sub-batch; 0 = nothing left to do for it
	auto launch = [&](Sub &S) -> int {
		hipStream_t st = S.st;
		bool any_active = false, any_resid = false;
		for (int p = S.lo; p < S.hi; p++)
		{
			any_active |= H[p].active;
			any_resid |= H[p].want_residual;
		}
		S.inflight = false;
		if (!any_active && !any_resid)
			return MULLS_OK;
		const auto t_launch0 = std::chrono::steady_clock::now();
		for (int p = S.lo; p < S.hi; p++)
		{
			PairState &s = B->states_h[p];
			const PairHost &h = H[p];
			for (int r = 0; r < 3; r++)
				for (int c = 0; c < 4; c++)
					s.T[r * 4 + c] = h.temp.at(r, c);
			std::memcpy(s.x, h.x, sizeof(s.x));
			std::memcpy(s.thr, h.thr, sizeof(s.thr));
			s.iter = h.want_residual ? h.iters - 1 : S.iter;
			s.active = h.active ? 1 : 0;
			s.want_residual = h.want_residual ? 1 : 0;
			s.pad_[0] = s.pad_[1] = s.pad_[2] = 0;
		}
		EvTimer &ev = S.evt;
		launch_push_states(st, B->states_pin + S.lo, B->states + S.lo, (uint32_t)(S.hi - S.lo));
		if (any_active)
		{
			int rcs;
			if ((rcs = launch_search(R, st, S.L, B->wl + S.L.cjob_lo, B->wl_ctr + 8 * (int)(&S - subs), S.nn_launches++, ev, S.iter)) != MULLS_OK)
				return rcs;
			ctx->prof.launches_nn++;
			if (&S == &subs[0])
				ctx->prof.iterations++;
		}
		ev.begin(any_active ? &ctx->prof.ms_accum : &ctx->prof.ms_residual);
		launch_accum(st, B->ajobs, S.L.ajob_split, B->jobs, B->descs, B->states, rp, B->spos, B->mq, B->flag, B->wd, B->partial, false, (uint32_t)ctx->opt[MULLS_OPT_ACCUM_WAVE_MIN_TRIPS]);
		launch_finish(st, (uint32_t)(S.hi - S.lo), B->descs, B->states, rp, B->partial, B->outs, B->outs_pin, B->bbox, S.ticket, S.word_dev, ++*S.epoch_ctr,
					  (uint32_t)S.lo);
		ev.end();
		S.inflight = true;
		S.seq = ++launch_seq;
		ctx->prof.ms_host_launch += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_launch0).count() * 1e3;
		return MULLS_OK;
	};

	// the host half of one iteration for a sub-batch whose sums have been published
	auto host_step = [&](Sub &S) {
		const auto t_step0 = std::chrono::steady_clock::now();
		uint64_t acc_evals = 0, acc_src = 0, acc_tgt = 0, acc_tgtu = 0;
		const int host_threads = std::max(1, std::min(16, (S.hi - S.lo) / 32));
		(void)host_threads;
#pragma omp parallel for num_threads(host_threads) schedule(static) reduction(+ : acc_evals, acc_src, acc_tgt, acc_tgtu) if (host_threads > 1)
		for (int p = S.lo; p < S.hi; p++)
		{
			PairHost &h = H[p];
			PairOut o;
			unpack_out(B, rp.used, p, o, true);
			mulls_result &R = results[p];
			if (h.want_residual)
			{
				// get_multi_metrics_lls_residual (cregistration.hpp:2518-2544) + information matrix (:1386); VTPV and the number of
				// observations were summed over the used classes in the reference's order by k_finish
				mulls::step_residual(h, K, o.comb[0], o.comb[1]);
				continue;
			}
			if (!h.active)
				continue;
			const int i = S.iter;
			h.iters = i + 1;
			if (h.first)
			{
				for (int c = 0; c < MULLS_NC; c++)
				{
					// while undistorting, the sizes the reference counts at :1195-1201 are those of the cloned clouds,
					// before the five non-vertex clouds are regenerated from block2->pc_*_down inside the loop
					R.nsrc0[c] = rp.undistort ? B->descs_h[p * MULLS_NC + c].src_n0 : o.src_n[c];
					R.ntgt0[c] = o.tgt_n[c];
					h.alive_prev[c] = o.src_n[c];
				}
				fill_crop_box(rp, B->setup_h[p].tgt_bound, o.bbox, R);
				h.src_feature_count = 0; // cregistration.hpp:1195-1201
				if (rp.used[1])
					h.src_feature_count += (int)R.nsrc0[MULLS_PILLAR];
				if (rp.used[2])
					h.src_feature_count += (int)R.nsrc0[MULLS_FACADE];
				if (rp.used[3])
					h.src_feature_count += (int)R.nsrc0[MULLS_BEAM];
				h.first = false;
			}
			for (int c = 0; c < MULLS_NC; c++)
			{
				if (rp.used[c] && h.alive_prev[c] >= 3 && o.tgt_n[c] >= 3)
				{
					if (!use_grid)
						acc_evals += (uint64_t)h.alive_prev[c] * o.tgt_n[c];
					acc_src += h.alive_prev[c];
					acc_tgtu += o.tgt_n[c];
					acc_tgt += (uint64_t)o.tgt_n[c] * (B->descs_h[p * MULLS_NC + c].job_end - B->descs_h[p * MULLS_NC + c].job_begin);
				}
				h.alive_prev[c] = o.n_alive[c];
				R.ncorr[c] = o.n_valid[c];
			}
			mulls_iter_trace *tr = nullptr;
			if (R.trace && R.trace_len < R.trace_cap)
			{
				tr = &R.trace[R.trace_len++];
				std::memset(tr, 0, sizeof(*tr));
				tr->iter = i;
				for (int c = 0; c < MULLS_NC; c++)
				{
					tr->ncorr[c] = o.n_valid[c];
					tr->nsrc[c] = o.n_alive[c];
					tr->thr[c] = h.thr[c];
				}
			}
			if (!mulls::step_counts(h, K, o.n_valid)) // :1305-1311, then update_corr_dist_thre :1855-1866
				continue;
			Mat6 N;
			double b[6];
			mulls::normal_from_row(o.comb, N, b);
			mulls::step_solve(h, K, N, b, i); // solve :1924-1964, step test :1348-1354, convergence :1357, guess update :1400
			if (tr)
			{
				std::memcpy(tr->atpa, N.v, sizeof(tr->atpa));
				std::memcpy(tr->atpb, b, sizeof(tr->atpb));
				std::memcpy(tr->x, h.x, sizeof(tr->x));
			}
		}
		ctx->prof.ms_host_step += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_step0).count() * 1e3;
		ctx->prof.nn_pair_evals += acc_evals;
		ctx->prof.nn_src_pts += acc_src;
		ctx->prof.nn_tgt_pts += acc_tgt;
		ctx->prof.nn_tgt_unique += acc_tgtu;
		S.iter++;
	};

	// an error from here on leaves kernels in flight that still write the pinned result / epoch buffers: drain both streams
	// before the caller can refill or free them
	struct DrainOnError
	{
		mulls_ctx *ctx;
		bool armed = true;
		~DrainOnError()
		{
			if (armed)
			{
				(void)hipStreamSynchronize(ctx->stream);
				(void)hipStreamSynchronize(ctx->stream2);
			}
		}
	} drain{ctx};
	for (int k = 0; k < nsub; k++)
		if ((rc = launch(subs[k])) != MULLS_OK)
			return rc;
	for (;;)
	{
		// whichever sub-batch in flight publishes first (one stream: the one queued first; two streams: either)
		bool any = false;
		for (int k = 0; k < nsub; k++)
			any |= subs[k].inflight;
		if (!any)
			break;
		const auto t_wait0 = std::chrono::steady_clock::now();
		Sub *next = nullptr;
		for (uint64_t spins = 0; !next; spins++)
		{
			for (int k = 0; k < nsub && !next; k++)
				if (subs[k].inflight && *subs[k].word == *subs[k].epoch_ctr)
					next = &subs[k];
			if (!next && (spins & 0xfff) == 0xfff && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wait0).count() > 2.0)
			{
				for (int k = 0; k < nsub; k++) // something is wrong: fall back to a blocking wait on the oldest launch
					if (subs[k].inflight && (!next || subs[k].seq < next->seq))
						next = &subs[k];
			}
		}
		if (wait_epoch_word(ctx, next->word, *next->epoch_ctr, next->evt.last(), next->st) != MULLS_OK)
			return MULLS_E_HIP;
		ctx->prof.ms_host_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wait0).count() * 1e3;
		next->evt.collect();
		host_step(*next);
		if ((rc = launch(*next)) != MULLS_OK)
			return rc;
	}

	HIPCHK(ctx, hipStreamSynchronize(st));
	if (two_streams)
		HIPCHK(ctx, hipStreamSynchronize(ctx->stream2));
	drain.armed = false;
	const double wall_ms = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count() * 1e3;
	for (int p = 0; p < n; p++)
	{
		PairHost &h = H[p];
		mulls_result &R = results[p];
		h.guess = h.temp * h.guess; // :1403
		R.code = h.code;
		R.iters = h.iters;
		std::memcpy(R.T, h.guess.v, sizeof(R.T));
		std::memcpy(R.info, h.info.v, sizeof(R.info));
		R.sigma = (float)std::sqrt(h.sigma2);
		R.confidence = h.ratio;
		R.singular = h.singular;
		R.ms_total = (float)(wall_ms / n);
	}
	return MULLS_OK;
}
} // namespace

extern "C"
{
	int mulls_batch_run(mulls_ctx *ctx, mulls_batch *B, const mulls_params *P, mulls_result *results)
	try
	{
		if (!ctx || !B || !results)
			return MULLS_E_INVALID;
		int rc = check_params(ctx, P);
		if (rc != MULLS_OK)
			return rc;
		HIPCHK(ctx, hipSetDevice(ctx->device));
		ctx->prof = mulls_profile{};
		Run R{ctx, B, P, results, B->n, ctx->stream, std::chrono::steady_clock::now(), EvTimer{ctx}};
		if ((rc = run_setup(R)) != MULLS_OK)
			return rc;
		if (R.resident)
			return run_resident(R);
		if (R.dstep)
			return run_device_step(R);
		return run_host_step(R);
	}
	catch (...)
	{
		return mulls::abi_caught(const_cast<mulls_ctx *>(ctx)); // nothing is thrown across the ABI
	}
}
