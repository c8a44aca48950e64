// variants.cpp — the two variants of the path built on the same kernels: lls_icp_3dof_ground (cregistration.hpp:1443-1582)
// and mm_lls_icp_4dof_global (:1584-1681).
#include "batch.h"

using namespace mulls_drv;

extern "C"
{
	// variants of the path

	// lls_icp_3dof_ground (cregistration.hpp:1443-1582): ground class only, unknowns (roll, pitch, z); same kernels, the
	// 3x3 system is a sub-block of the point-to-plane accumulation (rows/columns a, b, ntz — identical float products).
	int mulls_icp_3dof_ground_batch(mulls_ctx *ctx, const mulls_pair *pairs, int n, const mulls_params *P, mulls_result *results)
	try
	{
		if (!ctx || !pairs || n <= 0 || !results)
			return MULLS_E_INVALID;
		int rc = check_params(ctx, P);
		if (rc != MULLS_OK)
			return rc;
		if (!ctx->scratch)
			ctx->scratch = new mulls_batch();
		mulls_batch *B = ctx->scratch;
		rc = batch_fill(ctx, B, pairs, n);
		if (rc != MULLS_OK)
			return rc;
		hipStream_t st = ctx->stream;
		ctx->prof = mulls_profile{};
		RunParams rp;
		std::memset(&rp, 0, sizeof(rp));
		rp.used[MULLS_GROUND] = 1;
		rp.w_resid = P->weight_strategy[1] == '1';
		rp.w_dist = P->weight_strategy[2] == '1';
		rp.w_inten = P->weight_strategy[3] == '1';
		rp.faithful = 1;
		rp.rej_strict = P->rejector_strict != 0;
		rp.win_pl = rp.win_li = rp.win_pt = 0.1f;			  // residual_window_size default of pt2pl_ground_3dof_lls_summation (:2323)
		rp.cos_bearing = std::cos(40.0f / 180.0 * M_PI); // determine_corres' default angle_thre_degree (:1704)
		rp.resid_from_iter = -1;							  // no iteration gate in ground_3dof_lls_tran_estimation (:2294)
		init_cert(ctx, rp);
		if ((rc = take_epochs(ctx, B, (uint32_t)std::max(P->max_iter_num, 0) + 2u, rp)) != MULLS_OK)
			return rc;
		mulls_params Pj = *P;
		std::memset(Pj.used_feature_type, 0, sizeof(Pj.used_feature_type));
		std::strcpy(Pj.used_feature_type, "100000");
		uint32_t lds_cap = 0;
		int tier = 0;
		rc = prepare_run(ctx, B, &Pj, rp, &lds_cap, &tier);
		if (rc != MULLS_OK)
			return rc;
		launch_clone_src(st, (uint32_t)B->setup_jobs_h.size(), B->setup_jobs, B->descs, B->setup, B->stage, B->tmp_pos, B->tmp_nrm, B->bbox, rp);
		launch_crop(st, (uint32_t)n, B->descs, B->setup, B->bbox, B->stage, B->tmp_pos, B->tmp_nrm, B->spos, B->snrm, B->tpos, B->tnrm, B->flag,
					B->match, B->wd, rp, B->grids, (uint32_t)B->big_segs_h.size(), B->big_segs, (uint32_t)B->big_clouds_h.size(), B->big_clouds, B->seg_cnt,
					B->big_box);
		if (P->keep_less_source_points)
		{
			// random_downsample_pcl(pc_ground_sc, tc.size() / down_rate), down_rate = 3 (:1462, :1485): no filter ran, sizes are known
			std::vector<uint8_t> skeep(std::max<size_t>(B->n_src, 1), 1), tkeep(std::max<size_t>(B->n_tgt, 1), 1);
			for (int p = 0; p < n; p++)
			{
				const CloudDesc &dg = B->descs_h[(size_t)p * MULLS_NC + MULLS_GROUND];
				thin_mask(skeep.data() + dg.src_off, dg.src_n0, (int)(dg.tgt_n0 / 3), P->rng_seed, 1 * 6 + MULLS_GROUND);
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
				e = hipStreamSynchronize(st);
			}
			(void)hipFree(d_sk);
			(void)hipFree(d_tk);
			if (e != hipSuccess)
			{
				ctx->err = std::string("3dof keep_less_source_points: ") + hipGetErrorString(e);
				return MULLS_E_HIP;
			}
		}
		if (tier == 2)
			launch_grid_build_sort(st, (uint32_t)n, B->descs, B->grids, rp, B->tpos, B->cell_start, B->tsorted);
		launch_bm_build(st, (uint32_t)B->lclouds_h.size(), B->lclouds, (uint32_t)B->tjobs_h.size(), B->tjobs, B->descs, B->grids, B->tpos, B->bm, B->pf, B->cell_cnt, B->bm_cs,
						B->tsorted, B->bm_rank);

		struct H3
		{
			Mat4 s2t = Mat4::identity(), temp = Mat4::identity();
			float thr;
			int code = 0, iters = 0;
			bool active = true;
		};
		std::vector<H3> H(n);
		const float max_bearable_translation = (float)(2.0 * P->dis_thre_unit);
		const float converge_rotation = (float)(P->converge_rotation_d / 180.0 * M_PI);
		const float max_bearable_rotation = (float)(P->max_bearable_rotation_d / 180.0 * M_PI);
		for (int p = 0; p < n; p++)
		{
			H[p].thr = P->dis_thre_unit;
			H[p].active = P->max_iter_num > 0;
			results[p].trace_len = 0;
			std::memset(results[p].ncorr, 0, sizeof(results[p].ncorr));
			std::memset(results[p].nsrc0, 0, sizeof(results[p].nsrc0));
			std::memset(results[p].ntgt0, 0, sizeof(results[p].ntgt0));
		}
		for (int it = 0;; it++)
		{
			bool any = false;
			for (int p = 0; p < n; p++)
				any |= H[p].active;
			if (!any)
				break;
			for (int p = 0; p < n; p++)
			{
				PairState &s = B->states_h[p];
				std::memset(&s, 0, sizeof(s));
				for (int r = 0; r < 3; r++)
					for (int c = 0; c < 4; c++)
						s.T[r * 4 + c] = H[p].temp.at(r, c);
				for (int c = 0; c < MULLS_NC; c++)
					s.thr[c] = H[p].thr;
				s.iter = it;
				s.active = H[p].active ? 1 : 0;
			}
			launch_push_states(st, B->states_pin, B->states, (uint32_t)n);
			if (tier == 2)
			{
				if (launch_nn_lds(st, (uint32_t)B->cjobs_h.size(), B->cjobs, B->descs, B->states, rp, B->spos, B->snrm, B->grids, B->cell_start, B->tsorted, B->flag,
								  B->nn_idx, B->nn_d2, B->winner, B->tnrm, B->match, B->wd, B->tpos, B->nn_hint, B->mq, lds_cap, rp.grid_maxcells, B->wl, B->wl_ctr,
							  (uint32_t)it) != 0)
					return MULLS_E_HIP;
			}
			else if (tier == 1)
				launch_cert_big(st, (uint32_t)B->bjobs_h.size(), B->bjobs, 2048u, B->descs, B->states, rp, B->spos, B->snrm, B->grids, B->bm, B->pf, B->bm_cs, B->tsorted, B->flag,
								B->nn_idx, B->nn_d2, B->winner, B->tpos, B->tnrm, B->nn_hint, B->match, B->wd, B->mq);
			else
				launch_nn(st, B->njobs, B->jobs, B->descs, B->states, rp, B->spos, B->snrm, B->tpos, B->flag, B->nn_idx, B->nn_d2, B->winner);
			if (!rp.lds_dedup)
				launch_filter(st, B->njobs, B->jobs, B->descs, B->states, rp, B->snrm, B->tnrm, B->flag, B->nn_idx, B->nn_d2, B->match, B->wd, B->winner, B->tpos, B->mq);
			for (int k = 0; k < B->nsub; k++)
				launch_accum(st, B->ajobs, B->ajob_split[k], B->jobs, B->descs, B->states, rp, B->spos, B->mq, B->flag, B->wd, B->partial);
			launch_finish(st, (uint32_t)n, B->descs, B->states, rp, B->partial, B->outs, B->outs_pin, B->bbox, B->ticket, B->epoch_dev, ++B->epoch, 0);
			if (wait_epoch(ctx, B) != MULLS_OK)
				return MULLS_E_HIP;
			for (int p = 0; p < n; p++)
			{
				H3 &h = H[p];
				if (!h.active)
					continue;
				PairOut o;
				unpack_out(B, rp.used, p, o);
				mulls_result &R = results[p];
				h.iters = it + 1;
				if (it == 0)
				{
					R.nsrc0[MULLS_GROUND] = o.src_n[MULLS_GROUND];
					R.ntgt0[MULLS_GROUND] = o.tgt_n[MULLS_GROUND];
				}
				R.ncorr[MULLS_GROUND] = o.n_valid[MULLS_GROUND];
				mulls_iter_trace *tr = nullptr;
				if (R.trace && R.trace_len < R.trace_cap)
				{
					tr = &R.trace[R.trace_len++];
					std::memset(tr, 0, sizeof(*tr));
					tr->iter = it;
					tr->ncorr[0] = o.n_valid[MULLS_GROUND];
					tr->nsrc[0] = o.n_alive[MULLS_GROUND];
					tr->thr[0] = h.thr;
				}
				if ((int)o.n_valid[MULLS_GROUND] < 100) // min_total_corr_num (:1461, :1509)
				{
					h.code = -2;
					h.active = false;
					continue;
				}
				{
					const double v = 1.0 * h.thr / P->dis_thre_update_rate;
					h.thr = (float)((v > P->dis_thre_min) ? v : (double)P->dis_thre_min);
				}
				// ATPA (col-major 3x3) over (a, b, ntz) and ATPb, cut out of the packed 6x6 terms of the ground class
				const double *g = o.sums[MULLS_GROUND];
				const double aa = g[packed(3, 3)], ab = g[packed(3, 4)], an = g[packed(2, 3)], bb = g[packed(4, 4)], bn = g[packed(2, 4)],
							 nn = g[packed(2, 2)];
				const double A[9] = {aa, ab, an, ab, bb, bn, an, bn, nn}, b3[3] = {g[21 + 3], g[21 + 4], g[21 + 2]};
				double Ainv[9], x3[3];
				mulls::invert3(A, Ainv);
				for (int r = 0; r < 3; r++)
					x3[r] = (Ainv[r] * b3[0] + Ainv[r + 3] * b3[1]) + Ainv[r + 6] * b3[2];
				const double x6[6] = {0, 0, x3[2], x3[0], x3[1], 0}; // construct_trans_a(0, 0, z, roll, pitch, 0) (:1525)
				h.temp = mulls::euler_step_to_matrix(x6);
				if (tr)
				{
					std::memcpy(tr->x, x6, sizeof(x6));
					for (int k = 0; k < 9; k++)
						tr->atpa[k] = A[k];
					std::memcpy(tr->atpb, b3, sizeof(b3));
				}
				const double tsn = std::sqrt(h.temp.at(0, 3) * h.temp.at(0, 3) + h.temp.at(1, 3) * h.temp.at(1, 3) + h.temp.at(2, 3) * h.temp.at(2, 3));
				const double rsa = mulls::rotation_angle(h.temp);
				if (tsn > max_bearable_translation || std::fabs(rsa) > max_bearable_rotation)
				{
					h.code = -1;
					h.temp = Mat4::identity();
					h.active = false;
					continue;
				}
				if (it == P->max_iter_num - 1 || (it > 2 && tsn < P->converge_translation && std::fabs(rsa) < converge_rotation))
				{
					h.code = 1;
					h.active = false;
					continue;
				}
				h.s2t = h.temp * h.s2t; // the source itself is moved by the fused transform of the next search launch
			}
		}
		HIPCHK(ctx, hipStreamSynchronize(st));
		for (int p = 0; p < n; p++)
		{
			H3 &h = H[p];
			mulls_result &R = results[p];
			Mat4 guess;
			std::memcpy(guess.v, pairs[p].init_guess, sizeof(guess.v));
			h.s2t = h.temp * h.s2t;		// :1563
			const Mat4 T = h.s2t * guess; // :1566
			R.code = h.code;
			R.iters = h.iters;
			std::memcpy(R.T, T.v, sizeof(R.T));
			const Mat6 I6 = Mat6::identity(); // information matrix, sigma and confidence are not outputs of this variant
			std::memcpy(R.info, I6.v, sizeof(R.info));
			R.sigma = 3.402823466e+38f;
			R.confidence = 0.0f;
			R.singular = 0;
			R.cropped = 0;
			std::memset(R.crop_box, 0, sizeof(R.crop_box));
			R.ms_total = 0.0f;
		}
		return MULLS_OK;
	}
	catch (...)
	{
		return mulls::abi_caught(const_cast<mulls_ctx *>(ctx)); // nothing is thrown across the ABI
	}

	int mulls_icp_3dof_ground(mulls_ctx *ctx, const mulls_pair *pair, const mulls_params *params, mulls_result *result)
	try
	{
		return mulls_icp_3dof_ground_batch(ctx, pair, 1, params, result);
	}
	catch (...)
	{
		return mulls::abi_caught(const_cast<mulls_ctx *>(ctx)); // nothing is thrown across the ABI
	}

	// mm_lls_icp_4dof_global (cregistration.hpp:1584-1681): the heading trials are independent registrations that share
	// one target — one lock-step batch.
	int mulls_icp_4dof_global(mulls_ctx *ctx, const mulls_pair *pair, float heading_step_d, const double station[3], int max_iter_num,
							  float dis_thre_unit, float converge_translation, float converge_rotation_d, float dis_thre_min,
							  float dis_thre_update_rate, float max_bearable_rotation_d, mulls_result *result, int *success, float *best_heading_d)
	try
	{
		(void)converge_rotation_d;		// the reference passes converge_translation in its place (:1640-1642) ...
		(void)max_bearable_rotation_d; // ... and never uses this one
		if (!ctx || !pair || !station || !result || !(heading_step_d > 0.0f))
			return MULLS_E_INVALID;
		std::vector<mulls_pair> trials;
		std::vector<float> headings;
		float heading_d = 0.0f;
		while (heading_d < 360.0)
		{
			const float heading_rad = (float)(heading_d * M_PI / 180.0);
			Mat4 rot = Mat4::identity(), g2s = Mat4::identity(), s2g = Mat4::identity();
			rot.at(0, 0) = std::cos(heading_rad); // float overloads, like the unqualified calls under `using namespace std` upstream
			rot.at(0, 1) = std::sin(heading_rad);
			rot.at(1, 0) = -std::sin(heading_rad);
			rot.at(1, 1) = std::cos(heading_rad);
			for (int k = 0; k < 3; k++)
			{
				g2s.at(k, 3) = -station[k];
				s2g.at(k, 3) = station[k];
			}
			const Mat4 guess = (s2g * rot) * g2s;
			mulls_pair t = *pair;
			std::memcpy(t.init_guess, guess.v, sizeof(guess.v));
			trials.push_back(t);
			headings.push_back(heading_d);
			heading_d += heading_step_d;
			if (trials.size() > 100000)
				return MULLS_E_INVALID;
		}
		mulls_params P;
		mulls_default_params(&P);
		P.max_iter_num = max_iter_num;
		P.dis_thre_unit = dis_thre_unit;
		P.converge_translation = converge_translation;
		P.converge_rotation_d = converge_translation;
		P.dis_thre_min = dis_thre_min;
		P.dis_thre_update_rate = dis_thre_update_rate;
		std::strcpy(P.used_feature_type, "111110");
		std::strcpy(P.weight_strategy, "1001");
		std::vector<mulls_result> rs(trials.size());
		std::memset(rs.data(), 0, sizeof(mulls_result) * rs.size());
		const int rc = mulls_icp_batch(ctx, trials.data(), (int)trials.size(), &P, rs.data());
		if (rc != MULLS_OK)
			return rc;
		float best_score = 0.0f, best_heading = 0.0f;
		int best = -1;
		bool ok = false;
		for (size_t i = 0; i < rs.size(); i++)
			if (rs[i].code > 0)
			{
				const float score = rs[i].confidence / rs[i].sigma;
				if (score > best_score)
				{
					best = (int)i;
					best_score = score;
					best_heading = headings[i];
				}
				ok = true;
			}
		mulls_iter_trace *keep_trace = result->trace;
		const int keep_cap = result->trace_cap;
		if (best >= 0)
			*result = rs[best];
		else
		{
			std::memset(result, 0, sizeof(*result));
			const Mat4 I4 = Mat4::identity();
			const Mat6 I6 = Mat6::identity();
			std::memcpy(result->T, I4.v, sizeof(result->T));
			std::memcpy(result->info, I6.v, sizeof(result->info));
			result->sigma = 3.402823466e+38f;
		}
		result->trace = keep_trace;
		result->trace_cap = keep_cap;
		result->trace_len = 0;
		result->iters = (int)rs.size(); // number of heading trials
		if (success)
			*success = ok ? (best >= 0 ? 1 : 2) : 0; // 2: trials succeeded but none scored above 0 (e.g. a NaN sigma) — the reference then returns true
													 // and leaves registration_con untouched (:1645-1657)
		if (best_heading_d)
			*best_heading_d = best_heading;
		return MULLS_OK;
	}
	catch (...)
	{
		return mulls::abi_caught(const_cast<mulls_ctx *>(ctx)); // nothing is thrown across the ABI
	}

	// ------------------------------------------------------------------------------------------------------------
}
