// stage.cpp — stage-level entry points of the C ABI (mulls_stage_*): single stages of the path on caller clouds, through the very kernels the loops
// launch; used by the parity tests.
#include "batch.h"

using namespace mulls_drv;

extern "C"
{
	// stage-level entry points
	int mulls_stage_transform(mulls_ctx *ctx, void *pts, uint32_t n, uint32_t stride, const double T[16])
	try
	{
		if (!ctx || (n && !pts) || stride != MULLS_POINT_BYTES || !T)
			return MULLS_E_INVALID;
		HIPCHK(ctx, hipSetDevice(ctx->device));
		float4 *d = nullptr;
		double *dT = nullptr;
		double t12[12];
		rows12(T, t12);
		if (dmalloc(ctx, &d, (size_t)n * 3) != MULLS_OK || dmalloc(ctx, &dT, 12) != MULLS_OK)
			return MULLS_E_HIP;
		hipError_t e = hipMemcpyAsync(d, pts, (size_t)n * MULLS_POINT_BYTES, hipMemcpyHostToDevice, ctx->stream);
		if (e == hipSuccess)
			e = hipMemcpyAsync(dT, t12, sizeof(t12), hipMemcpyHostToDevice, ctx->stream);
		if (e == hipSuccess)
		{
			launch_transform_aos(ctx->stream, d, n, dT);
			e = hipMemcpyAsync(pts, d, (size_t)n * MULLS_POINT_BYTES, hipMemcpyDeviceToHost, ctx->stream);
		}
		if (e == hipSuccess)
			e = hipStreamSynchronize(ctx->stream);
		(void)hipFree(d);
		(void)hipFree(dT);
		if (e != hipSuccess)
		{
			ctx->err = hipGetErrorString(e);
			return MULLS_E_HIP;
		}
		return MULLS_OK;
	}
	catch (...)
	{
		return mulls::abi_caught(const_cast<mulls_ctx *>(ctx)); // nothing is thrown across the ABI
	}

	// ---- motion compensation of a frame's clouds after its registration (test/mulls_slam.cpp:703-712) ----------------------------------------------------
	int mulls_motion_compensate(mulls_ctx *ctx, void *pts, uint32_t n, uint32_t stride, const double Tran[16], float s_ambiguous_thre)
	try
	{
		if (!ctx || (n && !pts) || stride != MULLS_POINT_BYTES || !Tran)
			return MULLS_E_INVALID;
		if (!n)
			return MULLS_OK;
		HIPCHK(ctx, hipSetDevice(ctx->device));
		Mat4 T;
		std::memcpy(T.v, Tran, sizeof(T.v));
		double q[4];
		mulls::rotation_quaternion(T, q); // Eigen::Quaterniond(Tran.block<3, 3>(0, 0)), cfilter.hpp:476
		const double t[3] = {T.at(0, 3), T.at(1, 3), T.at(2, 3)};
		// where do the records live?  A cloud of the library (mulls_block_cloud / mulls_map_cloud) or any other device allocation of the caller's runs in place; host
		// memory (pageable or pinned) makes the round trip.  (Round 4 treated every pointer it did not own as host memory: a caller's own hipMalloc buffer got an
		// invalid copy kind — advisor.)
		bool on_device = mulls_is_map_memory(ctx, pts, (size_t)n * MULLS_POINT_BYTES);
		if (!on_device)
		{
			hipPointerAttribute_t at;
			std::memset(&at, 0, sizeof(at));
			if (hipPointerGetAttributes(&at, pts) == hipSuccess)
				on_device = at.type == hipMemoryTypeDevice;
			else
				(void)hipGetLastError(); // (an ordinary host pointer: the query reports an error on some runtimes — cleared)
		}
		if (on_device)
		{
			// in place, nothing crosses PCIe
			launch_motion_comp(ctx->stream, static_cast<float4 *>(pts), n, q, t, s_ambiguous_thre);
			HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
			return MULLS_OK;
		}
		float4 *d = nullptr;
		if (dmalloc(ctx, &d, (size_t)n * 3) != MULLS_OK)
			return MULLS_E_HIP;
		hipError_t e = hipMemcpyAsync(d, pts, (size_t)n * MULLS_POINT_BYTES, hipMemcpyHostToDevice, ctx->stream);
		if (e == hipSuccess)
		{
			launch_motion_comp(ctx->stream, d, n, q, t, s_ambiguous_thre);
			e = hipMemcpyAsync(pts, d, (size_t)n * MULLS_POINT_BYTES, hipMemcpyDeviceToHost, ctx->stream);
		}
		if (e == hipSuccess)
			e = hipStreamSynchronize(ctx->stream);
		(void)hipFree(d);
		if (e != hipSuccess)
		{
			ctx->err = hipGetErrorString(e);
			return MULLS_E_HIP;
		}
		return MULLS_OK;
	}
	catch (...)
	{
		return mulls::abi_caught(const_cast<mulls_ctx *>(ctx)); // nothing is thrown across the ABI
	}

	int mulls_block_motion_compensate(mulls_ctx *ctx, mulls_block *block, const double Tran[16], int undistort_keypoints)
	try
	{
		if (!ctx || !block || !Tran)
			return MULLS_E_INVALID;
		if (std::find(ctx->blocks.begin(), ctx->blocks.end(), block) == ctx->blocks.end())
		{
			ctx->err = "mulls_block_motion_compensate: the block does not belong to this context";
			return MULLS_E_INVALID;
		}
		HIPCHK(ctx, hipSetDevice(ctx->device));
		Mat4 T;
		std::memcpy(T.v, Tran, sizeof(T.v));
		double q[4];
		mulls::rotation_quaternion(T, q);
		const double t[3] = {T.at(0, 3), T.at(1, 3), T.at(2, 3)};
		// batch_apply_motion_compensation (cfilter.hpp:519-531) on the five class clouds, then on their *_down clouds, as test/mulls_slam.cpp:706-710 calls it;
		// pc_vertex rides in both calls and moves only with undistort_keypoints (twice then, as upstream would)
		static const int clouds[] = {MULLS_EX_GROUND, MULLS_EX_PILLAR + MULLS_CL_PILLAR, MULLS_EX_PILLAR + MULLS_CL_FACADE, MULLS_EX_PILLAR + MULLS_CL_BEAM,
									 MULLS_EX_PILLAR + MULLS_CL_ROOF, MULLS_EX_GROUND_DOWN, MULLS_EX_PILLAR + MULLS_CL_PILLAR_DOWN, MULLS_EX_PILLAR + MULLS_CL_FACADE_DOWN,
									 MULLS_EX_PILLAR + MULLS_CL_BEAM_DOWN, MULLS_EX_PILLAR + MULLS_CL_ROOF_DOWN};
		for (int which : clouds)
			if (block->n[which])
				launch_motion_comp(ctx->stream, reinterpret_cast<float4 *>(block->buf + block->off[which]), block->n[which], q, t, 0.0f);
		for (int rep = 0; rep < (undistort_keypoints ? 2 : 0); rep++)
			if (block->n[MULLS_EX_VERTEX])
				launch_motion_comp(ctx->stream, reinterpret_cast<float4 *>(block->buf + block->off[MULLS_EX_VERTEX]), block->n[MULLS_EX_VERTEX], q, t, 0.0f);
		HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
		return MULLS_OK;
	}
	catch (...)
	{
		return mulls::abi_caught(const_cast<mulls_ctx *>(ctx)); // nothing is thrown across the ABI
	}

	namespace
	{
	// one-pair, one-class batch with identity guess and no intersection filter; leaves the batch set up (clone + crop run)
	int stage_batch(mulls_ctx *ctx, int cls, const mulls_cloud *src, const mulls_cloud *tgt, mulls_batch **out, RunParams *rp,
					const char *used6)
	{
		mulls_pair pr;
		std::memset(&pr, 0, sizeof(pr));
		pr.src[cls] = *src;
		pr.tgt[cls] = *tgt;
		for (int k = 0; k < 4; k++)
			pr.init_guess[5 * k] = 1.0;
		int rc = mulls_batch_create(ctx, &pr, 1, out);
		if (rc != MULLS_OK)
			return rc;
		mulls_batch *B = *out;
		mulls_params P;
		mulls_default_params(&P);
		std::strcpy(P.used_feature_type, used6);
		hipStream_t st = ctx->stream;
		std::memset(rp, 0, sizeof(*rp));
		rp->used[cls] = 1;
		rp->faithful = 1;
		rp->rej_strict = P.rejector_strict != 0;
		rp->resid_from_iter = 2;
		if ((rc = take_epochs(ctx, B, 4u, *rp)) != MULLS_OK)
			return rc;
		uint32_t lds_cap = 0;
		int tier = 0;
		// the stage entry points hand out the raw nearest neighbours (nn_idx before the duplicate rule): k_filter applies the chain, not the search kernels
		const double dedup_opt = ctx->opt[MULLS_OPT_LDS_DEDUP];
		ctx->opt[MULLS_OPT_LDS_DEDUP] = 0.0;
		rc = prepare_run(ctx, B, &P, *rp, &lds_cap, &tier);
		ctx->opt[MULLS_OPT_LDS_DEDUP] = dedup_opt;
		if (rc != MULLS_OK)
			return rc;
		launch_clone_src(st, (uint32_t)B->setup_jobs_h.size(), B->setup_jobs, B->descs, B->setup, B->stage, B->tmp_pos, B->tmp_nrm, B->bbox, *rp);
		launch_crop(st, 1, B->descs, B->setup, B->bbox, B->stage, B->tmp_pos, B->tmp_nrm, B->spos, B->snrm, B->tpos, B->tnrm, B->flag, B->match,
					B->wd, *rp, B->grids, (uint32_t)B->big_segs_h.size(), B->big_segs, (uint32_t)B->big_clouds_h.size(), B->big_clouds, B->seg_cnt,
					B->big_box);
		if (tier == 2)
			launch_grid_build_sort(st, 1, B->descs, B->grids, *rp, B->tpos, B->cell_start, B->tsorted);
		launch_bm_build(st, (uint32_t)B->lclouds_h.size(), B->lclouds, (uint32_t)B->tjobs_h.size(), B->tjobs, B->descs, B->grids, B->tpos, B->bm, B->pf, B->cell_cnt, B->bm_cs,
						B->tsorted, B->bm_rank);
		return MULLS_OK;
	}
	void identity_state(PairState *s, int iter)
	{
		std::memset(s, 0, sizeof(*s));
		s->T[0] = s->T[5] = s->T[10] = 1.0;
		s->iter = iter;
		s->active = 1;
	}
	} // namespace

	int mulls_stage_correspond(mulls_ctx *ctx, const mulls_cloud *src, const mulls_cloud *tgt, float dis_thre, int normal_check,
							   float angle_thre_degree, int32_t *match, float *d2, uint8_t *flags)
	try
	{
		if (!ctx || !src || !tgt || !match || !d2 || !flags)
			return MULLS_E_INVALID;
		HIPCHK(ctx, hipSetDevice(ctx->device));
		if (src->n == 0)
			return MULLS_OK;
		const int cls = normal_check ? MULLS_GROUND : MULLS_VERTEX;
		mulls_batch *B = nullptr;
		RunParams rp;
		int rc = stage_batch(ctx, cls, src, tgt, &B, &rp, normal_check ? "100000" : "000001");
		if (rc == MULLS_OK)
		{
			rp.cos_bearing = std::cos(angle_thre_degree / 180.0 * M_PI);
			identity_state(&B->states_h[0], 0);
			for (int c = 0; c < MULLS_NC; c++)
				B->states_h[0].thr[c] = dis_thre;
			hipStream_t st = ctx->stream;
			hipError_t e = hipSuccess;
			launch_push_states(st, B->states_pin, B->states, 1);
			uint32_t lds_cap = 0;
			const int tier = choose_tier(ctx, B, rp.used, &lds_cap);
			if (tier == 2)
				launch_nn_lds(st, (uint32_t)B->cjobs_h.size(), B->cjobs, B->descs, B->states, rp, B->spos, B->snrm, B->grids, B->cell_start, B->tsorted, B->flag,
							  B->nn_idx, B->nn_d2, B->winner, B->tnrm, B->match, B->wd, B->tpos, B->nn_hint, B->mq, lds_cap, rp.grid_maxcells, B->wl, B->wl_ctr, 0u);
			else if (tier == 1)
				launch_cert_big(st, (uint32_t)B->bjobs_h.size(), B->bjobs, 2048u, B->descs, B->states, rp, B->spos, B->snrm, B->grids, B->bm, B->pf, B->bm_cs, B->tsorted, B->flag,
								B->nn_idx, B->nn_d2, B->winner, B->tpos, B->tnrm, B->nn_hint, B->match, B->wd, B->mq);
			else if (tier < 0)
				rc = MULLS_E_INVALID;
			else
				launch_nn(st, B->njobs, B->jobs, B->descs, B->states, rp, B->spos, B->snrm, B->tpos, B->flag, B->nn_idx, B->nn_d2, B->winner);
			if (!rp.lds_dedup)
				launch_filter(st, B->njobs, B->jobs, B->descs, B->states, rp, B->snrm, B->tnrm, B->flag, B->nn_idx, B->nn_d2, B->match, B->wd,
						  B->winner, B->tpos, B->mq);
			const uint32_t off = B->descs_h[cls].src_off;
			if (e == hipSuccess)
				e = hipMemcpyAsync(match, B->nn_idx + off, sizeof(int32_t) * src->n, hipMemcpyDeviceToHost, st);
			if (e == hipSuccess)
				e = hipMemcpyAsync(d2, B->nn_d2 + off, sizeof(float) * src->n, hipMemcpyDeviceToHost, st);
			if (e == hipSuccess)
				e = hipMemcpyAsync(flags, B->flag + off, src->n, hipMemcpyDeviceToHost, st);
			if (e == hipSuccess)
				e = hipStreamSynchronize(st);
			if (e != hipSuccess)
			{
				ctx->err = hipGetErrorString(e);
				rc = MULLS_E_HIP;
			}
			if (rc == MULLS_OK && (src->n < 3 || tgt->n < 3))
				for (uint32_t i = 0; i < src->n; i++) // search skipped (K_min): nothing was written by the kernels
				{
					match[i] = -1;
					d2[i] = 0.0f;
				}
		}
		mulls_batch_destroy(ctx, B);
		return rc;
	}
	catch (...)
	{
		return mulls::abi_caught(const_cast<mulls_ctx *>(ctx)); // nothing is thrown across the ABI
	}

	int mulls_stage_accumulate(mulls_ctx *ctx, int metric, const mulls_cloud *src, const mulls_cloud *tgt, const int32_t *corr_src,
							   const int32_t *corr_tgt, const float *corr_d2, uint32_t ncorr, int iter_num, float class_weight, int dist_w,
							   int resid_w, int inten_w, float window, double *out27, float *weight_out)
	try
	{
		if (!ctx || !src || !tgt || !out27 || metric < 0 || metric > 2 || (ncorr && (!corr_src || !corr_tgt)))
			return MULLS_E_INVALID;
		HIPCHK(ctx, hipSetDevice(ctx->device));
		for (int k = 0; k < 27; k++)
			out27[k] = 0.0;
		if (src->n == 0 || ncorr == 0)
			return MULLS_OK;
		for (uint32_t i = 0; i < ncorr; i++)
			if (corr_src[i] < 0 || (uint32_t)corr_src[i] >= src->n || corr_tgt[i] < 0 || (uint32_t)corr_tgt[i] >= tgt->n)
				return MULLS_E_INVALID;
		const int cls = metric == 0 ? MULLS_FACADE : (metric == 1 ? MULLS_PILLAR : MULLS_VERTEX);
		const char *used = metric == 0 ? "001000" : (metric == 1 ? "010000" : "000001");
		mulls_batch *B = nullptr;
		RunParams rp;
		int rc = stage_batch(ctx, cls, src, tgt, &B, &rp, used);
		int32_t *dcs = nullptr, *dct = nullptr;
		float *dcd = nullptr;
		if (rc == MULLS_OK)
		{
			rp.w_dist = dist_w != 0;
			rp.w_resid = resid_w != 0; // k_accum additionally requires iter_num > 2, like the reference
			rp.w_inten = inten_w != 0;
			rp.win_pl = rp.win_li = rp.win_pt = window;
			rp.force_class_w = 1;
			rp.class_w_value = class_weight;
			hipStream_t st = ctx->stream;
			hipError_t e = hipSuccess;
			if (dmalloc(ctx, &dcs, ncorr) != MULLS_OK || dmalloc(ctx, &dct, ncorr) != MULLS_OK || dmalloc(ctx, &dcd, ncorr) != MULLS_OK)
				e = hipErrorOutOfMemory;
			if (e == hipSuccess)
				e = hipMemcpyAsync(dcs, corr_src, sizeof(int32_t) * ncorr, hipMemcpyHostToDevice, st);
			if (e == hipSuccess)
				e = hipMemcpyAsync(dct, corr_tgt, sizeof(int32_t) * ncorr, hipMemcpyHostToDevice, st);
			if (e == hipSuccess && corr_d2)
				e = hipMemcpyAsync(dcd, corr_d2, sizeof(float) * ncorr, hipMemcpyHostToDevice, st);
			identity_state(&B->states_h[0], iter_num);
			launch_push_states(ctx->stream, B->states_pin, B->states, 1);
			const uint32_t off = B->descs_h[cls].src_off;
			if (e == hipSuccess)
			{
				// clear every flag to "alive, not a correspondence", then switch the requested ones on
				e = hipMemsetAsync(B->flag + off, MULLS_F_ALIVE, src->n, st);
				launch_set_corr(st, off, dcs, dct, corr_d2 ? dcd : nullptr, ncorr, B->flag, B->match, B->wd, B->descs_h[cls].tgt_off, B->tpos, B->tnrm, B->mq);
				for (int k = 0; k < B->nsub; k++)
				launch_accum(st, B->ajobs, B->ajob_split[k], B->jobs, B->descs, B->states, rp, B->spos, B->mq, B->flag, B->wd, B->partial);
				launch_finish(st, 1, B->descs, B->states, rp, B->partial, B->outs, B->outs_pin, B->bbox, B->ticket, B->epoch_dev, ++B->epoch, 0);
			}
			std::vector<float> wall(src->n);
			if (e == hipSuccess)
				e = hipMemcpyAsync(wall.data(), B->wd + off, sizeof(float) * src->n, hipMemcpyDeviceToHost, st);
			if (e == hipSuccess)
				e = hipStreamSynchronize(st);
			if (e != hipSuccess)
			{
				ctx->err = hipGetErrorString(e);
				rc = MULLS_E_HIP;
			}
			else
			{
				PairOut o;
				unpack_out(B, rp.used, 0, o);
				std::memcpy(out27, o.sums[cls], sizeof(double) * 27);
				if (weight_out)
					for (uint32_t i = 0; i < ncorr; i++)
						weight_out[i] = wall[corr_src[i]];
			}
		}
		if (dcs)
			(void)hipFree(dcs);
		if (dct)
			(void)hipFree(dct);
		if (dcd)
			(void)hipFree(dcd);
		mulls_batch_destroy(ctx, B);
		return rc;
	}
	catch (...)
	{
		return mulls::abi_caught(const_cast<mulls_ctx *>(ctx)); // nothing is thrown across the ABI
	}
}
