// batch.cpp — the device-resident batch of the host driver: options, job tables, staging of the caller's clouds (batch_fill), the per-run device
// tables and the tier choice (prepare_run).  Reference: the set-up part of CRegistration::mm_lls_icp (cregistration.hpp:1136-1232).
#include "batch.h"

namespace mulls_drv
{
void rows12(const double colmajor[16], double out[12])
{
	for (int r = 0; r < 3; r++)
		for (int c = 0; c < 4; c++)
			out[r * 4 + c] = colmajor[r + 4 * c];
}


// the intersection box the device used (cregistration.hpp:2912-2916, utility.hpp:857-865), re-derived for the caller
void fill_crop_box(const RunParams &rp, const double tgt_bound[6], const uint32_t keys[6], mulls_result &R)
{
	R.cropped = rp.crop ? 1 : 0;
	for (int k = 0; k < 3 && rp.crop; k++)
	{
		const uint32_t kmin = keys[k], kmax = keys[3 + k];
		const bool none = kmin == 0xffffffffu && kmax == 0u;
		const double mmin = none ? 1.7976931348623157e308 : (double)ord_to_float(kmin);
		const double mmax = none ? -1.7976931348623157e308 : (double)ord_to_float(kmax);
		const double b1min = tgt_bound[k], b1max = tgt_bound[3 + k];
		const float pad = 1.0f;
		R.crop_box[k] = ((b1min > mmin) ? b1min : mmin) - pad;
		R.crop_box[3 + k] = ((b1max < mmax) ? b1max : mmax) + pad;
	}
}

int check_params(mulls_ctx *ctx, const mulls_params *P)
{
	if (!P)
		return MULLS_E_INVALID;
	if (std::strlen(P->used_feature_type) < 6 || std::strlen(P->weight_strategy) < 4)
	{
		ctx->err = "used_feature_type needs 6 characters and weight_strategy 4";
		return MULLS_E_INVALID;
	}
	return MULLS_OK;
}

// certified correspondences of the LDS tier (k_cert / k_nn_lds): MULLS_OPT_CERTIFICATES, and how much farther than the hinted target a searched
// query sweeps (MULLS_OPT_CERT_SLACK_*: metres, metres, factor on the distance a point moved)
void init_cert(const mulls_ctx *ctx, RunParams &rp)
{
	rp.cert = ctx->opt[MULLS_OPT_CERTIFICATES] != 0.0 ? 1u : 0u;
	rp.cert_slack_min = (float)ctx->opt[MULLS_OPT_CERT_SLACK_MIN];
	rp.cert_slack_max = (float)ctx->opt[MULLS_OPT_CERT_SLACK_MAX];
	rp.cert_slack_rate = (float)ctx->opt[MULLS_OPT_CERT_SLACK_RATE];
	rp.kcert = rp.cert && ctx->opt[MULLS_OPT_KCERT] != 0.0 ? 1u : 0u;
	rp.kcert_min = (uint32_t)ctx->opt[MULLS_OPT_KCERT_MIN];
}

// sub-batches in flight of a host-stepped lock-step batch of n pairs
int subbatch_count(const mulls_ctx *ctx, int n)
{
	int nsub = n >= 2048 ? 2 : 1; // below that the half-size launches cost more (k_nn_lds tail) than the overlap returns
	if (ctx->opt[MULLS_OPT_SUBBATCHES] >= 1.0)
		nsub = std::max(1, std::min(2, (int)ctx->opt[MULLS_OPT_SUBBATCHES]));
	return n < 2 ? 1 : nsub;
}

// Is `*value` acceptable for `option`?  Options are cast to counts and sizes where they are used (no option changes a result, but a negative or infinite
// value would be an undefined cast, a huge one an absurd allocation): finite, inside the option's range; the array stagger is rounded down to a multiple of
// 256 bytes (the per-point arrays hold float4 records).
bool option_value_ok(int option, double *value)
{
	const double v = *value;
	if (!(v == v) || v > 1.0e15 || v < -1.0e15)
		return false;
	switch (option)
	{
	case MULLS_OPT_STAGGER:
		if (v < 0.0 || v > 65536.0)
			return false;
		*value = (double)((uint64_t)v & ~(uint64_t)255);
		return true;
	case MULLS_OPT_GRID_H0:
	case MULLS_OPT_BM_H0:
		return v >= 0.0 && v <= 1.0e4;
	case MULLS_OPT_CERT_SLACK_MIN:
	case MULLS_OPT_CERT_SLACK_MAX:
		return v >= 0.0 && v <= 1.0e3;
	case MULLS_OPT_CERT_SLACK_RATE:
		return v >= 0.0 && v <= 1.0e3;
	case MULLS_OPT_DEBUG_TICK:
		return v >= 0.0 && v <= 4294967295.0;
	default:
		return v >= 0.0 && v <= 2147483647.0; // switches, pair counts, iteration counts
	}
}

} // namespace mulls_drv
int SegCopier::flush(hipStream_t st)
{
	using namespace mulls;
	size_t total = 0;
	for (const Pending &h : host)
		total += (h.bytes + 255u) & ~(size_t)255;
	bool direct = false; // ranges k_copy_segs does not take (misaligned, not whole words, 4 GiB): plain copy commands
	for (const Pending &h : host)
		direct = direct || (h.bytes & 3u) || ((uintptr_t)h.dst & 15u) || h.bytes > 0xfffffff0ull;
	for (const CopySeg &g : segs)
		direct = direct || (g.bytes & 3u) || (g.dst & 15ull) || (g.src & 15ull);
	if (direct)
	{
		for (const CopySeg &g : segs)
			HIPCHK(ctx, hipMemcpyAsync((void *)(uintptr_t)g.dst, (const void *)(uintptr_t)g.src, g.bytes, hipMemcpyDeviceToDevice, st));
		for (const Pending &h : host)
			HIPCHK(ctx, hipMemcpyAsync(h.dst, h.src, h.bytes, hipMemcpyHostToDevice, st));
		segs.clear(), host.clear();
		return MULLS_OK;
	}
	if (total > ctx->mail_cap)
	{
		HIPCHK(ctx, hipStreamSynchronize(st)); // (nothing may still be reading the old mailbox)
		if (ctx->mail_h)
			(void)hipHostFree(ctx->mail_h);
		ctx->mail_h = nullptr, ctx->mail_cap = 0;
		const size_t want = std::max<size_t>(total + total / 2, (size_t)1 << 20);
		HIPCHK(ctx, hipHostMalloc((void **)&ctx->mail_h, want, hipHostMallocMapped));
		ctx->mail_cap = want;
	}
	unsigned char *mail_d = nullptr;
	if (total)
		HIPCHK(ctx, hipHostGetDevicePointer((void **)&mail_d, ctx->mail_h, 0));
	size_t off = 0;
	for (const Pending &h : host)
	{
		std::memcpy(ctx->mail_h + off, h.src, h.bytes);
		segs.push_back({(unsigned long long)(uintptr_t)h.dst, (unsigned long long)(uintptr_t)(mail_d + off), (uint32_t)h.bytes, 0u});
		off += (h.bytes + 255u) & ~(size_t)255;
	}
	if (!segs.empty())
	{
		launch_copy_segs(st, segs.data(), (uint32_t)segs.size());
		HIPCHK(ctx, hipGetLastError());
	}
	segs.clear(), host.clear();
	return MULLS_OK;
}
namespace mulls_drv
{
// defaults of enum mulls_option, then the presets from the environment (read here and nowhere else)
void options_init(mulls_ctx *ctx)
{
	double *o = ctx->opt;
	o[MULLS_OPT_HOST_STEP] = 0, o[MULLS_OPT_RESIDENT_MIN_PAIRS] = 1, o[MULLS_OPT_RESIDENT_MAX_PAIRS] = 0, o[MULLS_OPT_FEW_LAUNCHES_MAX_PAIRS] = 640;
	o[MULLS_OPT_SUBBATCHES] = 0, o[MULLS_OPT_TWO_STREAMS] = 0, o[MULLS_OPT_CERTIFICATES] = 1;
	o[MULLS_OPT_CERT_SLACK_MIN] = 0.02, o[MULLS_OPT_CERT_SLACK_MAX] = 0.10, o[MULLS_OPT_CERT_SLACK_RATE] = 1.0;
	o[MULLS_OPT_SPLIT_MIN_PAIRS] = 96, o[MULLS_OPT_SPLIT_MAX_PAIRS] = 1 << 30, o[MULLS_OPT_FUSED_TGT_SETUP] = 1, o[MULLS_OPT_STAGGER] = 4352, o[MULLS_OPT_STEP_LAUNCH_MAX_PAIRS] = 640, o[MULLS_OPT_LDS_DEDUP] = 1, o[MULLS_OPT_GRID_H0] = 0, o[MULLS_OPT_BM_H0] = 0, o[MULLS_OPT_LEAN_STAGING] = 0, o[MULLS_OPT_DEBUG_STOP] = 0, o[MULLS_OPT_DEBUG_TICK] = 0, o[MULLS_OPT_MIXED_TIERS] = 1, o[MULLS_OPT_BIG_EARLY_SETS] = 5, o[MULLS_OPT_KCERT] = 1, o[MULLS_OPT_KCERT_MIN] = 64, o[MULLS_OPT_ACCUM_WAVE_MIN_TRIPS] = 768;
	o[MULLS_OPT_FIRST_DIRECT] = 1;
	o[MULLS_OPT_SUM_STEP] = 1;
	static const struct
	{
		const char *name;
		int opt;
	} env[] = {{"MULLS_HOST_STEP", MULLS_OPT_HOST_STEP}, {"MULLS_RESIDENT_MIN_PAIRS", MULLS_OPT_RESIDENT_MIN_PAIRS}, {"MULLS_RESIDENT_MAX_PAIRS", MULLS_OPT_RESIDENT_MAX_PAIRS},
			   {"MULLS_FEW_LAUNCHES_MAX_PAIRS", MULLS_OPT_FEW_LAUNCHES_MAX_PAIRS}, {"MULLS_SUBBATCHES", MULLS_OPT_SUBBATCHES}, {"MULLS_TWO_STREAMS", MULLS_OPT_TWO_STREAMS},
			   {"MULLS_CERTIFICATES", MULLS_OPT_CERTIFICATES}, {"MULLS_LDS_DEDUP", MULLS_OPT_LDS_DEDUP}, {"MULLS_GRID_H0", MULLS_OPT_GRID_H0}, {"MULLS_BM_H0", MULLS_OPT_BM_H0},
			   {"MULLS_LEAN_STAGING", MULLS_OPT_LEAN_STAGING}, {"MULLS_FUSED_TGT_SETUP", MULLS_OPT_FUSED_TGT_SETUP}, {"MULLS_STAGGER", MULLS_OPT_STAGGER}, {"MULLS_STEP_LAUNCH_MAX_PAIRS", MULLS_OPT_STEP_LAUNCH_MAX_PAIRS}, {"MULLS_SPLIT_MIN_PAIRS", MULLS_OPT_SPLIT_MIN_PAIRS}, {"MULLS_SPLIT_MAX_PAIRS", MULLS_OPT_SPLIT_MAX_PAIRS}, {"MULLS_DEBUG_STOP", MULLS_OPT_DEBUG_STOP}, {"MULLS_DEBUG_TICK", MULLS_OPT_DEBUG_TICK}, {"MULLS_MIXED_TIERS", MULLS_OPT_MIXED_TIERS}, {"MULLS_BIG_EARLY_SETS", MULLS_OPT_BIG_EARLY_SETS}, {"MULLS_KCERT", MULLS_OPT_KCERT}, {"MULLS_KCERT_MIN", MULLS_OPT_KCERT_MIN}, {"MULLS_ACCUM_WAVE_MIN_TRIPS", MULLS_OPT_ACCUM_WAVE_MIN_TRIPS}, {"MULLS_FIRST_DIRECT", MULLS_OPT_FIRST_DIRECT}, {"MULLS_SUM_STEP", MULLS_OPT_SUM_STEP}};
	for (const auto &e : env)
		if (const char *v = std::getenv(e.name))
		{
			double x = std::strtod(v, nullptr);
			if (option_value_ok(e.opt, &x)) // (an unusable preset is ignored: the default stays)
				o[e.opt] = x;
		}
	if (std::getenv("MULLS_NO_CERT"))
		o[MULLS_OPT_CERTIFICATES] = 0;
	if (std::getenv("MULLS_NO_LDS_DEDUP"))
		o[MULLS_OPT_LDS_DEDUP] = 0;
	if (const char *e = std::getenv("MULLS_CERT_SLACK")) // "min,max,rate"
	{
		float a = 0, b = 0, c = 0;
		if (std::sscanf(e, "%f,%f,%f", &a, &b, &c) == 3 && a >= 0.0f && b >= a && c >= 0.0f)
			o[MULLS_OPT_CERT_SLACK_MIN] = a, o[MULLS_OPT_CERT_SLACK_MAX] = b, o[MULLS_OPT_CERT_SLACK_RATE] = c;
	}
}

// The largest target class cloud whose duplicate table (4 B per target) still fits next to the staged cloud in the heavy pass's LDS with a 4096-cell
// table (prepare_run: dedup_fits): the LDS tier's one-pass walk needs it, so auto mode keeps larger clouds off that tier
uint32_t lds_dedup_max_pts()
{
	const long room = 160L * 1024L - 64L - (long)MULLS_ICP_STATIC_LDS - (long)MULLS_LDS_QCHUNK * 16L - (long)MULLS_LDS_AUX - 2L * (4096L + 8L);
	return (uint32_t)((room / 18L) & ~7L);
}

// Search tier and grid slot of every class cloud pair of the batch.  mode 0 / 1 / 2: one tier for the whole batch (nn_mode 1 - 4, the variants, the stage
// calls); mode 3 (auto mode of mulls_batch_run): per class cloud — the LDS tier for the down-sampled clouds it was built for (target within
// lds_dedup_max_pts(), source within MULLS_SMALL_SRC_MAX), the global-memory tier for everything larger, so that one large cloud no longer decides for a batch.
// Classes the run does not search only need their crop (the reference reports their sizes): k_tgt_grid's when the target fits its lanes, k_crop's otherwise.
void assign_tiers(mulls_batch *B, const uint8_t used[MULLS_NC], int mode)
{
	B->lclouds_h.clear();
	B->tier_mode = mode;
	uint32_t n_used = 0;
	for (int c = 0; c < MULLS_NC; c++)
		n_used += used[c] ? 1u : 0u;
	const uint32_t s_max = lds_dedup_max_pts();
	for (int p = 0; p < B->n; p++)
	{
		// a pair whose large clouds hold most of its source points (a dense scan pair: two 100 k-point classes next to a few thousand pillar / beam / vertex
		// points) keeps its small clouds on the global-memory tier too: they ride along in that tier's launch instead of paying for the LDS tier's
		uint64_t src_small = 0, src_big = 0;
		for (int c = 0; c < MULLS_NC && mode == 3; c++)
		{
			const CloudDesc &d = B->descs_h[(size_t)p * MULLS_NC + c];
			if (used[c])
				((d.tgt_n0 <= s_max && d.src_cap <= MULLS_SMALL_SRC_MAX) ? src_small : src_big) += d.src_cap;
		}
		const bool all_big = src_big >= 4u * src_small && src_big > 0;
		uint32_t rank = 0;
		for (int c = 0; c < MULLS_NC; c++)
		{
			CloudDesc &d = B->descs_h[(size_t)p * MULLS_NC + c];
			uint32_t t = (uint32_t)mode;
			if (mode == 3)
			{
				if (used[c])
					t = (d.tgt_n0 <= s_max && d.src_cap <= MULLS_SMALL_SRC_MAX && !all_big) ? MULLS_TIER_LDS : MULLS_TIER_BM;
				else
					t = d.tgt_n0 <= MULLS_LDS_MAXPTS ? MULLS_TIER_LDS : MULLS_TIER_BM;
			}
			d.tier = t;
			d.grid_slot = (uint32_t)p * n_used + rank;
			if (t == MULLS_TIER_BM)
			{
				d.grid_slot = used[c] ? (uint32_t)B->lclouds_h.size() : 0u;
				if (used[c])
					B->lclouds_h.push_back((uint32_t)p * MULLS_NC + (uint32_t)c);
			}
			rank += used[c] ? 1u : 0u;
		}
	}
}

void build_jobs(mulls_batch *B, const mulls_params *P, int nsub, int mode)
{
	std::string key(P->used_feature_type, 6);
	key += (char)('0' + nsub);
	key += (char)('a' + mode);
	if (key == B->jobs_key)
		return;
	uint8_t used[MULLS_NC];
	for (int c = 0; c < MULLS_NC; c++)
		used[c] = P->used_feature_type[c] == '1';
	assign_tiers(B, used, mode);
	B->nsub = nsub;
	B->jobs_h.clear();
	for (int p = 0; p < B->n; p++)
		for (int c = 0; c < MULLS_NC; c++)
		{
			CloudDesc &d = B->descs_h[p * MULLS_NC + c];
			d.job_begin = (uint32_t)B->jobs_h.size();
			if (P->used_feature_type[c] == '1')
				for (uint32_t s = 0; s < d.src_cap; s += MULLS_SRC_PER_BLOCK)
				{
					Job j = {(uint32_t)p, (uint32_t)c, s, 0};
					B->jobs_h.push_back(j);
				}
			d.job_end = (uint32_t)B->jobs_h.size();
		}
	// LDS tier: one class-level job per (pair, used class) cloud on it
	B->cjobs_h.clear();
	for (int p = 0; p < B->n; p++)
		for (int c = 0; c < MULLS_NC; c++)
		{
			const CloudDesc &d = B->descs_h[p * MULLS_NC + c];
			if (used[c] && d.src_cap > 0 && d.tier == MULLS_TIER_LDS)
			{
				Job j = {(uint32_t)p, (uint32_t)c, 0u, d.src_cap};
				B->cjobs_h.push_back(j);
			}
		}
	uint32_t max_src_cap = 0;
	for (const Job &j : B->cjobs_h)
		max_src_cap = std::max(max_src_cap, j.count);
	if (mode == 2 && ((B->cjobs_h.size() < 512 && max_src_cap > 4096u) || max_src_cap > 65534u)) // (k_cert's duplicate table holds 16-bit source indices)
	{
		// whole batch on the LDS tier (nn_mode 3 / 4), few AND large class clouds (a pair of dense scans): split them into 512-query jobs (each stages its
		// target cloud itself) so that more than a handful of workgroups walk them.  (Auto mode sends such clouds to the global-memory tier.)
		B->cjobs_h.clear();
		for (int p = 0; p < B->n; p++)
			for (int c = 0; c < MULLS_NC; c++)
				if (P->used_feature_type[c] == '1')
					for (uint32_t s = 0; s < B->descs_h[p * MULLS_NC + c].src_cap; s += MULLS_SRC_PER_BLOCK)
					{
						Job j = {(uint32_t)p, (uint32_t)c, s, MULLS_SRC_PER_BLOCK};
						B->cjobs_h.push_back(j);
					}
	}
	// global-memory tier: a whole source class cloud per job where one workgroup can hold it (mixed batches: a down-sampled scan against a large map —
	// no k_filter launch for those), 512-point chunks otherwise
	B->bjobs_h.clear();
	B->fjobs_h.clear();
	B->ejobs_h.clear();
	bool any_class_job = false;
	for (int p = 0; p < B->n; p++)
		for (int c = 0; c < MULLS_NC; c++)
		{
			const CloudDesc &d = B->descs_h[p * MULLS_NC + c];
			if (!used[c] || d.src_cap == 0 || d.tier != MULLS_TIER_BM)
				continue;
			const bool whole = mode == 3 && d.src_cap <= MULLS_BIG_CLASS_MAX;
			if (whole)
			{
				Job j = {(uint32_t)p, (uint32_t)c, 0u, d.src_cap | MULLS_JOB_CLASS};
				B->bjobs_h.push_back(j);
				any_class_job = true;
			}
			for (uint32_t s = 0; s < d.src_cap; s += MULLS_SRC_PER_BLOCK)
			{
				Job j = {(uint32_t)p, (uint32_t)c, s, MULLS_SRC_PER_BLOCK};
				B->ejobs_h.push_back(j);
				if (!whole)
				{
					B->bjobs_h.push_back(j);
					B->fjobs_h.push_back(j);
				}
			}
		}
	if (!any_class_job)
		B->ejobs_h.clear(); // (the same list as bjobs_h)
	// device order of the class-level jobs: longest first inside each sub-batch's slice, so that the last round of workgroups
	// of a launch is made of the cheap class clouds (cost ~ queries x log(targets); ties keep the pair order)
	B->cjobs_dev_h = B->cjobs_h;
	{
		auto first_of = [&](uint32_t pair) {
			return std::lower_bound(B->cjobs_dev_h.begin(), B->cjobs_dev_h.end(), pair, [](const Job &j, uint32_t q) { return j.pair < q; });
		};
		auto cost = [&](const Job &j) {
			const CloudDesc &d = B->descs_h[j.pair * MULLS_NC + j.cls];
			return (uint64_t)j.count * (uint64_t)(64u + d.tgt_n0 / 64u);
		};
		for (int k = 0; k < nsub; k++)
			std::stable_sort(first_of((uint32_t)((long)B->n * k / nsub)), first_of((uint32_t)((long)B->n * (k + 1) / nsub)),
							 [&](const Job &a, const Job &b) { return cost(a) > cost(b); });
	}
	// device-resident loop: one class-level job per (pair, used class with source points), in pair order; pairs taken from the
	// queue most expensive first (same cost model), so that the last pairs in flight are the cheap ones
	B->rjobs_h.clear();
	B->pair_rjob_h.assign((size_t)B->n + 1, 0u);
	std::vector<uint64_t> pair_cost(B->n, 0);
	for (int p = 0; p < B->n; p++)
	{
		B->pair_rjob_h[p] = (uint32_t)B->rjobs_h.size();
		for (int c = 0; c < MULLS_NC; c++)
		{
			const CloudDesc &d = B->descs_h[p * MULLS_NC + c];
			if (P->used_feature_type[c] == '1' && d.src_cap > 0)
			{
				Job j = {(uint32_t)p, (uint32_t)c, 0u, d.src_cap};
				B->rjobs_h.push_back(j);
				pair_cost[p] += (uint64_t)d.src_cap * (uint64_t)(64u + d.tgt_n0 / 64u);
			}
		}
	}
	B->pair_rjob_h[B->n] = (uint32_t)B->rjobs_h.size();
	B->order_h.resize(B->n);
	for (int p = 0; p < B->n; p++)
		B->order_h[p] = (uint32_t)p;
	std::stable_sort(B->order_h.begin(), B->order_h.end(), [&](uint32_t a, uint32_t b) { return pair_cost[a] > pair_cost[b]; });
	B->tjobs_h.clear();
	for (int p = 0; p < B->n; p++)
		for (int c = 0; c < MULLS_NC; c++)
			if (P->used_feature_type[c] == '1' && B->descs_h[p * MULLS_NC + c].tier == MULLS_TIER_BM)
				for (uint32_t s = 0; s < B->descs_h[p * MULLS_NC + c].tgt_n0; s += MULLS_BLOCK)
				{
					Job j = {(uint32_t)p, (uint32_t)c, s, 0};
					B->tjobs_h.push_back(j);
				}
	B->ajobs_h.clear();
	{
		for (int k = 0; k < 2; k++)
			for (int b = 0; b < 4; b++)
				B->ajob_split[k][b] = 0;
		for (int k = 0; k < nsub; k++)
		{
			const uint32_t lo = (uint32_t)((long)B->n * k / nsub), hi = (uint32_t)((long)B->n * (k + 1) / nsub);
			for (int bucket = 0; bucket < 3; bucket++)
			{
				B->ajob_split[k][bucket] = (uint32_t)B->ajobs_h.size();
				for (uint32_t j = 0; j < (uint32_t)B->jobs_h.size(); j++)
				{
					const Job &jb = B->jobs_h[j];
					if (jb.start % 1024u != 0u || jb.pair < lo || jb.pair >= hi)
						continue;
					const uint32_t slots = std::min(1024u, B->descs_h[jb.pair * MULLS_NC + jb.cls].src_cap - jb.start);
					if ((slots > 512u ? 0 : (slots > 256u ? 1 : 2)) == bucket)
						B->ajobs_h.push_back(j);
				}
			}
			B->ajob_split[k][3] = (uint32_t)B->ajobs_h.size();
		}
	}
	B->njobs = (uint32_t)B->jobs_h.size();
	B->jobs_key = key;
}

// Wait until k_finish has published the current epoch.  The host spins on the pinned word (a few microseconds of latency
// instead of an interrupt-driven stream synchronisation); a stalled device is caught by falling back to
// hipStreamSynchronize, which also surfaces asynchronous HIP errors.
// `last`: while profiling, the event recorded behind the k_finish that publishes `want` — waiting on it (instead of the
// whole stream) leaves the other sub-batch's kernels running
int wait_epoch_word(mulls_ctx *ctx, volatile uint32_t *word, uint32_t want, hipEvent_t last, hipStream_t stream)
{
	const auto t0 = std::chrono::steady_clock::now();
	bool seen = false;
	for (uint64_t spins = 0;; spins++)
	{
		if (*word == want)
		{
			std::atomic_thread_fence(std::memory_order_acquire);
			seen = true;
			break;
		}
		if ((spins & 0xfff) == 0xfff && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0)
			break;
	}
	if (seen && ctx->profiling != 1)
		return MULLS_OK; // (level 2: the search's events lie before the kernel that published the epoch — complete)
	if (seen && last)
	{
		// profiling: the timing event recorded behind the publishing kernel completes right after it — poll, do not sleep
		hipError_t e;
		while ((e = hipEventQuery(last)) == hipErrorNotReady)
		{
		}
		if (e == hipSuccess)
			return MULLS_OK;
	}
	HIPCHK(ctx, hipStreamSynchronize(stream ? stream : ctx->stream));
	if (*word != want)
	{
		ctx->err = "device did not publish the iteration epoch";
		return MULLS_E_HIP;
	}
	return MULLS_OK;
}

// one pair's packed record (k_pull_outs: 128-B counter block, then the used classes' 224-B rows — or, with `comb`, the single
// combined row k_finish assembled) -> PairOut
void unpack_out(const mulls_batch *B, const uint8_t used[MULLS_NC], int p, PairOut &o, bool comb)
{
	int n_used = 0;
	for (int c = 0; c < MULLS_NC; c++)
		n_used += used[c] ? 1 : 0;
	const size_t row = sizeof(double) * MULLS_NTERM_PAD, rec = 128 + row * (size_t)(comb ? 1 : n_used);
	const unsigned char *src = reinterpret_cast<const unsigned char *>(B->outs_h) + rec * (size_t)p;
	std::memcpy(o.n_valid, src, 128); // n_valid, n_alive, src_n, tgt_n, bbox, pad_: contiguous
	src += 128;
	if (comb)
	{
		std::memcpy(o.comb, src, row);
		return;
	}
	for (int c = 0; c < MULLS_NC; c++)
		if (used[c])
		{
			std::memcpy(o.sums[c], src, row);
			src += row;
		}
		else
			std::memset(o.sums[c], 0, row); // never sent; contributes nothing
}

// the largest class cloud the LDS tier accepts must leave room for the 4096-cell floor lds_cells_for() promises
static_assert(160L * 1024L - 64L - (long)MULLS_LDS_QCHUNK * 16L - (long)MULLS_LDS_AUX - (long)MULLS_LDS_MAXPTS * 14L >= (4096L + 8L) * 2L,
			  "MULLS_LDS_MAXPTS does not fit next to the query block, the cost-sort tables and a 4096-cell table in 160 KiB of LDS");
// cell budget of the LDS tier: whatever the 160 KiB leave free next to the staged points (14 B each) and the query block
uint32_t lds_cells_for(uint32_t cap)
{
	const long free_bytes = 160L * 1024L - (long)MULLS_LDS_QCHUNK * 16L - (long)MULLS_LDS_AUX - (long)cap * 14L - 64L;
	long cells = free_bytes / 2 - 8;
	cells = std::min<long>(cells, (long)MULLS_MAXCELLS);
	return (uint32_t)std::max<long>(cells, 4096);
}

// search tier of a run: 0 = LDS-tiled brute force, 1 = bitmap grid in global memory, 2 = dense grid staged in LDS, 3 = per class cloud (assign_tiers).
// *lds_cap: the largest target class cloud the LDS tier's kernels will stage
int choose_tier(const mulls_ctx *ctx, const mulls_batch *B, const uint8_t used[MULLS_NC], uint32_t *lds_cap, const mulls_params *P_mixed)
{
	uint32_t max_t = 0;
	for (int p = 0; p < B->n; p++)
		for (int c = 0; c < MULLS_NC; c++)
			if (used[c])
				max_t = std::max(max_t, B->descs_h[p * MULLS_NC + c].tgt_n0);
	*lds_cap = std::max(8u, (max_t + 7u) & ~7u);
	const bool fits = max_t <= MULLS_LDS_MAXPTS;
	switch (ctx->nn_mode)
	{
	case 1:
		return 0;
	case 2:
		return 1;
	case 3:
	case 4:
		return fits ? 2 : -1;
	default:
		break;
	}
	// auto.  A caller that can run mixed batches gets the tier per class cloud as soon as one cloud is beyond the LDS tier's one-pass walk — unless the
	// run needs the cropped target copies everywhere (keep-less thinning, the normal-shooting search) or the LDS tier's fused paths are switched off
	const bool mixed_ok = P_mixed && !P_mixed->keep_less_source_points && !P_mixed->normal_shooting_on && ctx->opt[MULLS_OPT_MIXED_TIERS] != 0.0 &&
						  ctx->opt[MULLS_OPT_FUSED_TGT_SETUP] != 0.0 && ctx->opt[MULLS_OPT_LDS_DEDUP] != 0.0;
	if (mixed_ok)
	{
		const uint32_t s_max = lds_dedup_max_pts();
		uint32_t max_small = 0;
		bool any_big = false;
		for (int p = 0; p < B->n; p++)
			for (int c = 0; c < MULLS_NC; c++)
			{
				const CloudDesc &d = B->descs_h[p * MULLS_NC + c];
				if (!used[c] || d.src_cap == 0)
					continue;
				if (d.tgt_n0 <= s_max && d.src_cap <= MULLS_SMALL_SRC_MAX)
					max_small = std::max(max_small, d.tgt_n0);
				else
					any_big = true;
			}
		if (any_big)
		{
			*lds_cap = std::max(8u, (max_small + 7u) & ~7u);
			return 3;
		}
	}
	// the LDS tier whenever the clouds fit, whatever the batch size: with class-level jobs and four launches per iteration one KITTI pair takes
	// 0.84 ms there against 0.94 ms on the global-memory tier (profiles/r03_modes.txt)
	return fits ? 2 : 1;
}

// float4 units a staged cloud of n points takes (device_types.h: MULLS_STAGE_*)
static inline size_t stage_quads(uint32_t n, uint32_t fmt) { return fmt == MULLS_STAGE_AOS48 ? (size_t)n * 3 : (fmt == MULLS_STAGE_PACK32 ? (size_t)n * 2 : (size_t)n + ((size_t)n * 3 + 3) / 4); }

// lay the pairs out in the batch arenas, (re)allocate what is too small and stage the caller's clouds in HBM.  Host clouds are gathered out of the
// caller's records into the packed layouts (32 of the 48 bytes are live; 28 when the run is known not to undistort: P given); class clouds of a
// device-resident local map keep their 48-byte records and are copied device to device.  P + MULLS_OPT_LEAN_STAGING: the clouds the run never
// reads are staged as empty.
int batch_fill(mulls_ctx *ctx, mulls_batch *B, const mulls_pair *pairs, int n, const mulls_params *P)
{
	const auto t_fill0 = std::chrono::steady_clock::now();
	const uint32_t host_fmt = (P && !P->apply_motion_undistortion) ? MULLS_STAGE_PACK28 : MULLS_STAGE_PACK32;
	const bool lean = P && ctx->opt[MULLS_OPT_LEAN_STAGING] != 0.0;
	const bool crop_on = P && P->apply_intersection_filter != 0 && !P->apply_motion_undistortion;
	// is class c's cloud read by the run?  (source ground / pillar / facade feed the intersection box whatever the used classes are, :2912-2915)
	auto wanted = [&](int c, bool source) { return !lean || P->used_feature_type[c] == '1' || (source && crop_on && c <= 2); };
	HIPCHK(ctx, hipSetDevice(ctx->device));
	B->n = n;
	B->descs_h.assign((size_t)n * MULLS_NC, CloudDesc());
	B->setup_h.assign(n, PairSetup());
	B->setup_jobs_h.clear();
	B->big_segs_h.clear();
	B->big_clouds_h.clear();
	B->n_big_tgt = 0;
	B->jobs_key.clear(); // the job table depends on the layout
	B->dev_key.clear();
	size_t stage_rec = 0, so = 0, to = 0; // stage_rec: float4 units
	static const mulls_cloud no_cloud = {nullptr, 0u, MULLS_POINT_BYTES};
	auto fmt_of = [&](const mulls_cloud &c) { return (c.n && mulls_is_map_memory(ctx, c.pts, (size_t)c.n * MULLS_POINT_BYTES)) ? MULLS_STAGE_AOS48 : host_fmt; };
	for (int p = 0; p < n; p++)
	{
		for (int c = 0; c < MULLS_NC; c++)
		{
			CloudDesc &d = B->descs_h[p * MULLS_NC + c];
			std::memset(&d, 0, sizeof(d));
			const mulls_cloud &s = wanted(c, true) ? pairs[p].src[c] : no_cloud, &t = wanted(c, false) ? pairs[p].tgt[c] : no_cloud;
			if ((s.n && (!s.pts || s.stride < MULLS_POINT_BYTES)) || (t.n && (!t.pts || t.stride < MULLS_POINT_BYTES)))
			{
				ctx->err = "cloud with points but null pointer or stride < 48";
				return MULLS_E_INVALID;
			}
			const uint32_t sf = fmt_of(s), tf = fmt_of(t);
			d.src_stage = (uint32_t)stage_rec;
			d.src_n0 = s.n;
			stage_rec += stage_quads(s.n, sf);
			d.tgt_stage = (uint32_t)stage_rec;
			d.tgt_n0 = t.n;
			stage_rec += stage_quads(t.n, tf);
			// block2->pc_*_down for the undistortion branch: staged separately only when it is a different cloud (and the run can undistort)
			const mulls_cloud &sd = (wanted(c, true) && !(P && !P->apply_motion_undistortion)) ? pairs[p].src_down[c] : no_cloud;
			const bool own_down = c != MULLS_VERTEX && sd.pts && sd.n && !(sd.pts == s.pts && sd.n == s.n && sd.stride == s.stride);
			if (own_down && sd.stride < MULLS_POINT_BYTES)
			{
				ctx->err = "src_down cloud with stride < 48";
				return MULLS_E_INVALID;
			}
			const uint32_t df = own_down ? fmt_of(sd) : sf;
			d.stage_fmt = sf | (tf << 2) | (df << 4);
			d.sd_stage = own_down ? (uint32_t)stage_rec : d.src_stage;
			d.sd_n0 = own_down ? sd.n : s.n;
			if (own_down)
				stage_rec += stage_quads(sd.n, df);
			d.src_cap = std::max(d.src_n0, d.sd_n0);
			d.src_off = (uint32_t)so;
			d.tgt_off = (uint32_t)to;
			so += d.src_cap;
			to += t.n;
			if (t.n > MULLS_BIG_CLOUD)
			{
				const uint32_t slot = (uint32_t)B->big_clouds_h.size(), first = (uint32_t)B->big_segs_h.size();
				d.big_slot = slot + 1u;
				for (uint32_t k = 0; k < t.n; k += MULLS_SEG)
					B->big_segs_h.push_back({(uint32_t)p, (uint32_t)c, k, slot});
				B->big_clouds_h.push_back({(uint32_t)p, (uint32_t)c, first, (uint32_t)B->big_segs_h.size() - first});
				B->n_big_tgt++;
			}
			if (d.src_cap > MULLS_BIG_CLOUD)
			{
				const uint32_t slot = (uint32_t)B->big_clouds_h.size(), first = (uint32_t)B->big_segs_h.size();
				d.src_big_slot = slot + 1u;
				for (uint32_t k = 0; k < d.src_cap; k += MULLS_SEG)
					B->big_segs_h.push_back({(uint32_t)p, (uint32_t)c | MULLS_BIG_SRC_SIDE, k, slot});
				B->big_clouds_h.push_back({(uint32_t)p, (uint32_t)c | MULLS_BIG_SRC_SIDE, first, (uint32_t)B->big_segs_h.size() - first});
			}
			// clone jobs: 256 source points each; eight times that for a dense scan's clouds (k_clone_src: fewer atomics on the pair's box)
			const uint32_t chunks = d.src_cap > 8192u ? 8u : 1u;
			for (uint32_t k = 0; k < d.src_cap; k += MULLS_BLOCK * chunks)
			{
				Job j = {(uint32_t)p, (uint32_t)c, k, chunks};
				B->setup_jobs_h.push_back(j);
			}
		}
		rows12(pairs[p].init_guess, B->setup_h[p].guess);
		std::memcpy(B->setup_h[p].tgt_bound, pairs[p].tgt_bound, sizeof(double) * 6);
		{
			// inverse(initial_guess) as quaternion + translation (cregistration.hpp:1248, cfilter.hpp:497-500)
			Mat4 g;
			std::memcpy(g.v, pairs[p].init_guess, sizeof(g.v));
			const Mat4 gi = mulls::invert4(g);
			mulls::rotation_quaternion(gi, B->setup_h[p].inv_q);
			B->setup_h[p].inv_t[0] = gi.at(0, 3);
			B->setup_h[p].inv_t[1] = gi.at(1, 3);
			B->setup_h[p].inv_t[2] = gi.at(2, 3);
			// theta of the slerp from the identity (k_clone_src): one value per pair, by the host's libm — the reference's own acos; its sine and the two sines per
			// point are detmath.h's on the device (the same bits on every toolchain)
			B->setup_h[p].inv_t[3] = std::acos(std::min(1.0, std::fabs(B->setup_h[p].inv_q[0])));
		}
	}
	if (stage_rec >= (1ull << 32) || so >= (1ull << 31) || to >= (1ull << 31))
	{
		ctx->err = "batch too large (>= 2^31 points)";
		return MULLS_E_INVALID;
	}
	B->n_src = so;
	B->n_tgt = to;

	int rc = MULLS_OK;
	auto A = [&](int r) { if (rc == MULLS_OK) rc = r; };
	bool winner_grew = false, cand_grew = false;
	const size_t SG = (size_t)ctx->opt[MULLS_OPT_STAGGER]; // bytes between the starts of the per-point arrays inside their 2 MiB pages
	A(grow(ctx, &B->stage, &B->cap_stage, stage_rec));
	A(grow(ctx, &B->tmp_pos, &B->cap_src[0], so, nullptr, 12 * SG));
	A(grow(ctx, &B->tmp_nrm, &B->cap_src[1], so, nullptr, 13 * SG));
	A(grow(ctx, &B->spos, &B->cap_src[2], so, nullptr, 1 * SG));
	A(grow(ctx, &B->snrm, &B->cap_src[3], so, nullptr, 2 * SG));
	A(grow(ctx, &B->flag, &B->cap_src[4], so, nullptr, 3 * SG));
	A(grow(ctx, &B->match, &B->cap_src[5], so, nullptr, 4 * SG));
	A(grow(ctx, &B->nn_idx, &B->cap_src[6], so, nullptr, 5 * SG));
	A(grow(ctx, &B->wd, &B->cap_src[7], so, nullptr, 6 * SG));
	A(grow(ctx, &B->nn_d2, &B->cap_src[8], so, nullptr, 7 * SG));
	A(grow(ctx, &B->nn_hint, &B->cap_src[9], 2 * so, nullptr, 8 * SG)); // LDS tier: (hint word, bound) records
	A(grow(ctx, &B->mq, &B->cap_src[10], 2 * so, nullptr, 9 * SG));
	A(grow(ctx, &B->nn_cand, &B->cap_src[11], so, &cand_grew, 14 * SG)); // k-candidate certificates: 16-byte candidate records
	A(grow(ctx, &B->tpos, &B->cap_tgt[0], to));
	A(grow(ctx, &B->tnrm, &B->cap_tgt[1], to));
	A(grow(ctx, &B->tsorted, &B->cap_tgt[2], to, nullptr, 10 * SG));
	A(grow(ctx, &B->tmap, &B->cap_tgt[4], to, nullptr, 11 * SG));
	A(grow(ctx, &B->winner, &B->cap_tgt[3], to, &winner_grew));
	A(grow(ctx, &B->descs, &B->cap_pairs[0], (size_t)n * MULLS_NC));
	A(grow(ctx, &B->setup, &B->cap_pairs[1], (size_t)n));
	A(grow(ctx, &B->states, &B->cap_pairs[2], (size_t)n));
	A(grow(ctx, &B->outs, &B->cap_outs, (size_t)n));
	A(grow(ctx, &B->bbox, &B->cap_pairs[3], (size_t)n * 6));
	A(grow(ctx, &B->grids, &B->cap_pairs[4], (size_t)n * MULLS_NC));
	A(grow(ctx, &B->setup_jobs, &B->cap_setup_jobs, B->setup_jobs_h.size()));
	A(grow(ctx, &B->big_segs, &B->cap_big[0], B->big_segs_h.size()));
	A(grow(ctx, &B->big_clouds, &B->cap_big[1], B->big_clouds_h.size()));
	A(grow(ctx, &B->seg_cnt, &B->cap_big[2], B->big_segs_h.size()));
	A(grow(ctx, &B->big_box, &B->cap_big[3], (B->big_clouds_h.size() + B->big_segs_h.size()) * 6)); // per cloud (k_crop's reset), then per segment (k_crop_big_count)
	if (!B->ticket)
	{
		A(dmalloc(ctx, &B->ticket, 32));
		if (rc == MULLS_OK && hipMemset(B->ticket, 0, 32 * sizeof(uint32_t)) != hipSuccess)
			rc = MULLS_E_HIP;
	}
	A(grow_pinned(ctx, &B->states_h, &B->cap_pin[0], (size_t)n, hipHostMallocMapped));
	A(grow_pinned(ctx, &B->outs_h, &B->cap_pin[1], (size_t)n, hipHostMallocMapped));
	A(grow_pinned(ctx, &B->bbox_h, &B->cap_pin[2], (size_t)n * 6, hipHostMallocDefault));
	A(grow_pinned(ctx, &B->upload_h, &B->cap_pin[3], std::max<size_t>(stage_rec, 1) * 16, hipHostMallocDefault));
	if (rc == MULLS_OK && !B->epoch_h)
	{
		if (hipHostMalloc((void **)&B->epoch_h, 256, hipHostMallocMapped) != hipSuccess)
			rc = MULLS_E_HIP;
		else
			std::memset((void *)B->epoch_h, 0, 256);
	}
	if (rc != MULLS_OK)
		return rc;
	if (hipHostGetDevicePointer((void **)&B->states_pin, B->states_h, 0) != hipSuccess ||
		hipHostGetDevicePointer((void **)&B->outs_pin, B->outs_h, 0) != hipSuccess ||
		hipHostGetDevicePointer((void **)&B->epoch_dev, (void *)B->epoch_h, 0) != hipSuccess)
	{
		ctx->err = "pinned host memory setup failed";
		return MULLS_E_HIP;
	}
	std::memset(B->states_h, 0, sizeof(PairState) * n);
	for (int p = 0; p < n; p++)
		for (int k = 0; k < 6; k++)
			B->bbox_h[p * 6 + k] = k < 3 ? 0xffffffffu : 0u;

	// gather the live fields of the caller's 48-byte records into pinned memory (packed layouts), blocks of pairs at a time: the host threads pack
	// block k while the copy engine moves block k - 1
	struct DevCopy
	{
		size_t dst;
		const void *src;
		size_t bytes;
	};
	std::vector<DevCopy> dev_copies;
	struct HostCopy
	{
		uint8_t *dst;
		const uint8_t *src;
		uint32_t n, stride, fmt;
	};
	std::vector<HostCopy> host_copies;
	std::vector<size_t> first_copy_of_pair(n + 1, 0);
	for (int p = 0; p < n; p++)
	{
		first_copy_of_pair[p] = host_copies.size();
		for (int c = 0; c < MULLS_NC; c++)
		{
			const CloudDesc &d = B->descs_h[p * MULLS_NC + c];
			const mulls_cloud *cl[3] = {&pairs[p].src[c], &pairs[p].tgt[c], &pairs[p].src_down[c]};
			const uint32_t off[3] = {d.src_stage, d.tgt_stage, d.sd_stage}, cnt[3] = {d.src_n0, d.tgt_n0, d.sd_n0};
			const uint32_t fmt[3] = {d.stage_fmt & 3u, (d.stage_fmt >> 2) & 3u, (d.stage_fmt >> 4) & 3u};
			for (int k = 0; k < (d.sd_stage != d.src_stage ? 3 : 2); k++)
			{
				if (!cnt[k]) // empty, or left out by the lean staging
					continue;
				const uint8_t *src = (const uint8_t *)cl[k]->pts;
				if (fmt[k] == MULLS_STAGE_AOS48)
				{
					// a class cloud of a device-resident local map (mulls_map_cloud): staged by a device-to-device copy below
					if (cl[k]->stride != MULLS_POINT_BYTES)
					{
						ctx->err = "device-resident cloud with stride != 48";
						return MULLS_E_INVALID;
					}
					dev_copies.push_back({(size_t)off[k] * 16, src, (size_t)cnt[k] * MULLS_POINT_BYTES});
					continue;
				}
				host_copies.push_back({B->upload_h + (size_t)off[k] * 16, src, cnt[k], cl[k]->stride, fmt[k]});
			}
		}
	}
	first_copy_of_pair[n] = host_copies.size();
	hipStream_t st = ctx->stream;
	hipError_t e = hipSuccess;
	const int block = 64;
	double pack_s = 0.0;
	for (int p0 = 0; p0 < n && e == hipSuccess; p0 += block)
	{
		const int p1 = std::min(n, p0 + block);
		const long c0 = (long)first_copy_of_pair[p0], c1 = (long)first_copy_of_pair[p1];
		const auto t_pack0 = std::chrono::steady_clock::now();
		// (the process's sleeping thread pool, not an OpenMP team: see HostPool, ctx.h)
		const std::function<void(long)> pack_one = [&](long i) {
			const HostCopy &hc = host_copies[i];
			float *pos = reinterpret_cast<float *>(hc.dst), *nrm = pos + (size_t)hc.n * 4;
			const int nw = hc.fmt == MULLS_STAGE_PACK32 ? 4 : 3;
			for (uint32_t k = 0; k < hc.n; k++)
			{
				float r[10]; // x y z _ nx ny nz _ intensity curvature
				std::memcpy(r, hc.src + (size_t)k * hc.stride, sizeof(r));
				float *pp = pos + (size_t)k * 4, *nn = nrm + (size_t)k * nw;
				pp[0] = r[0], pp[1] = r[1], pp[2] = r[2], pp[3] = r[8];
				nn[0] = r[4], nn[1] = r[5], nn[2] = r[6];
				if (nw == 4)
					nn[3] = r[9];
			}
		};
		if (c1 - c0 >= 12)
			shared_host_pool().parallel_for(c0, c1, 2, pack_one);
		else
			for (long i = c0; i < c1; i++)
				pack_one(i);
		pack_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_pack0).count();
		// the staged clouds of pairs [p0, p1) are one contiguous range (offsets grow with the pair index)
		const size_t r0 = B->descs_h[(size_t)p0 * MULLS_NC].src_stage;
		const size_t r1 = p1 < n ? B->descs_h[(size_t)p1 * MULLS_NC].src_stage : stage_rec;
		if (r1 > r0)
			e = hipMemcpyAsync(reinterpret_cast<uint8_t *>(B->stage) + r0 * 16, B->upload_h + r0 * 16, (r1 - r0) * 16, hipMemcpyHostToDevice, st);
	}
	if (e == hipSuccess)
	{
		// the device-resident clouds and the fill's tables: one launch (SegCopier)
		SegCopier up(ctx);
		for (const DevCopy &dc : dev_copies)
			up.add_dev(reinterpret_cast<uint8_t *>(B->stage) + dc.dst, dc.src, dc.bytes);
		up.add_host(B->setup_jobs, B->setup_jobs_h.data(), B->setup_jobs_h.size() * sizeof(Job));
		up.add_host(B->big_segs, B->big_segs_h.data(), B->big_segs_h.size() * sizeof(Job));
		up.add_host(B->big_clouds, B->big_clouds_h.data(), B->big_clouds_h.size() * sizeof(Job));
		up.add_host(B->setup, B->setup_h.data(), sizeof(PairSetup) * n);
		if (up.flush(st) != MULLS_OK)
		{
			(void)hipStreamSynchronize(st); // (the uploads queued above still read upload_h / setup_h: nothing may rewrite them before they have run)
			return MULLS_E_HIP;
		}
	}
	if (e == hipSuccess && cand_grew) // epoch 0 never passes the epoch test of a run (take_epochs starts at 1)
		e = hipMemsetAsync(B->nn_cand, 0, B->cap_src[11] * sizeof(uint4), st);
	if (e == hipSuccess && winner_grew) // later epochs always sort below older entries (k_nn), so only fresh memory needs the fill
		e = hipMemsetAsync(B->winner, 0xff, B->cap_tgt[3] * sizeof(unsigned long long), st);
	if (e == hipSuccess)
		e = hipStreamSynchronize(st); // setup_jobs_h / setup_h / upload_h may be rewritten by the next fill
	if (e != hipSuccess)
	{
		ctx->err = std::string("staging upload: ") + hipGetErrorString(e);
		return MULLS_E_HIP;
	}
	B->fill_ms = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_fill0).count() * 1e3;
	B->fill_pack_ms = pack_s * 1e3;
	B->fill_bytes = (uint64_t)stage_rec * 16;
	return MULLS_OK;
}

// Per-run device tables.  Job tables and the pristine descriptor block only change with the batch layout or the set of
// used classes, so they are uploaded once (pinned copies would not help: they are simply not re-sent) and every run
// restores the mutable descriptors / box keys with device-to-device copies — no pageable H2D traffic per run.
// nsub: sub-batches the lock-step job tables are laid out for (0 = subbatch_count)
int prepare_run(mulls_ctx *ctx, mulls_batch *B, const mulls_params *P_jobs, RunParams &rp, uint32_t *lds_cap_out, int *tier_out, bool *resident_out, int nsub, bool allow_mixed)
{
	hipStream_t st = ctx->stream;
	const int n = B->n;
	uint32_t lds_cap = 0;
	const int tier = choose_tier(ctx, B, rp.used, &lds_cap, allow_mixed ? P_jobs : nullptr);
	if (tier < 0)
	{
		ctx->err = "nn mode 3 (grid staged in LDS) needs every searched target class cloud to hold <= 9728 points";
		return MULLS_E_INVALID;
	}
	build_jobs(B, P_jobs, nsub > 0 ? nsub : subbatch_count(ctx, n), tier);
	*lds_cap_out = lds_cap;
	*tier_out = tier;
	int n_used = 0;
	for (int c = 0; c < MULLS_NC; c++)
		n_used += rp.used[c];
	rp.bm_h0 = 0.0f;
	rp.bm_maxwords = rp.bm_stride = 0u;
	rp.grid_h0 = ctx->opt[MULLS_OPT_GRID_H0] > 0.0 ? std::max(0.05f, (float)ctx->opt[MULLS_OPT_GRID_H0]) : MULLS_GRID_H0;
	rp.lds_dedup = 0;
	rp.grid_maxcells = MULLS_MAXCELLS;
	rp.cell_stride = ((rp.grid_maxcells + 1u + 15u) & ~15u);
	bool resident = false;
	if (tier == 2 || tier == 3)
	{
		rp.grid_maxcells = lds_cells_for(lds_cap);
		// class-level jobs (one workgroup sees every query of a class cloud): keep the duplicate table in LDS if 4 B per target
		// still leave a useful cell budget next to the staged cloud (MULLS_ICP_STATIC_LDS bytes stay free for the static LDS of k_icp)
		const bool class_level = tier == 3 || (!B->cjobs_h.empty() && B->cjobs_h[0].count != MULLS_SRC_PER_BLOCK);
		const long left = 160L * 1024L - 64L - (long)MULLS_ICP_STATIC_LDS - (long)MULLS_LDS_QCHUNK * 16L - (long)MULLS_LDS_AUX - (long)lds_cap * 18L;
		const bool dedup_fits = !rp.normal_shooting && left / 2 - 8 >= 4096 && ctx->opt[MULLS_OPT_LDS_DEDUP] != 0.0; // k_nn_shoot uses the global table
		if (tier == 3 && !dedup_fits)
		{
			ctx->err = "internal: a mixed batch whose LDS-tier clouds do not fit the on-chip duplicate table";
			return MULLS_E_INVALID;
		}
		// Device-resident loop (k_icp: one workgroup carries a pair through all its iterations): only on request (nn_mode 4) since round 3 — the lock-step
		// path, whose light kernels run several workgroups per CU and whose per-iteration step runs on the device too, is at least as fast at every
		// batch size (profiles/r03_modes.txt); MULLS_OPT_RESIDENT_MIN_PAIRS .. _MAX_PAIRS can still open a window for it in auto mode.  It needs the
		// on-chip duplicate table, the plain mm_lls_icp loop (resident_out) and no source class cloud so large that one workgroup per pair would be the
		// wrong shape; nn_mode 4 gets the lock-step LDS tier where the loop does not apply.
		uint32_t max_src = 0;
		for (const Job &j : B->rjobs_h)
			max_src = std::max(max_src, j.count);
		resident = tier == 2 && resident_out && dedup_fits && max_src <= 16384u && P_jobs->max_iter_num > 0 &&
				   (ctx->nn_mode == 4 || (ctx->nn_mode == 0 && n >= (int)ctx->opt[MULLS_OPT_RESIDENT_MIN_PAIRS] && n <= (int)ctx->opt[MULLS_OPT_RESIDENT_MAX_PAIRS]));
		if ((class_level || resident) && dedup_fits)
		{
			rp.lds_dedup = 1;
			rp.grid_maxcells = (uint32_t)std::min<long>(left / 2 - 8, (long)MULLS_MAXCELLS);
		}
		rp.cell_stride = ((rp.grid_maxcells + 1u + 15u) & ~15u);
	}
	const size_t nl = B->lclouds_h.size();
	if (nl)
	{
		// occupancy-bitmap grids: bm_maxwords / bm_stride count 64-cell words per cloud
		rp.bm_h0 = MULLS_BM_H0;
		rp.bm_auto = 1;
		if (ctx->opt[MULLS_OPT_BM_H0] > 0.0) // diagnostics: one fixed cell edge for every cloud
		{
			rp.bm_h0 = std::max(0.05f, (float)ctx->opt[MULLS_OPT_BM_H0]);
			rp.bm_auto = 0;
		}
		size_t words = std::min<size_t>(MULLS_BM_MAXWORDS, MULLS_BM_TOTALWORDS / nl);
		words = std::max<size_t>(words & ~(size_t)15, 4096);
		rp.bm_maxwords = rp.bm_stride = (uint32_t)words;
	}

	if (resident_out)
		*resident_out = resident;

	bool grew = false, g2 = false;
	int rc = MULLS_OK;
	auto A = [&](int r) { if (rc == MULLS_OK) rc = r; };
	A(grow(ctx, &B->jobs, &B->cap_jobs[0], (size_t)B->njobs, &g2));
	grew |= g2;
	A(grow(ctx, &B->partial, &B->cap_jobs[1], (size_t)B->njobs * MULLS_NTERM));
	A(grow(ctx, &B->tjobs, &B->cap_jobs[2], B->tjobs_h.size(), &g2));
	grew |= g2;
	A(grow(ctx, &B->cjobs, &B->cap_jobs[3], B->cjobs_h.size(), &g2));
	grew |= g2;
	A(grow(ctx, &B->bjobs, &B->cap_bjobs[0], B->bjobs_h.size(), &g2));
	grew |= g2;
	A(grow(ctx, &B->fjobs, &B->cap_bjobs[1], B->fjobs_h.size(), &g2));
	grew |= g2;
	A(grow(ctx, &B->lclouds, &B->cap_bjobs[2], B->lclouds_h.size(), &g2));
	grew |= g2;
	A(grow(ctx, &B->ejobs, &B->cap_bjobs[3], B->ejobs_h.size(), &g2));
	grew |= g2;
	A(grow(ctx, &B->wl, &B->cap_wl, B->cjobs_h.size()));
	A(grow(ctx, &B->ajobs, &B->cap_ajobs, B->ajobs_h.size(), &g2));
	grew |= g2;
	if (resident)
	{
		A(grow(ctx, &B->rjobs, &B->cap_icp[0], B->rjobs_h.size(), &g2));
		grew |= g2;
		A(grow(ctx, &B->pair_rjob, &B->cap_icp[1], (size_t)n + 1, &g2));
		grew |= g2;
		A(grow(ctx, &B->order, &B->cap_icp[2], (size_t)n, &g2));
		grew |= g2;
		A(grow(ctx, &B->icp_outs, &B->cap_icp[3], (size_t)n));
		if (!B->icp_queue)
			A(dmalloc(ctx, &B->icp_queue, 16));
		if (rc == MULLS_OK)
			HIPCHK(ctx, hipMemsetAsync(B->icp_queue, 0, 16 * sizeof(uint32_t), st));
	}
	if (!B->wl_ctr)
		A(dmalloc(ctx, &B->wl_ctr, 16));
	if (rc == MULLS_OK)
		HIPCHK(ctx, hipMemsetAsync(B->wl_ctr, 0, 16 * sizeof(uint32_t), st));
	A(grow(ctx, &B->descs_init, &B->cap_jobs[4], B->descs_h.size(), &g2));
	grew |= g2;
	A(grow(ctx, &B->bbox_init, &B->cap_jobs[5], (size_t)n * 6, &g2));
	grew |= g2;
	if (tier == 2 || tier == 3)
		A(grow(ctx, &B->cell_start, &B->cap_cells[1], (size_t)n * n_used * rp.cell_stride));
	if (nl)
	{
		const size_t words = nl * (size_t)rp.bm_stride, cells = B->n_tgt + (size_t)n * MULLS_NC + 1;
		A(grow(ctx, &B->bm, &B->cap_bm, words));
		A(grow(ctx, &B->pf, &B->cap_pf, words));
		A(grow(ctx, &B->bm_cs, &B->cap_bm_cs, cells));
		A(grow(ctx, &B->cell_cnt, &B->cap_cells[0], cells));
		A(grow(ctx, &B->bm_rank, &B->cap_bm_rank, 2 * B->n_tgt)); // (counter index, arrival) per target point
		if (rc == MULLS_OK)
		{
			HIPCHK(ctx, hipMemsetAsync(B->cell_cnt, 0, cells * sizeof(uint32_t), st));
		}
	}
	if (rc != MULLS_OK)
		return rc;
	const std::string want_key = B->jobs_key + (resident ? "R" : "");
	if (grew || B->dev_key != want_key || B->dev_key.empty())
	{
		SegCopier up(ctx); // one launch for the dozen tables
		up.add_host(B->jobs, B->jobs_h.data(), sizeof(Job) * B->njobs);
		up.add_host(B->tjobs, B->tjobs_h.data(), sizeof(Job) * B->tjobs_h.size());
		up.add_host(B->cjobs, B->cjobs_dev_h.data(), sizeof(Job) * B->cjobs_dev_h.size());
		up.add_host(B->bjobs, B->bjobs_h.data(), sizeof(Job) * B->bjobs_h.size());
		up.add_host(B->fjobs, B->fjobs_h.data(), sizeof(Job) * B->fjobs_h.size());
		up.add_host(B->ejobs, B->ejobs_h.data(), sizeof(Job) * B->ejobs_h.size());
		up.add_host(B->lclouds, B->lclouds_h.data(), sizeof(uint32_t) * B->lclouds_h.size());
		up.add_host(B->ajobs, B->ajobs_h.data(), sizeof(uint32_t) * B->ajobs_h.size());
		if (resident)
		{
			up.add_host(B->rjobs, B->rjobs_h.data(), sizeof(Job) * B->rjobs_h.size());
			up.add_host(B->pair_rjob, B->pair_rjob_h.data(), sizeof(uint32_t) * B->pair_rjob_h.size());
			up.add_host(B->order, B->order_h.data(), sizeof(uint32_t) * B->order_h.size());
		}
		up.add_host(B->descs_init, B->descs_h.data(), sizeof(CloudDesc) * B->descs_h.size());
		up.add_host(B->bbox_init, B->bbox_h, sizeof(uint32_t) * 6 * n);
		if (up.flush(st) != MULLS_OK)
		{
			(void)hipStreamSynchronize(st);
			return MULLS_E_HIP;
		}
		HIPCHK(ctx, hipStreamSynchronize(st)); // the host vectors may be rebuilt by a later call
		B->dev_key = want_key;
	}
	{
		SegCopier reset(ctx); // this run's working copies of the descriptors and of the crop boxes
		reset.add_dev(B->descs, B->descs_init, sizeof(CloudDesc) * B->descs_h.size());
		reset.add_dev(B->bbox, B->bbox_init, sizeof(uint32_t) * 6 * n);
		if (reset.flush(st) != MULLS_OK)
			return MULLS_E_HIP;
	}
	return MULLS_OK;
}

// run-wide constants of the per-iteration algebra (the float conversions of cregistration.hpp:1150-1157)
mulls::IcpConst icp_const(const mulls_params *P)
{
	mulls::IcpConst K;
	K.max_iter_num = P->max_iter_num;
	K.converge_translation = P->converge_translation;
	K.converge_rotation = (float)(P->converge_rotation_d / 180.0 * M_PI);
	K.max_bearable_translation = (float)(2.0 * P->dis_thre_unit);
	K.max_bearable_rotation = (float)(P->max_bearable_rotation_d / 180.0 * M_PI);
	K.dis_thre_unit = P->dis_thre_unit;
	K.dis_thre_min = P->dis_thre_min;
	K.dis_thre_update_rate = P->dis_thre_update_rate;
	K.min_neccessary_corr_ratio = P->min_neccessary_corr_ratio;
	K.sigma_thre = P->sigma_thre;
	return K;
}

// Reserve `n` consecutive epochs of the batch's duplicate table.  The winner key is (descending epoch << 32 | source index)
// under atomicMin, so newer epochs must sort below older ones: before the 32-bit counter would wrap, the table is refilled
// with 0xff and the count restarts (stream order puts the fill before this run's kernels).
int take_epochs(mulls_ctx *ctx, mulls_batch *B, uint32_t n, RunParams &rp)
{
	uint32_t *base = &rp.tick_base;
	rp.cand = B->nn_cand;
	if (ctx->opt[MULLS_OPT_DEBUG_TICK] > 0.0 && B->tick == 1) // tests only: put a fresh batch's counter next to the wrap
		B->tick = (uint32_t)ctx->opt[MULLS_OPT_DEBUG_TICK];
	if (B->tick > 0xfffffff0u - n)
	{
		if (B->winner)
			HIPCHK(ctx, hipMemsetAsync(B->winner, 0xff, B->cap_tgt[3] * sizeof(unsigned long long), ctx->stream));
		if (B->nn_cand) // the candidate records carry epochs of the same counter
			HIPCHK(ctx, hipMemsetAsync(B->nn_cand, 0, B->cap_src[11] * sizeof(uint4), ctx->stream));
		B->tick = 1;
	}
	*base = B->tick;
	B->tick += n;
	return MULLS_OK;
}
} // namespace mulls_drv
