// driver.cpp — host side of libmulls_hip.so: the C ABI of include/mulls_hip.h, device-resident batches and the
// lock-step ICP iteration loop (reference: CRegistration::mm_lls_icp, include/common/cregistration.hpp:1114-1440).
//
// Division of labour (BASELINE.json north_star): correspondences, rejection, the normal-equation reduction AND the per-iteration 6x6 solve with
// its step / convergence / health tests run in the HIP kernels of k_setup / k_grid / k_search / k_reduce / k_icp .hip; the host queues launch
// sets, reads one 8-byte word per set to learn how many pairs still iterate, and downloads the result records at the end (batches of up to
// 1024 pairs: one launch, k_icp).  Only when the caller asks for per-iteration traces does the host step the loop itself: per iteration and
// pair it then receives the combined 27-double system + a few counters, solves it with the same icp_step.h functions and writes the next
// PairState.  There is no CPU fallback anywhere in this file: without a usable HIP device every entry point fails with
// MULLS_E_NO_DEVICE / MULLS_E_HIP.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mulls_hip.h"
#include "device_types.h"
#include "hostmath.h"
#include "icp_step.h"
#include "launch.h"
#include "ctx.h"


using mulls::Mat4;
using mulls::Mat6;

struct mulls_batch
{
	int n = 0;
	size_t n_src = 0, n_tgt = 0; // staged points over all pairs and classes
	std::vector<CloudDesc> descs_h;
	std::vector<PairSetup> setup_h;
	std::vector<Job> setup_jobs_h;
	std::vector<Job> big_segs_h, big_clouds_h; // target class clouds cropped segment-wise (k_crop_big_*): segments, clouds
	std::vector<Job> jobs_h;
	std::vector<Job> cjobs_h; // one entry per (pair, used class) with source points: the LDS tier's unit of work
	std::vector<Job> cjobs_dev_h; // the same entries as uploaded: inside each sub-batch's slice the most expensive class clouds come first
	std::vector<Job> tjobs_h; // target-side chunks (256 points) of the used classes, for the grid build
	std::vector<uint32_t> ajobs_h; // jobs that start a trip of 1024 source slots: k_accum's workgroups (indices into jobs_h) — per sub-batch slice,
								   // and inside a slice grouped by trip length (ajob_split)
	uint32_t ajob_split[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}; // [sub-batch][0..3]: the slice's trips of > 512, 257..512, <= 256 slots
	// device-resident loop (k_icp): class-level jobs in pair order, each pair's range in them, the pairs most expensive first
	std::vector<Job> rjobs_h;
	std::vector<uint32_t> pair_rjob_h, order_h;
	std::vector<IcpOut> icp_outs_h;
	std::vector<mulls_iter_trace> trace_h;
	std::string jobs_key;
	uint32_t njobs = 0;
	// device
	float4 *stage = nullptr;
	float4 *tmp_pos = nullptr, *tmp_nrm = nullptr;
	float4 *spos = nullptr, *snrm = nullptr, *tpos = nullptr, *tnrm = nullptr;
	uint8_t *flag = nullptr;
	int32_t *match = nullptr, *nn_idx = nullptr, *nn_hint = nullptr;
	float4 *mq = nullptr; // per source point: position and direction of its matched target (2 records), written with match[]
	float *wd = nullptr, *nn_d2 = nullptr;
	unsigned long long *winner = nullptr;
	uint32_t tick = 1; // duplicate-table epoch counter of THIS batch's winner table, monotone between resets (take_epochs)
	CloudDesc *descs = nullptr;
	PairSetup *setup = nullptr;
	PairState *states = nullptr;	 // HBM copy of the pair states (filled by k_push_states every iteration)
	PairState *states_pin = nullptr; // device address of the pinned host array states_h
	PairOut *outs = nullptr;	   // HBM: filled by k_finish
	size_t cap_outs = 0;
	PairOut *outs_pin = nullptr; // device address of the pinned host array outs_h (packed records, k_pull_outs)
	uint32_t *bbox = nullptr;
	Job *setup_jobs = nullptr;
	Job *big_segs = nullptr, *big_clouds = nullptr;
	uint32_t *seg_cnt = nullptr, *big_box = nullptr;
	size_t cap_big[4] = {};
	Job *jobs = nullptr;
	double *partial = nullptr;
	Job *tjobs = nullptr;
	Job *cjobs = nullptr;
	Job *rjobs = nullptr;
	uint32_t *ajobs = nullptr;
	size_t cap_ajobs = 0;
	uint32_t *pair_rjob = nullptr, *order = nullptr, *icp_queue = nullptr;
	IcpOut *icp_outs = nullptr;
	mulls_iter_trace *trace_dev = nullptr;
	size_t cap_icp[5] = {};
	mulls::StepState *steps = nullptr; // lock-step loop with the device step: per-pair loop state
	size_t cap_steps = 0;
	uint32_t epoch2 = 0; // ... and the last epoch issued on its 8-byte word (words 32-33 of epoch_h)
	int nsub = 1;		 // sub-batches the job tables are laid out for (build_jobs)
	double fill_ms = 0.0, fill_pack_ms = 0.0; // the last batch_fill: wall time, host packing time ...
	uint64_t fill_bytes = 0;				   // ... and bytes staged
	uint32_t *wl = nullptr;		// LDS tier: class clouds k_cert queued for k_nn_lds (one slot per class-level job)
	uint32_t *wl_ctr = nullptr; // ... and the queue counters: per sub-batch 8 words = (queued, taken) x launch parity
	size_t cap_wl = 0;
	GridDesc *grids = nullptr;
	float4 *tsorted = nullptr;
	uint32_t *cell_cnt = nullptr, *cell_start = nullptr; // global tier: per-occupied-cell counters / start positions; LDS tier: dense cell table
	unsigned long long *bm = nullptr;					  // global tier: occupancy words of every grid
	uint32_t *pf = nullptr;								  // global tier: occupied cells before each word
	size_t cap_bm = 0, cap_pf = 0;
	// pinned, device-mapped host memory (zero-copy): per-iteration pair states in, per-pair sums out, completion epoch
	PairState *states_h = nullptr;
	PairOut *outs_h = nullptr;
	volatile uint32_t *epoch_h = nullptr;
	uint32_t *epoch_dev = nullptr;
	uint32_t epoch = 0;			// last epoch issued on word 0 (sub-batch 0 and the single-shot entry points)
	uint32_t epoch1 = 0;		// last epoch issued on word 16 (sub-batch 1)
	uint32_t *ticket = nullptr; // device: arrival counters of k_finish (one per sub-batch, 16 words apart)
	uint32_t *bbox_h = nullptr;
	uint8_t *upload_h = nullptr; // pinned staging buffer of the caller's point records
	CloudDesc *descs_init = nullptr; // pristine descriptors (device): restored into `descs` by a D2D copy every run
	uint32_t *bbox_init = nullptr;
	std::string dev_key;			 // jobs_key of the tables currently resident on the device
	size_t cap_jobs[6] = {}, cap_cells[2] = {};
	// capacities (elements) of the grow-only arrays
	size_t cap_stage = 0, cap_src[11] = {}, cap_tgt[4] = {}, cap_pairs[5] = {}, cap_setup_jobs = 0, cap_pin[4] = {};
};

namespace
{

inline float ord_to_float(uint32_t k)
{
	const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
	float f;
	std::memcpy(&f, &u, sizeof(f));
	return f;
}

inline int metric_of(int c) { return (c == MULLS_PILLAR || c == MULLS_BEAM) ? 1 : (c == MULLS_VERTEX ? 2 : 0); }

void rows12(const double colmajor[16], double out[12])
{
	for (int r = 0; r < 3; r++)
		for (int c = 0; c < 4; c++)
			out[r * 4 + c] = colmajor[r + 4 * c];
}

// host-side life of one pair during a run: the shared per-iteration state (icp_step.h) + host-only bookkeeping
struct PairHost : mulls::PairIter
{
	bool first = true;
	uint32_t alive_prev[MULLS_NC];
};

// packed index of (r,c), r <= c, in the row-major-upper enumeration used by k_accum
inline int packed(int r, int c) { return r * 6 - r * (r - 1) / 2 + (c - r); }

// The 6x6 the reference inverts and its right-hand side, as k_finish combined them from the class rows (lower / upper triangle
// bookkeeping of cregistration.hpp:1914-1938 happens there)
void normal_from_comb(const PairOut &o, Mat6 &N, double b[6])
{
	for (int r = 0; r < 6; r++)
		for (int c = r; c < 6; c++)
		{
			const double val = o.comb[packed(r, c)];
			N.at(c, r) = val;
			N.at(r, c) = val;
		}
	for (int j = 0; j < 6; j++)
		b[j] = o.comb[21 + j];
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
}

// sub-batches in flight of a host-stepped lock-step batch of n pairs
int subbatch_count(const mulls_ctx *ctx, int n)
{
	int nsub = n >= 2048 ? 2 : 1; // below that the half-size launches cost more (k_nn_lds tail) than the overlap returns
	if (ctx->opt[MULLS_OPT_SUBBATCHES] >= 1.0)
		nsub = std::max(1, std::min(2, (int)ctx->opt[MULLS_OPT_SUBBATCHES]));
	return n < 2 ? 1 : nsub;
}

// defaults of enum mulls_option, then the presets from the environment (read here and nowhere else)
void options_init(mulls_ctx *ctx)
{
	double *o = ctx->opt;
	o[MULLS_OPT_HOST_STEP] = 0, o[MULLS_OPT_RESIDENT_MIN_PAIRS] = 160, o[MULLS_OPT_RESIDENT_MAX_PAIRS] = 320, o[MULLS_OPT_FEW_LAUNCHES_MAX_PAIRS] = 384;
	o[MULLS_OPT_SUBBATCHES] = 0, o[MULLS_OPT_TWO_STREAMS] = 0, o[MULLS_OPT_CERTIFICATES] = 1;
	o[MULLS_OPT_CERT_SLACK_MIN] = 0.02, o[MULLS_OPT_CERT_SLACK_MAX] = 0.10, o[MULLS_OPT_CERT_SLACK_RATE] = 1.0;
	o[MULLS_OPT_LDS_DEDUP] = 1, o[MULLS_OPT_GRID_H0] = 0, o[MULLS_OPT_BM_H0] = 0, o[MULLS_OPT_LEAN_STAGING] = 0, o[MULLS_OPT_DEBUG_STOP] = 0, o[MULLS_OPT_DEBUG_TICK] = 0;
	static const struct
	{
		const char *name;
		int opt;
	} env[] = {{"MULLS_HOST_STEP", MULLS_OPT_HOST_STEP}, {"MULLS_RESIDENT_MIN_PAIRS", MULLS_OPT_RESIDENT_MIN_PAIRS}, {"MULLS_RESIDENT_MAX_PAIRS", MULLS_OPT_RESIDENT_MAX_PAIRS},
			   {"MULLS_FEW_LAUNCHES_MAX_PAIRS", MULLS_OPT_FEW_LAUNCHES_MAX_PAIRS}, {"MULLS_SUBBATCHES", MULLS_OPT_SUBBATCHES}, {"MULLS_TWO_STREAMS", MULLS_OPT_TWO_STREAMS},
			   {"MULLS_CERTIFICATES", MULLS_OPT_CERTIFICATES}, {"MULLS_LDS_DEDUP", MULLS_OPT_LDS_DEDUP}, {"MULLS_GRID_H0", MULLS_OPT_GRID_H0}, {"MULLS_BM_H0", MULLS_OPT_BM_H0},
			   {"MULLS_LEAN_STAGING", MULLS_OPT_LEAN_STAGING}, {"MULLS_DEBUG_STOP", MULLS_OPT_DEBUG_STOP}, {"MULLS_DEBUG_TICK", MULLS_OPT_DEBUG_TICK}};
	for (const auto &e : env)
		if (const char *v = std::getenv(e.name))
			o[e.opt] = std::strtod(v, nullptr);
	if (std::getenv("MULLS_NO_CERT"))
		o[MULLS_OPT_CERTIFICATES] = 0;
	if (std::getenv("MULLS_NO_LDS_DEDUP"))
		o[MULLS_OPT_LDS_DEDUP] = 0;
	if (std::getenv("MULLS_NO_RESIDENT"))
		o[MULLS_OPT_RESIDENT_MAX_PAIRS] = 0;
	if (const char *e = std::getenv("MULLS_CERT_SLACK")) // "min,max,rate"
	{
		float a = 0, b = 0, c = 0;
		if (std::sscanf(e, "%f,%f,%f", &a, &b, &c) == 3 && a >= 0.0f && b >= a && c >= 0.0f)
			o[MULLS_OPT_CERT_SLACK_MIN] = a, o[MULLS_OPT_CERT_SLACK_MAX] = b, o[MULLS_OPT_CERT_SLACK_RATE] = c;
	}
}

void build_jobs(mulls_batch *B, const mulls_params *P, int nsub)
{
	std::string key(P->used_feature_type, 6);
	key += (char)('0' + nsub);
	if (key == B->jobs_key)
		return;
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
	B->cjobs_h.clear();
	for (int p = 0; p < B->n; p++)
		for (int c = 0; c < MULLS_NC; c++)
			if (P->used_feature_type[c] == '1' && B->descs_h[p * MULLS_NC + c].src_cap > 0)
			{
				Job j = {(uint32_t)p, (uint32_t)c, 0u, B->descs_h[p * MULLS_NC + c].src_cap};
				B->cjobs_h.push_back(j);
			}
	uint32_t max_src_cap = 0;
	for (const Job &j : B->cjobs_h)
		max_src_cap = std::max(max_src_cap, j.count);
	if (B->cjobs_h.size() < 512 && max_src_cap > 4096u)
	{
		// few AND large class clouds (a pair of dense scans): split them into 512-query jobs (each stages its target cloud itself) so that more than a
		// handful of workgroups walk them.  Down-sampled class clouds stay whole whatever the batch size: a class-level job resolves the duplicate
		// rule and the rejection chain itself (no k_filter launch), and a small batch is bound by the number of launches, not by their width
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
			if (P->used_feature_type[c] == '1')
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
int wait_epoch_word(mulls_ctx *ctx, volatile uint32_t *word, uint32_t want, hipEvent_t last = nullptr, hipStream_t stream = nullptr);
int wait_epoch(mulls_ctx *ctx, mulls_batch *B) { return wait_epoch_word(ctx, B->epoch_h, B->epoch); }
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
void unpack_out(const mulls_batch *B, const uint8_t used[MULLS_NC], int p, PairOut &o, bool comb = false)
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

// search tier of a run: 0 = LDS-tiled brute force, 1 = uniform grid in global memory, 2 = uniform grid staged in LDS
int choose_tier(const mulls_ctx *ctx, const mulls_batch *B, const uint8_t used[MULLS_NC], uint32_t *lds_cap)
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
		// the LDS tier whenever the clouds fit, whatever the batch size: with class-level jobs and four launches per iteration one KITTI pair takes
		// 0.84 ms there against 0.94 ms on the global-memory tier (profiles/r03_modes.txt)
		return fits ? 2 : 1;
	}
}

// float4 units a staged cloud of n points takes (device_types.h: MULLS_STAGE_*)
inline size_t stage_quads(uint32_t n, uint32_t fmt) { return fmt == MULLS_STAGE_AOS48 ? (size_t)n * 3 : (fmt == MULLS_STAGE_PACK32 ? (size_t)n * 2 : (size_t)n + ((size_t)n * 3 + 3) / 4); }

// lay the pairs out in the batch arenas, (re)allocate what is too small and stage the caller's clouds in HBM.  Host clouds are gathered out of the
// caller's records into the packed layouts (32 of the 48 bytes are live; 28 when the run is known not to undistort: P given); class clouds of a
// device-resident local map keep their 48-byte records and are copied device to device.  P + MULLS_OPT_LEAN_STAGING: the clouds the run never
// reads are staged as empty.
int batch_fill(mulls_ctx *ctx, mulls_batch *B, const mulls_pair *pairs, int n, const mulls_params *P = nullptr)
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
			}
			for (uint32_t k = 0; k < d.src_cap; k += MULLS_BLOCK)
			{
				Job j = {(uint32_t)p, (uint32_t)c, k, 0};
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
			B->setup_h[p].inv_t[3] = 0.0;
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
	bool winner_grew = false;
	A(grow(ctx, &B->stage, &B->cap_stage, stage_rec));
	A(grow(ctx, &B->tmp_pos, &B->cap_src[0], so));
	A(grow(ctx, &B->tmp_nrm, &B->cap_src[1], so));
	A(grow(ctx, &B->spos, &B->cap_src[2], so));
	A(grow(ctx, &B->snrm, &B->cap_src[3], so));
	A(grow(ctx, &B->flag, &B->cap_src[4], so));
	A(grow(ctx, &B->match, &B->cap_src[5], so));
	A(grow(ctx, &B->nn_idx, &B->cap_src[6], so));
	A(grow(ctx, &B->wd, &B->cap_src[7], so));
	A(grow(ctx, &B->nn_d2, &B->cap_src[8], so));
	A(grow(ctx, &B->nn_hint, &B->cap_src[9], 2 * so)); // LDS tier: (hint word, bound) records
	A(grow(ctx, &B->mq, &B->cap_src[10], 2 * so));
	A(grow(ctx, &B->tpos, &B->cap_tgt[0], to));
	A(grow(ctx, &B->tnrm, &B->cap_tgt[1], to));
	A(grow(ctx, &B->tsorted, &B->cap_tgt[2], to));
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
	A(grow(ctx, &B->big_box, &B->cap_big[3], B->big_clouds_h.size() * 6));
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
		const int threads = (int)std::max<long>(1, std::min<long>(32, (c1 - c0) / 6));
		(void)threads;
		const auto t_pack0 = std::chrono::steady_clock::now();
#pragma omp parallel for num_threads(threads) schedule(dynamic, 2) if (threads > 1)
		for (long i = c0; i < c1; i++)
		{
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
		}
		pack_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_pack0).count();
		// the staged clouds of pairs [p0, p1) are one contiguous range (offsets grow with the pair index)
		const size_t r0 = B->descs_h[(size_t)p0 * MULLS_NC].src_stage;
		const size_t r1 = p1 < n ? B->descs_h[(size_t)p1 * MULLS_NC].src_stage : stage_rec;
		if (r1 > r0)
			e = hipMemcpyAsync(reinterpret_cast<uint8_t *>(B->stage) + r0 * 16, B->upload_h + r0 * 16, (r1 - r0) * 16, hipMemcpyHostToDevice, st);
	}
	for (const DevCopy &dc : dev_copies)
		if (e == hipSuccess)
			e = hipMemcpyAsync(reinterpret_cast<uint8_t *>(B->stage) + dc.dst, dc.src, dc.bytes, hipMemcpyDeviceToDevice, st);
	if (e == hipSuccess)
		e = hipMemcpyAsync(B->setup_jobs, B->setup_jobs_h.data(), B->setup_jobs_h.size() * sizeof(Job), hipMemcpyHostToDevice, st);
	if (e == hipSuccess && !B->big_segs_h.empty())
		e = hipMemcpyAsync(B->big_segs, B->big_segs_h.data(), B->big_segs_h.size() * sizeof(Job), hipMemcpyHostToDevice, st);
	if (e == hipSuccess && !B->big_clouds_h.empty())
		e = hipMemcpyAsync(B->big_clouds, B->big_clouds_h.data(), B->big_clouds_h.size() * sizeof(Job), hipMemcpyHostToDevice, st);
	if (e == hipSuccess)
		e = hipMemcpyAsync(B->setup, B->setup_h.data(), sizeof(PairSetup) * n, hipMemcpyHostToDevice, st);
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
int prepare_run(mulls_ctx *ctx, mulls_batch *B, const mulls_params *P_jobs, RunParams &rp, uint32_t *lds_cap_out, int *tier_out, bool *resident_out = nullptr, int nsub = 0)
{
	hipStream_t st = ctx->stream;
	const int n = B->n;
	const std::string old_key = B->jobs_key;
	build_jobs(B, P_jobs, nsub > 0 ? nsub : subbatch_count(ctx, n));
	uint32_t lds_cap = 0;
	const int tier = choose_tier(ctx, B, rp.used, &lds_cap);
	if (tier < 0)
	{
		ctx->err = "nn mode 3 (grid staged in LDS) needs every searched target class cloud to hold <= 9728 points";
		return MULLS_E_INVALID;
	}
	*lds_cap_out = lds_cap;
	*tier_out = tier;
	int n_used = 0;
	for (int c = 0; c < MULLS_NC; c++)
		n_used += rp.used[c];
	rp.bm_h0 = 0.0f;
	rp.grid_h0 = ctx->opt[MULLS_OPT_GRID_H0] > 0.0 ? std::max(0.05f, (float)ctx->opt[MULLS_OPT_GRID_H0]) : MULLS_GRID_H0;
	rp.lds_dedup = 0;
	bool resident = false;
	if (tier == 2)
	{
		rp.grid_maxcells = lds_cells_for(lds_cap);
		// class-level jobs (one workgroup sees every query of a class cloud): keep the duplicate table in LDS if 4 B per target
		// still leave a useful cell budget next to the staged cloud (MULLS_ICP_STATIC_LDS bytes stay free for the static LDS of k_icp)
		const bool class_level = !B->cjobs_h.empty() && B->cjobs_h[0].count != MULLS_SRC_PER_BLOCK;
		const long left = 160L * 1024L - 64L - (long)MULLS_ICP_STATIC_LDS - (long)MULLS_LDS_QCHUNK * 16L - (long)MULLS_LDS_AUX - (long)lds_cap * 18L;
		const bool dedup_fits = !rp.normal_shooting && left / 2 - 8 >= 4096 && ctx->opt[MULLS_OPT_LDS_DEDUP] != 0.0; // k_nn_shoot uses the global table
		// Device-resident loop (k_icp: one workgroup carries a pair through all its iterations): the default whenever the LDS tier
		// applies with its on-chip duplicate table, the loop is the plain mm_lls_icp one (resident_out) and no source class cloud is
		// so large that one workgroup per pair would be the wrong shape (those pairs are spread over many workgroups by the
		// lock-step path).  In auto mode it runs batches of MULLS_RESIDENT_MIN_PAIRS .. MAX_PAIRS pairs, where it is the faster of the two
		// (measured, tools/gpu_modes.py, profiles/r02_zzz_modes.txt: 73 k vs 67 k registrations/s at 128 pairs; the lock-step path, whose
		// light kernels run several workgroups per CU and whose per-iteration step runs on the device too, wins from 512 pairs on — 151 k vs
		// 142 k, 174 k vs 146 k at 1024 — and the two tie below ~40).  nn_mode 3 keeps the lock-step LDS tier; nn_mode 4 asks for the
		// resident loop (and gets the lock-step LDS tier where the loop does not apply).
		uint32_t max_src = 0;
		for (const Job &j : B->rjobs_h)
			max_src = std::max(max_src, j.count);
		resident = resident_out && dedup_fits && max_src <= 16384u && P_jobs->max_iter_num > 0 && (ctx->nn_mode == 4 || (ctx->nn_mode == 0 && n >= (int)ctx->opt[MULLS_OPT_RESIDENT_MIN_PAIRS] && n <= (int)ctx->opt[MULLS_OPT_RESIDENT_MAX_PAIRS]));
		if ((class_level || resident) && dedup_fits)
		{
			rp.lds_dedup = 1;
			rp.grid_maxcells = (uint32_t)std::min<long>(left / 2 - 8, (long)MULLS_MAXCELLS);
		}
		rp.cell_stride = ((rp.grid_maxcells + 1u + 15u) & ~15u);
	}
	else if (tier == 1)
	{
		// occupancy-bitmap grids: grid_maxcells / cell_stride count 64-cell words per cloud
		rp.bm_h0 = MULLS_BM_H0;
		rp.bm_auto = 1;
		if (ctx->opt[MULLS_OPT_BM_H0] > 0.0) // diagnostics: one fixed cell edge for every cloud
		{
			rp.bm_h0 = std::max(0.05f, (float)ctx->opt[MULLS_OPT_BM_H0]);
			rp.bm_auto = 0;
		}
		const size_t clouds = std::max<size_t>((size_t)n * std::max(n_used, 1), 1);
		size_t words = std::min<size_t>(MULLS_BM_MAXWORDS, MULLS_BM_TOTALWORDS / clouds);
		words = std::max<size_t>(words & ~(size_t)15, 4096);
		rp.grid_maxcells = (uint32_t)words;
		rp.cell_stride = (uint32_t)words;
	}
	else
	{
		rp.grid_maxcells = MULLS_MAXCELLS;
		rp.cell_stride = ((rp.grid_maxcells + 1u + 15u) & ~15u);
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
	if (tier == 2)
		A(grow(ctx, &B->cell_start, &B->cap_cells[1], (size_t)n * n_used * rp.cell_stride));
	else if (tier == 1)
	{
		const size_t words = (size_t)n * n_used * rp.cell_stride, cells = B->n_tgt + (size_t)n * MULLS_NC + 1;
		A(grow(ctx, &B->bm, &B->cap_bm, words));
		A(grow(ctx, &B->pf, &B->cap_pf, words));
		A(grow(ctx, &B->cell_start, &B->cap_cells[1], cells));
		A(grow(ctx, &B->cell_cnt, &B->cap_cells[0], cells));
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
		HIPCHK(ctx, hipMemcpyAsync(B->jobs, B->jobs_h.data(), sizeof(Job) * B->njobs, hipMemcpyHostToDevice, st));
		HIPCHK(ctx, hipMemcpyAsync(B->tjobs, B->tjobs_h.data(), sizeof(Job) * B->tjobs_h.size(), hipMemcpyHostToDevice, st));
		HIPCHK(ctx, hipMemcpyAsync(B->cjobs, B->cjobs_dev_h.data(), sizeof(Job) * B->cjobs_dev_h.size(), hipMemcpyHostToDevice, st));
		HIPCHK(ctx, hipMemcpyAsync(B->ajobs, B->ajobs_h.data(), sizeof(uint32_t) * B->ajobs_h.size(), hipMemcpyHostToDevice, st));
		if (resident)
		{
			HIPCHK(ctx, hipMemcpyAsync(B->rjobs, B->rjobs_h.data(), sizeof(Job) * B->rjobs_h.size(), hipMemcpyHostToDevice, st));
			HIPCHK(ctx, hipMemcpyAsync(B->pair_rjob, B->pair_rjob_h.data(), sizeof(uint32_t) * B->pair_rjob_h.size(), hipMemcpyHostToDevice, st));
			HIPCHK(ctx, hipMemcpyAsync(B->order, B->order_h.data(), sizeof(uint32_t) * B->order_h.size(), hipMemcpyHostToDevice, st));
		}
		HIPCHK(ctx, hipMemcpyAsync(B->descs_init, B->descs_h.data(), sizeof(CloudDesc) * B->descs_h.size(), hipMemcpyHostToDevice, st));
		HIPCHK(ctx, hipMemcpyAsync(B->bbox_init, B->bbox_h, sizeof(uint32_t) * 6 * n, hipMemcpyHostToDevice, st));
		HIPCHK(ctx, hipStreamSynchronize(st)); // the host vectors may be rebuilt by a later call
		B->dev_key = want_key;
	}
	HIPCHK(ctx, hipMemcpyAsync(B->descs, B->descs_init, sizeof(CloudDesc) * B->descs_h.size(), hipMemcpyDeviceToDevice, st));
	HIPCHK(ctx, hipMemcpyAsync(B->bbox, B->bbox_init, sizeof(uint32_t) * 6 * n, hipMemcpyDeviceToDevice, st));
	(void)old_key;
	return MULLS_OK;
}

struct EvTimer
{
	mulls_ctx *ctx;
	hipStream_t stream = nullptr; // where its events are recorded (default: ctx->stream)
	int base = 0; // first event of this timer's set in ctx->ev
	int used = 0;
	bool open = false;
	double *slot[5];
	void begin(double *acc)
	{
		open = ctx->profiling == 1 || (ctx->profiling == 2 && acc == &ctx->prof.ms_nn);
		if (!open)
			return;
		slot[used / 2] = acc;
		(void)hipEventRecord(ctx->ev[base + used], stream ? stream : ctx->stream);
	}
	void end()
	{
		if (!open)
			return;
		open = false;
		(void)hipEventRecord(ctx->ev[base + used + 1], stream ? stream : ctx->stream);
		used += 2;
	}
	hipEvent_t last() const { return (ctx->profiling == 1 && used) ? ctx->ev[base + used - 1] : nullptr; }
	void collect() // after the last recorded event completed
	{
		for (int i = 0; i < used; i += 2)
		{
			float ms = 0;
			if (hipEventElapsedTime(&ms, ctx->ev[base + i], ctx->ev[base + i + 1]) != hipSuccess)
			{
				(void)hipEventSynchronize(ctx->ev[base + i + 1]);
				(void)hipEventElapsedTime(&ms, ctx->ev[base + i], ctx->ev[base + i + 1]);
			}
			*slot[i / 2] += ms;
		}
		used = 0;
	}
};

} // namespace

// run-wide constants of the per-iteration algebra (the float conversions of cregistration.hpp:1150-1157)
static mulls::IcpConst icp_const(const mulls_params *P)
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
static int take_epochs(mulls_ctx *ctx, mulls_batch *B, uint32_t n, uint32_t *base)
{
	if (ctx->opt[MULLS_OPT_DEBUG_TICK] > 0.0 && B->tick == 1) // tests only: put a fresh batch's counter next to the wrap
		B->tick = (uint32_t)ctx->opt[MULLS_OPT_DEBUG_TICK];
	if (B->tick > 0xfffffff0u - n)
	{
		if (B->winner)
			HIPCHK(ctx, hipMemsetAsync(B->winner, 0xff, B->cap_tgt[3] * sizeof(unsigned long long), ctx->stream));
		B->tick = 1;
	}
	*base = B->tick;
	B->tick += n;
	return MULLS_OK;
}

extern "C"
{

	void mulls_default_params(mulls_params *p)
	{
		std::memset(p, 0, sizeof(*p));
		p->max_iter_num = 20;
		p->dis_thre_unit = 1.5f;
		p->converge_translation = 0.002f;
		p->converge_rotation_d = 0.01f;
		p->dis_thre_min = 0.4f;
		p->dis_thre_update_rate = 1.1f;
		std::strcpy(p->used_feature_type, "111110");
		std::strcpy(p->weight_strategy, "1101");
		p->z_xy_balanced_ratio = 1.0f;
		p->pt2pt_residual_window = 0.1f;
		p->pt2pl_residual_window = 0.1f;
		p->pt2li_residual_window = 0.1f;
		p->apply_intersection_filter = 1;
		p->normal_bearing = 45.0f;
		p->faithful = 1;
		p->rejector_strict = 1; // pcl::registration::CorrespondenceRejectorDistance: `distance < max_distance_` (correspondence_rejection_distance.cpp)
		p->sigma_thre = 0.5f;
		p->min_neccessary_corr_ratio = 0.03f;
		p->max_bearable_rotation_d = 45.0f;
	}

	int mulls_create(int device, mulls_ctx **out)
	try
	{
		if (!out)
			return MULLS_E_INVALID;
		*out = nullptr;
		int count = 0;
		if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
			return MULLS_E_NO_DEVICE;
		mulls_ctx *ctx = new mulls_ctx();
		ctx->device = device;
		options_init(ctx);
		if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
			hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess ||
			hipEventCreateWithFlags(&ctx->ev_setup, hipEventDisableTiming) != hipSuccess)
		{
			delete ctx;
			return MULLS_E_NO_DEVICE;
		}
		for (auto &e : ctx->ev)
			(void)hipEventCreate(&e);
		*out = ctx;
		return MULLS_OK;
	}
	catch (...)
	{
		return mulls::abi_caught(nullptr); // nothing is thrown across the ABI
	}

	void mulls_destroy(mulls_ctx *ctx)
	{
		if (!ctx)
			return;
		(void)hipSetDevice(ctx->device);
		if (ctx->gf_buf)
			(void)hipFree(ctx->gf_buf);
		if (ctx->gf_rnd)
			(void)hipFree(ctx->gf_rnd);
		if (ctx->cl_buf)
			(void)hipFree(ctx->cl_buf);
		if (ctx->scratch)
			mulls_batch_destroy(ctx, ctx->scratch);
		while (!ctx->maps.empty()) // local maps die with their context (mulls_map_destroy unregisters them)
			mulls_map_destroy(ctx, ctx->maps.back());
		while (!ctx->blocks.empty()) // ... and so do feature blocks
			mulls_block_destroy(ctx, ctx->blocks.back());
		for (auto &e : ctx->ev)
			if (e)
				(void)hipEventDestroy(e);
		if (ctx->ev_setup)
			(void)hipEventDestroy(ctx->ev_setup);
		if (ctx->stream2)
			(void)hipStreamDestroy(ctx->stream2);
		if (ctx->stream)
			(void)hipStreamDestroy(ctx->stream);
		delete ctx;
	}

	const char *mulls_last_error(const mulls_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

	int mulls_set_profiling(mulls_ctx *ctx, int on)
	try
	{
		if (!ctx)
			return MULLS_E_INVALID;
		ctx->profiling = on == 2 ? 2 : (on != 0 ? 1 : 0);
		return MULLS_OK;
	}
	catch (...)
	{
		return mulls::abi_caught(const_cast<mulls_ctx *>(ctx)); // nothing is thrown across the ABI
	}

	int mulls_get_profile(const mulls_ctx *ctx, mulls_profile *out)
	try
	{
		if (!ctx || !out)
			return MULLS_E_INVALID;
		*out = ctx->prof;
		return MULLS_OK;
	}
	catch (...)
	{
		return mulls::abi_caught(const_cast<mulls_ctx *>(ctx)); // nothing is thrown across the ABI
	}

	void *mulls_stream(mulls_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

	int mulls_set_option(mulls_ctx *ctx, int option, double value)
	{
		if (!ctx || option < 0 || option >= MULLS_OPT_COUNT || !(value == value))
			return MULLS_E_INVALID;
		ctx->opt[option] = value;
		return MULLS_OK;
	}
	int mulls_get_option(const mulls_ctx *ctx, int option, double *value)
	{
		if (!ctx || !value || option < 0 || option >= MULLS_OPT_COUNT)
			return MULLS_E_INVALID;
		*value = ctx->opt[option];
		return MULLS_OK;
	}

	int mulls_set_nn_mode(mulls_ctx *ctx, int mode)
	try
	{
		if (!ctx || mode < 0 || mode > 4)
			return MULLS_E_INVALID;
		ctx->nn_mode = mode;
		return MULLS_OK;
	}
	catch (...)
	{
		return mulls::abi_caught(const_cast<mulls_ctx *>(ctx)); // nothing is thrown across the ABI
	}

	void mulls_batch_destroy(mulls_ctx *ctx, mulls_batch *B)
	{
		if (!B)
			return;
		if (ctx)
			(void)hipSetDevice(ctx->device);
		void *dev[] = {B->stage, B->tmp_pos, B->tmp_nrm, B->spos, B->snrm, B->tpos, B->tnrm, B->flag, B->match, B->nn_idx, B->nn_hint, B->mq, B->wd,
					   B->nn_d2, B->winner, B->descs, B->setup, B->states, B->outs, B->ticket, B->bbox, B->setup_jobs, B->big_segs, B->big_clouds, B->seg_cnt, B->big_box, B->jobs, B->partial,
					   B->tjobs, B->cjobs, B->wl, B->wl_ctr, B->grids, B->tsorted, B->cell_cnt, B->cell_start, B->bm, B->pf, B->descs_init, B->bbox_init,
					   B->rjobs, B->ajobs, B->pair_rjob, B->order, B->icp_queue, B->icp_outs, B->trace_dev, B->steps};
		for (void *p : dev)
			if (p)
				(void)hipFree(p);
		if (B->states_h)
			(void)hipHostFree(B->states_h);
		if (B->outs_h)
			(void)hipHostFree(B->outs_h);
		if (B->bbox_h)
			(void)hipHostFree(B->bbox_h);
		if (B->epoch_h)
			(void)hipHostFree((void *)B->epoch_h);
		if (B->upload_h)
			(void)hipHostFree(B->upload_h);
		delete B;
	}

	int mulls_batch_create(mulls_ctx *ctx, const mulls_pair *pairs, int n, mulls_batch **out)
	try
	{
		if (!ctx || !pairs || n <= 0 || !out)
			return MULLS_E_INVALID;
		*out = nullptr;
		mulls_batch *B = new mulls_batch();
		const int rc = batch_fill(ctx, B, pairs, n);
		if (rc != MULLS_OK)
		{
			mulls_batch_destroy(ctx, B);
			return rc;
		}
		*out = B;
		return MULLS_OK;
	}
	catch (...)
	{
		return mulls::abi_caught(const_cast<mulls_ctx *>(ctx)); // nothing is thrown across the ABI
	}

	int mulls_batch_run(mulls_ctx *ctx, mulls_batch *B, const mulls_params *P, mulls_result *results)
	try
	{
		if (!ctx || !B || !results)
			return MULLS_E_INVALID;
		int rc = check_params(ctx, P);
		if (rc != MULLS_OK)
			return rc;
		HIPCHK(ctx, hipSetDevice(ctx->device));
		const auto wall0 = std::chrono::steady_clock::now();
		const int n = B->n;
		hipStream_t st = ctx->stream;
		ctx->prof = mulls_profile{};
		EvTimer evt{ctx};

		RunParams rp;
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
		if ((rc = take_epochs(ctx, B, (uint32_t)std::max(P->max_iter_num, 0) + 2u, &rp.tick_base)) != MULLS_OK)
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
		rc = prepare_run(ctx, B, P, rp, &lds_cap, &tier, &resident, dstep ? 1 : 0);
		if (rc != MULLS_OK)
			return rc;
		const bool use_grid = tier != 0;

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
		if (use_grid)
			launch_grid_build(st, (uint32_t)n, (uint32_t)B->tjobs_h.size(), B->tjobs, B->descs, B->grids, rp, B->tpos, B->bm, B->pf, B->cell_cnt, B->cell_start,
							  B->tsorted, tier == 2);
		evt.end();

		const mulls::IcpConst K = icp_const(P);
		// results of the loops that end on the device (k_icp; k_finish_step): IcpOut records -> mulls_result, profile counters
		auto results_from_device = [&](uint32_t trace_cap, bool from_icp) -> int {
			B->icp_outs_h.resize(n);
			HIPCHK(ctx, hipMemcpyAsync(B->icp_outs_h.data(), B->icp_outs, sizeof(IcpOut) * (size_t)n, hipMemcpyDeviceToHost, st));
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
				const IcpOut &o = B->icp_outs_h[p];
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
			return MULLS_OK;
		};
		if (resident)
		{
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
			const int rcr = results_from_device(trace_cap, true);
			if (rcr != MULLS_OK)
				return rcr;
			ctx->prof.launches_nn = 1;
			return MULLS_OK;
		}
		if (dstep)
		{
			// ---- lock-step loop with the O(1) half of the iteration on the device (k_reduce.hip: k_finish_step) ----------------------------
			// One launch set per iteration for the whole batch: search (+ filter), accumulation, finish + step.  The host keeps two sets queued
			// and reads one 8-byte word per set — (epoch << 32 | pairs still iterating) — to know when to stop queueing; a set queued behind the
			// last useful one finds no active pair and falls through.
			if (grow(ctx, &B->steps, &B->cap_steps, (size_t)n) != MULLS_OK || grow(ctx, &B->icp_outs, &B->cap_icp[3], (size_t)n) != MULLS_OK)
				return MULLS_E_HIP;
			volatile unsigned long long *word = reinterpret_cast<volatile unsigned long long *>(B->epoch_h + 32);
			unsigned long long *word_dev = reinterpret_cast<unsigned long long *>(B->epoch_dev + 32);
			HIPCHK(ctx, hipMemsetAsync(B->icp_outs, 0, sizeof(IcpOut) * (size_t)n, st));
			launch_step_init(st, (uint32_t)n, B->setup, K, B->steps, B->states);
			EvTimer ev2[2] = {EvTimer{ctx}, EvTimer{ctx}};
			ev2[1].base = 10;
			ev2[0].used = evt.used; // the setup events were recorded on the first set
			for (int k = 0; k < 5; k++)
				ev2[0].slot[k] = evt.slot[k];
			evt.used = 0;
			struct DrainOnError // an error from here on leaves kernels in flight that still write the pinned word
			{
				mulls_ctx *ctx;
				bool armed = true;
				~DrainOnError()
				{
					if (armed)
						(void)hipStreamSynchronize(ctx->stream);
				}
			} drain{ctx};
			const uint32_t epoch0 = B->epoch2;
			uint32_t left = (uint32_t)n, nn_launches = 0;
			// Small batches are bound by the NUMBER of launches (a kernel of a few hundred workgroups takes ~5 us whatever it does; one pair is bound by
			// the host's ~4 us per launch): the three accumulation launches become one, finish + step + publication one (k_finish_step)
			const bool few_launches = n <= (int)ctx->opt[MULLS_OPT_FEW_LAUNCHES_MAX_PAIRS];
			// wait until launch set `set` has published; left = pairs still iterating after the newest published set
			auto wait_set = [&](int set) -> int {
				const uint32_t want = epoch0 + (uint32_t)set + 1u;
				const auto t0 = std::chrono::steady_clock::now();
				bool synced = false;
				for (uint64_t spins = 0;; spins++)
				{
					const unsigned long long w = *word;
					if ((int32_t)((uint32_t)(w >> 32) - want) >= 0)
					{
						std::atomic_thread_fence(std::memory_order_acquire);
						left = (uint32_t)w;
						return MULLS_OK;
					}
					if (synced)
						break;
					if ((spins & 0xfff) == 0xfff && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0)
					{
						HIPCHK(ctx, hipStreamSynchronize(st)); // a stalled device, or an asynchronous error: surfaces here
						synced = true;
					}
				}
				ctx->err = "device did not publish the iteration epoch";
				return MULLS_E_HIP;
			};
			const auto t_loop0 = std::chrono::steady_clock::now();
			for (int s = 0; s <= P->max_iter_num; s++) // max_iter_num iterations and the residual pass of the last pairs to finish
			{
				if (s >= 2)
				{
					const auto t_wait0 = std::chrono::steady_clock::now();
					if ((rc = wait_set(s - 2)) != MULLS_OK)
						return rc;
					ctx->prof.ms_host_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wait0).count() * 1e3;
					ev2[s & 1].collect();
					if (left == 0)
						break;
				}
				EvTimer &ev = ev2[s & 1];
				const bool search = s < P->max_iter_num; // the last set can only hold residual passes
				if (search)
				{
					ev.begin(&ctx->prof.ms_nn);
					if (tier == 2)
					{
						if (launch_nn_lds(st, (uint32_t)B->cjobs_h.size(), B->cjobs, B->descs, B->states, rp, B->spos, B->snrm, B->grids, B->cell_start, B->tsorted, B->flag,
										  B->nn_idx, B->nn_d2, B->winner, B->tnrm, B->match, B->wd, B->tpos, B->nn_hint, B->mq, lds_cap, rp.grid_maxcells, B->wl, B->wl_ctr,
										  nn_launches++) != 0)
						{
							ctx->err = "could not raise the dynamic LDS limit of k_nn_lds";
							return MULLS_E_HIP;
						}
					}
					else if (tier == 1)
						launch_nn_grid(st, B->njobs, B->jobs, B->descs, B->states, rp, B->spos, B->snrm, B->grids, B->bm, B->pf, B->cell_start, B->tsorted, B->flag, B->nn_idx,
									   B->nn_d2, B->winner, B->tpos, B->nn_hint, B->match, B->mq);
					else
						launch_nn(st, B->njobs, B->jobs, B->descs, B->states, rp, B->spos, B->snrm, B->tpos, B->flag, B->nn_idx, B->nn_d2, B->winner);
					if (rp.normal_shooting)
						launch_nn_shoot(st, B->njobs, B->jobs, B->descs, B->states, rp, B->spos, B->snrm, B->tpos, B->flag, B->nn_idx, B->nn_d2, B->winner);
					ev.end();
					ev.begin(&ctx->prof.ms_filter);
					if (!rp.lds_dedup) // else k_nn_lds ran the rejection chain itself
						launch_filter(st, B->njobs, B->jobs, B->descs, B->states, rp, B->snrm, B->tnrm, B->flag, B->nn_idx, B->nn_d2, B->match, B->wd, B->winner, B->tpos, B->mq);
					ev.end();
					ctx->prof.launches_nn++;
				}
				// a set without a search holds only posterior-residual passes (every pair ran its last iteration in the set before): that is the
				// residual kernel time; a set of a converging batch mixes both kinds of pairs and is charged to the accumulation
				ev.begin(search ? &ctx->prof.ms_accum : &ctx->prof.ms_residual);
				for (int k = 0; k < B->nsub; k++)
					launch_accum(st, B->ajobs, B->ajob_split[k], B->jobs, B->descs, B->states, rp, B->spos, B->mq, B->flag, B->wd, B->partial, few_launches);
				launch_finish_step(st, (uint32_t)n, B->descs, B->states, rp, K, B->partial, B->outs, B->bbox, B->steps, B->icp_outs, word_dev, ++B->epoch2,
								   use_grid ? 0 : 1, few_launches ? B->ticket + 2 : nullptr);
				ev.end();
			}
			ctx->prof.ms_host_launch += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop0).count() * 1e3 - ctx->prof.ms_host_wait;
			HIPCHK(ctx, hipStreamSynchronize(st));
			drain.armed = false;
			ev2[0].collect();
			ev2[1].collect();
			return results_from_device(0, false);
		}
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
			uint32_t job_lo = 0, job_n = 0, cjob_lo = 0, cjob_n = 0;
			const uint32_t *ajob_split = nullptr;
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
			auto first_of = [](const std::vector<Job> &v, uint32_t pair) {
				return (uint32_t)(std::lower_bound(v.begin(), v.end(), pair, [](const Job &j, uint32_t q) { return j.pair < q; }) - v.begin());
			};
			S.job_lo = first_of(B->jobs_h, (uint32_t)S.lo);
			S.job_n = first_of(B->jobs_h, (uint32_t)S.hi) - S.job_lo;
			S.cjob_lo = first_of(B->cjobs_h, (uint32_t)S.lo);
			S.cjob_n = first_of(B->cjobs_h, (uint32_t)S.hi) - S.cjob_lo;
			S.ajob_split = B->ajob_split[k];
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
			const Job *jobs = B->jobs + S.job_lo;
			launch_push_states(st, B->states_pin + S.lo, B->states + S.lo, (uint32_t)(S.hi - S.lo));
			if (any_active)
			{
				ev.begin(&ctx->prof.ms_nn);
				if (tier == 2)
				{
					if (launch_nn_lds(st, S.cjob_n, B->cjobs + S.cjob_lo, B->descs, B->states, rp, B->spos, B->snrm, B->grids, B->cell_start, B->tsorted,
									  B->flag, B->nn_idx, B->nn_d2, B->winner, B->tnrm, B->match, B->wd, B->tpos, B->nn_hint, B->mq, lds_cap, rp.grid_maxcells,
									  B->wl + S.cjob_lo, B->wl_ctr + 8 * (int)(&S - subs), S.nn_launches++) != 0)
					{
						ctx->err = "could not raise the dynamic LDS limit of k_nn_lds";
						return MULLS_E_HIP;
					}
				}
				else if (tier == 1)
					launch_nn_grid(st, S.job_n, jobs, B->descs, B->states, rp, B->spos, B->snrm, B->grids, B->bm, B->pf, B->cell_start, B->tsorted, B->flag, B->nn_idx,
								   B->nn_d2, B->winner, B->tpos, B->nn_hint, B->match, B->mq);
				else
					launch_nn(st, S.job_n, jobs, B->descs, B->states, rp, B->spos, B->snrm, B->tpos, B->flag, B->nn_idx, B->nn_d2, B->winner);
				if (rp.normal_shooting)
					launch_nn_shoot(st, S.job_n, jobs, B->descs, B->states, rp, B->spos, B->snrm, B->tpos, B->flag, B->nn_idx, B->nn_d2, B->winner);
				ev.end();
				ev.begin(&ctx->prof.ms_filter);
				if (!rp.lds_dedup) // else k_nn_lds ran the rejection chain itself
					launch_filter(st, S.job_n, jobs, B->descs, B->states, rp, B->snrm, B->tnrm, B->flag, B->nn_idx, B->nn_d2, B->match, B->wd, B->winner, B->tpos, B->mq);
				ev.end();
				ctx->prof.launches_nn++;
				if (&S == &subs[0])
					ctx->prof.iterations++;
			}
			ev.begin(any_active ? &ctx->prof.ms_accum : &ctx->prof.ms_residual);
			launch_accum(st, B->ajobs, S.ajob_split, B->jobs, B->descs, B->states, rp, B->spos, B->mq, B->flag, B->wd, B->partial);
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
	catch (...)
	{
		return mulls::abi_caught(const_cast<mulls_ctx *>(ctx)); // nothing is thrown across the ABI
	}

	int mulls_icp_batch(mulls_ctx *ctx, const mulls_pair *pairs, int n, const mulls_params *params, mulls_result *results)
	try
	{
		if (!ctx)
			return MULLS_E_INVALID;
		int rc = check_params(ctx, params);
		if (rc != MULLS_OK)
			return rc;
		if (!pairs || n <= 0 || !results)
			return MULLS_E_INVALID;
		if (!ctx->scratch)
			ctx->scratch = new mulls_batch();
		rc = batch_fill(ctx, ctx->scratch, pairs, n, params);
		if (rc != MULLS_OK)
			return rc;
		rc = mulls_batch_run(ctx, ctx->scratch, params, results);
		ctx->prof.ms_stage = ctx->scratch->fill_ms, ctx->prof.ms_stage_pack = ctx->scratch->fill_pack_ms, ctx->prof.stage_bytes = ctx->scratch->fill_bytes;
		return rc;
	}
	catch (...)
	{
		return mulls::abi_caught(const_cast<mulls_ctx *>(ctx)); // nothing is thrown across the ABI
	}

	int mulls_icp(mulls_ctx *ctx, const mulls_pair *pair, const mulls_params *params, mulls_result *result)
	try
	{
		return mulls_icp_batch(ctx, pair, 1, params, result);
	}
	catch (...)
	{
		return mulls::abi_caught(const_cast<mulls_ctx *>(ctx)); // nothing is thrown across the ABI
	}


	// ------------------------------------------------------------------------------------------------------------
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
		if ((rc = take_epochs(ctx, B, (uint32_t)std::max(P->max_iter_num, 0) + 2u, &rp.tick_base)) != MULLS_OK)
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
		if (tier != 0)
			launch_grid_build(st, (uint32_t)n, (uint32_t)B->tjobs_h.size(), B->tjobs, B->descs, B->grids, rp, B->tpos, B->bm, B->pf, B->cell_cnt, B->cell_start,
							  B->tsorted, tier == 2);

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
				launch_nn_grid(st, B->njobs, B->jobs, B->descs, B->states, rp, B->spos, B->snrm, B->grids, B->bm, B->pf, B->cell_start, B->tsorted, B->flag,
							   B->nn_idx, B->nn_d2, B->winner, B->tpos, B->nn_hint, B->match, B->mq);
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
		if ((rc = take_epochs(ctx, B, 4u, &rp->tick_base)) != MULLS_OK)
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
		if (tier != 0)
			launch_grid_build(st, 1, (uint32_t)B->tjobs_h.size(), B->tjobs, B->descs, B->grids, *rp, B->tpos, B->bm, B->pf, B->cell_cnt, B->cell_start,
							  B->tsorted, tier == 2);
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
				launch_nn_grid(st, B->njobs, B->jobs, B->descs, B->states, rp, B->spos, B->snrm, B->grids, B->bm, B->pf, B->cell_start, B->tsorted, B->flag,
							   B->nn_idx, B->nn_d2, B->winner, B->tpos, B->nn_hint, B->match, B->mq);
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
