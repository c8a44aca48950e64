// batch.h — internals shared by the host driver's translation units: the device-resident batch (mulls_batch), its job tables and staging
// (batch.cpp), the iteration loops (loop.cpp), the variants (variants.cpp) and the stage-level entry points (stage.cpp).  Nothing here is ABI.
#pragma once
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

// Several small uploads and device-to-device copies as ONE launch (k_copy_segs) instead of one copy command each: a fill stages a pair's device-resident clouds
// and four tables, a run up to a dozen job tables — at 5-10 us a command that was 0.3 ms of a single scan-to-map registration's 1.2 ms.  Host data waits
// in the context's host-mapped mailbox until the kernel has read it: the caller synchronises the stream before the next SegCopier of the context is filled.
struct SegCopier
{
	mulls_ctx *ctx;
	std::vector<CopySeg> segs;
	struct Pending
	{
		void *dst;
		const void *src;
		size_t bytes;
	};
	std::vector<Pending> host; // copied into the mailbox by flush(), once the total is known
	explicit SegCopier(mulls_ctx *c) : ctx(c) {}
	void add_host(void *dst, const void *src, size_t bytes)
	{
		if (bytes)
			host.push_back({dst, src, bytes});
	}
	void add_dev(void *dst, const void *src, size_t bytes)
	{
		// (CopySeg carries a 32-bit size: a range of 4 GiB or more — a device-resident cloud of 89 M records — goes in as several segments, each a multiple of 16 bytes)
		const size_t piece = (size_t)0xfffffff0u;
		for (size_t off = 0; off < bytes; off += piece)
		{
			const size_t b = std::min(piece, bytes - off);
			segs.push_back({(unsigned long long)(uintptr_t)dst + off, (unsigned long long)(uintptr_t)src + off, (uint32_t)b, 0u});
		}
	}
	int flush(hipStream_t st); // batch.cpp
};

struct mulls_batch
{
	int n = 0;
	size_t n_src = 0, n_tgt = 0; // staged points over all pairs and classes
	std::vector<CloudDesc> descs_h;
	std::vector<PairSetup> setup_h;
	std::vector<Job> setup_jobs_h;
	std::vector<Job> big_segs_h, big_clouds_h; // class clouds cropped segment-wise (k_crop_big_*): segments, clouds (Job::cls carries MULLS_BIG_SRC_SIDE for a source cloud)
	uint32_t n_big_tgt = 0;					   // ... how many of those clouds are targets
	std::vector<Job> jobs_h;
	std::vector<Job> cjobs_h; // one entry per (pair, used class) with source points: the LDS tier's unit of work
	std::vector<Job> cjobs_dev_h; // the same entries as uploaded: inside each sub-batch's slice the most expensive class clouds come first
	std::vector<Job> bjobs_h; // global-memory tier (k_cert_big): class-level jobs (MULLS_JOB_CLASS) and 512-point chunk-level jobs of the MULLS_TIER_BM clouds, pair order
	std::vector<Job> fjobs_h; // ... the chunk-level ones among them: k_filter finishes those clouds
	std::vector<Job> ejobs_h; // ... and every MULLS_TIER_BM cloud as chunk-level jobs, for the first iterations of a mixed batch (empty when bjobs_h holds no class-level job):
							  // while most points still need a search, many (splittable) workgroups per cloud + k_filter beat one workgroup per cloud
	std::vector<uint32_t> lclouds_h; // the used class clouds on a bitmap grid (pair * MULLS_NC + class), in grid_slot order
	int tier_mode = -1;		  // what the descriptors' tier fields were assigned for: 0 / 1 / 2 one tier for the whole batch, 3 per class cloud (assign_tiers)
	std::vector<Job> tjobs_h; // target-side chunks (256 points) of the bitmap-grid clouds, for the grid build
	std::vector<uint32_t> ajobs_h; // jobs that start a trip of 1024 source slots: k_accum's workgroups (indices into jobs_h) — per sub-batch slice,
								   // and inside a slice grouped by trip length (ajob_split)
	uint32_t ajob_split[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}; // [sub-batch][0..3]: the slice's trips of > 512, 257..512, <= 256 slots
	// device-resident loop (k_icp): class-level jobs in pair order, each pair's range in them, the pairs most expensive first
	std::vector<Job> rjobs_h;
	std::vector<uint32_t> pair_rjob_h, order_h;
	IcpOut *icp_outs_pin = nullptr; // the result records of a device-stepped / device-resident run, downloaded into pinned memory (results_from_device)
	size_t cap_icp_pin = 0;
	std::vector<mulls_iter_trace> trace_h;
	std::string jobs_key;
	uint32_t njobs = 0;
	// device
	float4 *stage = nullptr;
	float4 *tmp_pos = nullptr, *tmp_nrm = nullptr;
	float4 *spos = nullptr, *snrm = nullptr, *tpos = nullptr, *tnrm = nullptr;
	uint8_t *flag = nullptr;
	int32_t *match = nullptr, *nn_idx = nullptr, *nn_hint = nullptr;
	uint4 *nn_cand = nullptr; // per source point: candidate record of the k-candidate certificates (RunParams::cand)
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
	Job *bjobs = nullptr, *fjobs = nullptr, *ejobs = nullptr;
	uint32_t *lclouds = nullptr;
	size_t cap_bjobs[4] = {};
	uint32_t *bm_cs = nullptr; // bitmap grids: first sorted position of every occupied cell (indexed like cell_cnt)
	size_t cap_bm_cs = 0;
	uint32_t *bm_rank = nullptr; // bitmap grids: every target point's (cell counter index, arrival number in its cell), from k_bm_count to k_bm_scatter — the scatter is
								 // pure data movement: one pass of atomics per build instead of two
	size_t cap_bm_rank = 0;
	Job *rjobs = nullptr;
	uint32_t *ajobs = nullptr;
	size_t cap_ajobs = 0;
	uint32_t *pair_rjob = nullptr, *order = nullptr, *icp_queue = nullptr;
	IcpOut *icp_outs = nullptr;
	mulls_iter_trace *trace_dev = nullptr;
	size_t cap_icp[5] = {};
	mulls::StepState *steps = nullptr; // lock-step loop with the device step: per-pair loop state
	size_t cap_steps = 0;
	uint32_t epoch2 = 0, epoch3 = 0; // ... and the last epochs issued on its 8-byte words (words 32-33 / 48-49 of epoch_h: one per sub-batch)
	int nsub = 1;		 // sub-batches the job tables are laid out for (build_jobs)
	double fill_ms = 0.0, fill_pack_ms = 0.0; // the last batch_fill: wall time, host packing time ...
	uint64_t fill_bytes = 0;				   // ... and bytes staged
	uint32_t *wl = nullptr;		// LDS tier: class clouds k_cert queued for k_nn_lds (one slot per class-level job)
	uint32_t *wl_ctr = nullptr; // ... and the queue counters: per sub-batch 8 words = (queued, taken) x launch parity
	size_t cap_wl = 0;
	GridDesc *grids = nullptr;
	float4 *tsorted = nullptr;
	unsigned long long *dbg = nullptr; // diagnostics (MULLS_OPT_DEBUG_STOP = 20): RunParams::dbg_ticks
	uint16_t *tmap = nullptr; // LDS tier without a cropped copy of the target clouds (k_tgt_grid): rank in the cropped cloud -> staged index
	uint32_t *cell_cnt = nullptr, *cell_start = nullptr; // bitmap grids: per-occupied-cell counters (start positions: bm_cs); LDS tier: dense cell tables
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
	size_t cap_stage = 0, cap_src[13] = {}, cap_tgt[5] = {}, cap_pairs[5] = {}, cap_setup_jobs = 0, cap_pin[4] = {};
};

namespace mulls_drv
{
using mulls::Mat4;
using mulls::Mat6;

inline float ord_to_float(uint32_t k)
{
	const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
	float f;
	std::memcpy(&f, &u, sizeof(f));
	return f;
}
inline int metric_of(int c) { return (c == MULLS_PILLAR || c == MULLS_BEAM) ? 1 : (c == MULLS_VERTEX ? 2 : 0); }
// packed index of (r,c), r <= c, in the row-major-upper enumeration used by k_accum
inline int packed(int r, int c) { return r * 6 - r * (r - 1) / 2 + (c - r); }
void rows12(const double colmajor[16], double out[12]);

// host-side life of one pair during a run: the shared per-iteration state (icp_step.h) + host-only bookkeeping
struct PairHost : mulls::PairIter
{
	bool first = true;
	uint32_t alive_prev[MULLS_NC];
};

void fill_crop_box(const RunParams &rp, const double tgt_bound[6], const uint32_t keys[6], mulls_result &R);
int check_params(mulls_ctx *ctx, const mulls_params *P);
void init_cert(const mulls_ctx *ctx, RunParams &rp);
int subbatch_count(const mulls_ctx *ctx, int n);
void options_init(mulls_ctx *ctx);
bool option_value_ok(int option, double *value);
uint32_t lds_dedup_max_pts();
void assign_tiers(mulls_batch *B, const uint8_t used[MULLS_NC], int mode);
void build_jobs(mulls_batch *B, const mulls_params *P, int nsub, int mode);
// `last`: while profiling, the event recorded behind the k_finish that publishes `want` — waiting on it (instead of the whole stream) leaves the
// other sub-batch's kernels running
int wait_epoch_word(mulls_ctx *ctx, volatile uint32_t *word, uint32_t want, hipEvent_t last = nullptr, hipStream_t stream = nullptr);
inline int wait_epoch(mulls_ctx *ctx, mulls_batch *B) { return wait_epoch_word(ctx, B->epoch_h, B->epoch); }
void unpack_out(const mulls_batch *B, const uint8_t used[MULLS_NC], int p, PairOut &o, bool comb = false);
uint32_t lds_cells_for(uint32_t cap);
// P_mixed: the caller can run a batch whose class clouds are on different tiers (mulls_batch_run): auto mode may answer 3
int choose_tier(const mulls_ctx *ctx, const mulls_batch *B, const uint8_t used[MULLS_NC], uint32_t *lds_cap, const mulls_params *P_mixed = nullptr);
int batch_fill(mulls_ctx *ctx, mulls_batch *B, const mulls_pair *pairs, int n, const mulls_params *P = nullptr);
// nsub: sub-batches the lock-step job tables are laid out for (0 = subbatch_count)
int prepare_run(mulls_ctx *ctx, mulls_batch *B, const mulls_params *P_jobs, RunParams &rp, uint32_t *lds_cap_out, int *tier_out, bool *resident_out = nullptr, int nsub = 0,
				bool allow_mixed = false);
mulls::IcpConst icp_const(const mulls_params *P);
int take_epochs(mulls_ctx *ctx, mulls_batch *B, uint32_t n, RunParams &rp);

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
} // namespace mulls_drv
