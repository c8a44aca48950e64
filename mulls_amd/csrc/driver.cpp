// driver.cpp — the C ABI of include/mulls_hip.h: contexts, options, profiles, batches, mulls_icp / mulls_icp_batch
// (reference: CRegistration::mm_lls_icp, include/common/cregistration.hpp:1114-1440).
//
// Division of labour (BASELINE.json north_star): correspondences, rejection, the normal-equation reduction AND the per-iteration 6x6 solve with
// its step / convergence / health tests run in the HIP kernels of k_setup / k_grid / k_search / k_reduce / k_icp .hip; the host queues launch
// sets (loop.cpp), reads one 8-byte word per set to learn how many pairs still iterate, and downloads the result records at the end.  Only when
// the caller asks for per-iteration traces does the host step the loop itself.  There is no CPU fallback anywhere in the driver: without a usable
// HIP device every entry point fails with MULLS_E_NO_DEVICE / MULLS_E_HIP.
//   batch.cpp    options, job tables, staging (batch_fill), per-run tables and tier choice (prepare_run)
//   loop.cpp     mulls_batch_run: set-up launches and the three loops
//   variants.cpp lls_icp_3dof_ground, mm_lls_icp_4dof_global
//   stage.cpp    mulls_stage_*
#include "batch.h"

#include <sched.h>
#include <cstdio>
#include <cstdlib>

#include <mutex>
#include <unordered_map>
namespace
{
std::mutex g_stagger_mu;
std::unordered_map<void *, void *> g_stagger;
} // namespace
void staggered_note(void *p, void *base)
{
	std::lock_guard<std::mutex> lk(g_stagger_mu);
	g_stagger[p] = base;
}
void *staggered_base(void *p, bool forget)
{
	std::lock_guard<std::mutex> lk(g_stagger_mu);
	auto it = g_stagger.find(p);
	if (it == g_stagger.end())
		return p;
	void *b = it->second;
	if (forget)
		g_stagger.erase(it);
	return b;
}

// CPUs this process may actually use: the affinity mask, cut by the container's CPU quota (cgroup v2 cpu.max, cgroup v1 cfs quota / period).  A pool sized from
// hardware_concurrency() alone — 32 threads on a 256-CPU host whose container holds a 16-CPU quota — gets the whole process throttled for the rest of every
// scheduler period as soon as two contexts gather at once (profiles/r05_pipe_calls.txt: 83 ms of "staging" per call with three contexts).
int usable_cpus()
{
	int n = (int)std::max(1u, std::thread::hardware_concurrency());
	cpu_set_t set;
	CPU_ZERO(&set);
	if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0)
		n = std::min(n, CPU_COUNT(&set));
	auto quota = [&](const char *path_quota, const char *path_period) {
		FILE *f = std::fopen(path_quota, "r");
		if (!f)
			return;
		char a[64] = {0}, b[64] = {0};
		const int got = std::fscanf(f, "%63s %63s", a, b);
		std::fclose(f);
		double q = -1.0, per = 100000.0;
		if (got >= 1 && std::strcmp(a, "max") != 0)
			q = std::atof(a);
		if (got >= 2)
			per = std::atof(b);
		else if (path_period)
		{
			if (FILE *g = std::fopen(path_period, "r"))
			{
				if (std::fscanf(g, "%63s", b) == 1)
					per = std::atof(b);
				std::fclose(g);
			}
		}
		if (q > 0.0 && per > 0.0)
			n = std::min(n, std::max(1, (int)(q / per)));
	};
	quota("/sys/fs/cgroup/cpu.max", nullptr);
	quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
	return std::max(1, n);
}

// The process's one pool (every context's staging gather and host sorts go through it, one parallel_for at a time: the CPUs are one resource however many contexts
// and pipes the process holds): up to 32 threads, and two of the usable CPUs left to the threads that drive the launch loops.
HostPool &shared_host_pool()
{
	static HostPool pool(std::max(1, std::min(32, usable_cpus() - 2)));
	return pool;
}

HostPool::HostPool(int n_threads)
{
	for (int k = 1; k < n_threads; k++)
		th_.emplace_back([this] { worker(); });
}
HostPool::~HostPool()
{
	{
		std::lock_guard<std::mutex> lk(mu_);
		stop_ = true;
	}
	cv_.notify_all();
	for (std::thread &t : th_)
		t.join();
}
// the share of one thread (a worker or the caller); an exception ends the loop for everybody and is kept for the caller to rethrow
void HostPool::drain(const std::function<void(long)> &fn, long end, long grain)
{
	try
	{
		for (long i = next_.fetch_add(grain); i < end; i = next_.fetch_add(grain))
			for (long k = i; k < std::min(end, i + grain); k++)
				fn(k);
	}
	catch (...)
	{
		next_.store(end); // nobody takes another index
		std::lock_guard<std::mutex> lk(mu_);
		if (!error_)
			error_ = std::current_exception();
	}
}
void HostPool::worker()
{
	unsigned long seen = 0;
	for (;;)
	{
		const std::function<void(long)> *fn;
		long end, grain;
		{
			std::unique_lock<std::mutex> lk(mu_);
			cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
			if (stop_)
				return;
			seen = gen_;
			fn = fn_, end = end_, grain = grain_;
		}
		drain(*fn, end, grain);
		{
			std::lock_guard<std::mutex> lk(mu_);
			if (--busy_ == 0)
				done_.notify_one();
		}
	}
}
void HostPool::parallel_for(long begin, long end, long grain, const std::function<void(long)> &fn)
{
	if (end <= begin)
		return;
	grain = std::max<long>(grain, 1);
	if (th_.empty() || end - begin <= grain)
	{
		for (long k = begin; k < end; k++)
			fn(k);
		return;
	}
	std::lock_guard<std::mutex> one_call(call_mu_); // callers of different contexts take turns
	{
		std::lock_guard<std::mutex> lk(mu_);
		fn_ = &fn, end_ = end, grain_ = grain;
		next_.store(begin);
		busy_ = (int)th_.size();
		error_ = nullptr;
		gen_++;
	}
	cv_.notify_all();
	drain(fn, end, grain);
	std::exception_ptr err;
	{
		std::unique_lock<std::mutex> lk(mu_);
		done_.wait(lk, [&] { return busy_ == 0; }); // (every worker has left fn before it goes out of scope, also after an exception)
		err = error_;
		error_ = nullptr;
	}
	if (err)
		std::rethrow_exception(err); // reaches the ABI's function-try-block like an exception of the calling thread
}

using namespace mulls_drv;

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
		// the profile's timing events: no system-scope fence when they complete (hipEventDisableSystemFence — "for events that are only being used to measure timing").
		// With the default flags every recorded event wrote the L2 back and invalidated it: 10 us in front of an iteration's search and 19 us behind it, 0.58 ms of a
		// 4096-pair step whose search launches bench.py brackets (profiles/r06_experiments.txt item 33).  Nothing the host reads is ordered by these events: results and
		// the iteration word are read behind hipStreamSynchronize or the kernels' own __threadfence_system.
		for (auto &e : ctx->ev)
			if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess)
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
		if (ctx->mail_h)
			(void)hipHostFree(ctx->mail_h);
		if (ctx->cl_pin)
			(void)hipHostFree(ctx->cl_pin);
		if (ctx->scan_pin)
			(void)hipHostFree(ctx->scan_pin);
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
		if (!ctx || option < 0 || option >= MULLS_OPT_COUNT)
			return MULLS_E_INVALID;
		if (!option_value_ok(option, &value))
		{
			ctx->err = "mulls_set_option: value out of the option's range";
			return MULLS_E_INVALID;
		}
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
		void *dev[] = {B->stage, B->tmp_pos, B->tmp_nrm, B->spos, B->snrm, B->tpos, B->tnrm, B->flag, B->match, B->nn_idx, B->nn_hint, B->nn_cand, B->mq, B->wd,
					   B->nn_d2, B->winner, B->descs, B->setup, B->states, B->outs, B->ticket, B->bbox, B->setup_jobs, B->big_segs, B->big_clouds, B->seg_cnt, B->big_box, B->jobs, B->partial,
					   B->tjobs, B->cjobs, B->bjobs, B->fjobs, B->ejobs, B->lclouds, B->bm_cs, B->wl, B->wl_ctr, B->grids, B->tsorted, B->tmap, B->dbg, B->cell_cnt, B->cell_start, B->bm, B->pf, B->descs_init, B->bbox_init,
					   B->rjobs, B->ajobs, B->pair_rjob, B->order, B->icp_queue, B->icp_outs, B->trace_dev, B->steps, B->bm_rank};
		for (void *p : dev)
			if (p)
				staggered_free(p);
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
		if (B->icp_outs_pin)
			(void)hipHostFree(B->icp_outs_pin);
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

}
