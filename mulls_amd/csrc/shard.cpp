// shard.cpp — independent scan pairs over several contexts, inside one process ("one host thread + one HIP stream set per GPU"):
//   mulls_icp_batch_sharded   a pair list block-partitioned over the caller's contexts (one per GPU of the node, or several on one GPU), one host thread each,
//                             results written in place — no collective: the shards share nothing once their inputs are marshalled
//   mulls_pipe_*              calls from host buffers in flight on alternating contexts of one device, so that call k + 1's host gather and PCIe upload run under
//                             call k's kernels (a serial caller waits 9 ms of staging in front of 3.7 ms of kernels per 1024 pairs, profiles/r04_e2e_calls.txt)
// Both sit on mulls_icp_batch: a context is driven by one host thread at a time, contexts are independent (own streams, staging arenas, thread pool, profile,
// error text), and what the launch wrappers keep per device is set up under a lock (launch.h: DevLaunch).
#include "ctx.h"

#include <deque>
#include <memory>

namespace
{
// pair p -> shard floor(p * n_shards / n): contiguous blocks whose sizes differ by at most one (mulls_amd/shard.py::block_partition)
inline void block_of(int n, int n_shards, int r, int *lo, int *hi)
{
	*lo = (int)(((long long)n * r) / n_shards);
	*hi = (int)(((long long)n * (r + 1)) / n_shards);
}
int run_shard(mulls_ctx *ctx, const mulls_pair *pairs, int lo, int hi, const mulls_params *params, mulls_result *results) noexcept
{
	if (hi <= lo)
		return MULLS_OK;
	try
	{
		return mulls_icp_batch(ctx, pairs + lo, hi - lo, params, results + lo);
	}
	catch (...)
	{
		return mulls::abi_caught(ctx);
	}
}
} // namespace

// One call in flight on one lane of a pipe
struct PipeJob
{
	const mulls_pair *pairs = nullptr;
	int n = 0;
	mulls_params params{};
	mulls_result *results = nullptr;
	int ticket = -1, rc = MULLS_OK;
	bool done = false;
};
struct PipeLane
{
	mulls_ctx *ctx = nullptr;
	std::thread th;
	std::mutex mu;
	std::condition_variable cv;
	std::deque<std::shared_ptr<PipeJob>> queue; // submitted, not yet run (at most one: a lane takes its next call when the previous one was waited for)
	std::shared_ptr<PipeJob> last;				// the lane's latest job (running or done)
	bool stop = false;
};
struct mulls_pipe
{
	int device = 0;
	std::vector<std::unique_ptr<PipeLane>> lanes;
	std::mutex mu; // tickets
	int next_ticket = 0;
	std::string err;
};

namespace
{
void lane_main(PipeLane *L)
{
	for (;;)
	{
		std::shared_ptr<PipeJob> job;
		{
			std::unique_lock<std::mutex> lk(L->mu);
			L->cv.wait(lk, [&] { return L->stop || !L->queue.empty(); });
			if (L->queue.empty())
				return; // (stop, and nothing left to run)
			job = L->queue.front();
			L->queue.pop_front();
		}
		const int rc = run_shard(L->ctx, job->pairs, 0, job->n, &job->params, job->results);
		{
			std::lock_guard<std::mutex> lk(L->mu);
			job->rc = rc;
			job->done = true;
		}
		L->cv.notify_all();
	}
}
} // namespace

extern "C"
{
	int mulls_icp_batch_sharded(mulls_ctx *const *ctxs, int n_ctx, const mulls_pair *pairs, int n, const mulls_params *params, mulls_result *results)
	try
	{
		if (!ctxs || n_ctx <= 0 || !pairs || n <= 0 || !params || !results)
			return MULLS_E_INVALID;
		for (int r = 0; r < n_ctx; r++)
		{
			if (!ctxs[r])
				return MULLS_E_INVALID;
			for (int q = 0; q < r; q++)
				if (ctxs[q] == ctxs[r])
					return MULLS_E_INVALID; // a context is driven by one host thread at a time
		}
		std::vector<int> rcs((size_t)n_ctx, MULLS_OK);
		std::vector<std::thread> th;
		th.reserve((size_t)n_ctx);
		for (int r = 1; r < n_ctx; r++)
		{
			int lo, hi;
			block_of(n, n_ctx, r, &lo, &hi);
			try
			{
				th.emplace_back([=, &rcs] { rcs[(size_t)r] = run_shard(ctxs[r], pairs, lo, hi, params, results); });
			}
			catch (...) // no thread to be had: the shards already running are waited for (a joinable std::thread must not be destroyed), nothing leaves the ABI
			{
				for (auto &t : th)
					t.join();
				return MULLS_E_NOMEM;
			}
		}
		int lo0, hi0;
		block_of(n, n_ctx, 0, &lo0, &hi0);
		rcs[0] = run_shard(ctxs[0], pairs, lo0, hi0, params, results); // the caller's thread drives the first shard
		for (auto &t : th)
			t.join();
		for (int r = 0; r < n_ctx; r++)
			if (rcs[(size_t)r] != MULLS_OK)
				return rcs[(size_t)r]; // (mulls_last_error(ctxs[r]) has the text)
		return MULLS_OK;
	}
	catch (...)
	{
		return mulls::abi_caught(nullptr); // nothing is thrown across the ABI
	}

	void mulls_pack_results(const mulls_result *results, int n, double *table)
	{
		if (!results || !table)
			return;
		for (int i = 0; i < n; i++)
		{
			const mulls_result &r = results[i];
			double *row = table + (size_t)i * 56;
			std::memcpy(row, r.T, sizeof(double) * 16);
			std::memcpy(row + 16, r.info, sizeof(double) * 36);
			row[52] = (double)r.code, row[53] = (double)r.iters, row[54] = (double)r.sigma, row[55] = (double)r.confidence;
		}
	}

	int mulls_pipe_create(int device, int depth, mulls_pipe **out)
	try
	{
		if (!out || depth < 1 || depth > 8)
			return MULLS_E_INVALID;
		*out = nullptr;
		std::unique_ptr<mulls_pipe> P(new mulls_pipe());
		P->device = device;
		for (int k = 0; k < depth; k++)
		{
			std::unique_ptr<PipeLane> L(new PipeLane());
			const int rc = mulls_create(device, &L->ctx);
			if (rc != MULLS_OK)
			{
				for (auto &l : P->lanes)
					mulls_destroy(l->ctx);
				return rc;
			}
			P->lanes.push_back(std::move(L));
		}
		try
		{
			for (auto &l : P->lanes)
				l->th = std::thread(lane_main, l.get());
		}
		catch (...) // no thread to be had: the lanes already started are stopped and joined, every context destroyed
		{
			mulls_pipe_destroy(P.release());
			return MULLS_E_NOMEM;
		}
		*out = P.release();
		return MULLS_OK;
	}
	catch (...)
	{
		return mulls::abi_caught(nullptr);
	}

	void mulls_pipe_destroy(mulls_pipe *P)
	{
		if (!P)
			return;
		for (auto &l : P->lanes)
		{
			{
				std::lock_guard<std::mutex> lk(l->mu);
				l->stop = true;
			}
			l->cv.notify_all();
		}
		for (auto &l : P->lanes)
		{
			if (l->th.joinable())
				l->th.join(); // (a call still in flight finishes first: its buffers belong to the caller)
			mulls_destroy(l->ctx);
		}
		delete P;
	}

	int mulls_pipe_depth(const mulls_pipe *P) { return P ? (int)P->lanes.size() : 0; }

	mulls_ctx *mulls_pipe_ctx(mulls_pipe *P, int lane) { return (P && lane >= 0 && lane < (int)P->lanes.size()) ? P->lanes[(size_t)lane]->ctx : nullptr; }

	int mulls_pipe_set_option(mulls_pipe *P, int option, double value)
	{
		if (!P)
			return MULLS_E_INVALID;
		for (auto &l : P->lanes)
		{
			std::unique_lock<std::mutex> lk(l->mu);
			l->cv.wait(lk, [&] { return l->queue.empty() && (!l->last || l->last->done); }); // (options are read by a running call)
			const int rc = mulls_set_option(l->ctx, option, value);
			if (rc != MULLS_OK)
				return rc;
		}
		return MULLS_OK;
	}

	int mulls_icp_batch_begin(mulls_pipe *P, const mulls_pair *pairs, int n, const mulls_params *params, mulls_result *results)
	try
	{
		if (!P || !pairs || n <= 0 || !params || !results)
			return MULLS_E_INVALID;
		int ticket;
		{
			std::lock_guard<std::mutex> lk(P->mu);
			ticket = P->next_ticket++;
			if (P->next_ticket < 0)
				P->next_ticket = 0;
		}
		PipeLane *L = P->lanes[(size_t)ticket % P->lanes.size()].get();
		std::shared_ptr<PipeJob> job(new PipeJob());
		job->pairs = pairs, job->n = n, job->params = *params, job->results = results, job->ticket = ticket;
		{
			std::unique_lock<std::mutex> lk(L->mu);
			L->cv.wait(lk, [&] { return L->queue.empty() && (!L->last || L->last->done); }); // the lane's previous call has finished (its results are the caller's)
			L->last = job;
			L->queue.push_back(job);
		}
		L->cv.notify_all();
		return ticket;
	}
	catch (...)
	{
		return mulls::abi_caught(nullptr);
	}

	int mulls_icp_batch_end(mulls_pipe *P, int ticket)
	try
	{
		if (!P || ticket < 0)
			return MULLS_E_INVALID;
		PipeLane *L = P->lanes[(size_t)ticket % P->lanes.size()].get();
		std::unique_lock<std::mutex> lk(L->mu);
		if (!L->last || L->last->ticket != ticket)
			return MULLS_E_INVALID; // not this lane's latest call: already waited for and replaced, or never begun
		std::shared_ptr<PipeJob> job = L->last;
		L->cv.wait(lk, [&] { return job->done; });
		return job->rc;
	}
	catch (...)
	{
		return mulls::abi_caught(nullptr);
	}
}
