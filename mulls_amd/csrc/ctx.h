// ctx.h — pieces of the host driver shared by driver.cpp (registration) and map.cpp (device-resident local map)
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "../../include/mulls_hip.h"

struct mulls_batch;
struct mulls_map;
struct mulls_block;

// A few host threads that SLEEP between their jobs (condition variable), for the staging gather of mulls_icp_batch.  An OpenMP team keeps spinning after its
// parallel region; 32 spinning threads inside a container with a CPU quota get the whole process throttled for the rest of the scheduler period: every fourth
// call of a 1024-pair mulls_icp_batch took 63 ms instead of 12 (profiles/r04_e2e_calls.txt; OMP_WAIT_POLICY=passive showed the same cure).
class HostPool
{
  public:
	explicit HostPool(int n_threads);
	~HostPool();
	// fn(i) for every i in [begin, end), `grain` indices at a time, on the pool's threads and the caller's; returns when all are done
	void parallel_for(long begin, long end, long grain, const std::function<void(long)> &fn);
	int size() const { return (int)th_.size() + 1; }

  private:
	void worker();
	void drain(const std::function<void(long)> &fn, long end, long grain);
	std::vector<std::thread> th_;
	std::mutex mu_, call_mu_;
	std::exception_ptr error_;
	std::condition_variable cv_, done_;
	const std::function<void(long)> *fn_ = nullptr;
	std::atomic<long> next_{0};
	long end_ = 0, grain_ = 1;
	unsigned long gen_ = 0;
	int busy_ = 0;
	bool stop_ = false;
};

int usable_cpus();			   // affinity mask cut by the container's CPU quota
HostPool &shared_host_pool(); // the process's one pool

struct mulls_ctx
{
	int device = 0;
	hipStream_t stream = nullptr;
	hipStream_t stream2 = nullptr; // second sub-batch of a large batch iterates here, so that its latency-bound filter / accumulate
								   // kernels overlap the issue-bound search of the other sub-batch
	hipEvent_t ev_setup = nullptr; // setup done on `stream` -> stream2 may start
	std::string err;
	int profiling = 0; // 0 off, 1 hipEvent pairs around every launch group, 2 around the correspondence search only
	mulls_profile prof{};
	hipEvent_t ev[20] = {}; // two sets of ten: one per sub-batch in flight
	mulls_batch *scratch = nullptr; // cached batch reused by mulls_icp / mulls_icp_batch (no allocator traffic per call)
	std::vector<mulls_map *> maps; // live local maps: their buffers are the device clouds mulls_pair may point to
	std::vector<mulls_block *> blocks; // live feature blocks (mulls_extract_features_resident): device clouds too
	void *gf_buf = nullptr; // mulls_ground_filter's device arena (grow-only)
	void *gf_rnd = nullptr; // ... and PCL's RANSAC sample sequence (normal method 3), made on first use
	void *cl_buf = nullptr; // mulls_classify_nground's device arena (grow-only)
	size_t cl_cap = 0;
	size_t gf_cap = 0;
	double opt[MULLS_OPT_COUNT] = {}; // enum mulls_option (mulls_set_option; preset from the environment by mulls_create)
	unsigned char *mail_h = nullptr; // host-mapped mailbox of the small-table uploads (SegCopier, batch.h): written by the host, read by k_copy_segs
	size_t mail_cap = 0;
	unsigned char *cl_pin = nullptr, *scan_pin = nullptr; // pinned host scratch of the feature stage: the visiting orders' keys and permutations; the raw scan on its way up
	size_t cl_pin_cap = 0, scan_pin_cap = 0;
	uint32_t cl_rounds_hint[2] = {4, 8}; // rounds the promotion loop / the suppression rounds needed in the previous call: this call's first batch
	int nn_mode = 0;   // 0 auto, 1 LDS-tiled brute force, 2 uniform grid in global memory, 3 uniform grid staged in LDS
};

namespace mulls
{
// Handler of every extern "C" entry point's function-try-block: nothing is thrown across the ABI (include/mulls_hip.h).
// A failed host allocation (std::vector growth on an absurd size) becomes MULLS_E_NOMEM, anything else MULLS_E_INVALID.
inline int abi_caught(mulls_ctx *ctx) noexcept
{
	int rc = MULLS_E_INVALID;
	const char *what = "unknown C++ exception";
	try
	{
		throw;
	}
	catch (const std::bad_alloc &)
	{
		rc = MULLS_E_NOMEM;
		what = "host allocation failed";
	}
	catch (const std::exception &e)
	{
		what = e.what();
	}
	catch (...)
	{
	}
	if (ctx)
		try
		{
			ctx->err = what;
		}
		catch (...)
		{
		}
	return rc;
}
// Scope guard of an entry point that queues asynchronous copies into caller buffers or into host vectors of its own: whichever way
// the function is left — an error return included — the stream is idle before those buffers go back to their owners.
struct StreamDrain
{
	hipStream_t st;
	~StreamDrain() { (void)hipStreamSynchronize(st); }
};
} // namespace mulls

// registry of device arrays that start inside their allocation (driver.cpp)
void *staggered_base(void *p, bool forget);
void staggered_note(void *p, void *base);

namespace
{

#define HIPCHK(ctx, call)                                                                                                   \
	do                                                                                                                      \
	{                                                                                                                       \
		hipError_t e_ = (call);                                                                                             \
		if (e_ != hipSuccess)                                                                                               \
		{                                                                                                                   \
			(ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                                 \
			return MULLS_E_HIP;                                                                                             \
		}                                                                                                                   \
	} while (0)

template <typename T>
int dmalloc(mulls_ctx *ctx, T **p, size_t count)
{
	*p = nullptr;
	HIPCHK(ctx, hipMalloc((void **)p, std::max<size_t>(count, 1) * sizeof(T)));
	return MULLS_OK;
}

// seeded order-preserving selection sampling shared with the oracle's definition (include/mulls_hip.h: rng_seed)
inline uint64_t splitmix64(uint64_t &x)
{
	x += 0x9E3779B97F4A7C15ull;
	uint64_t z = x;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
// CFilter::random_downsample_pcl semantics (cfilter.hpp:606-628) expressed as a keep mask; returns the new size
uint32_t thin_mask(uint8_t *mask, uint32_t n, int keep_number, uint64_t seed, int cloud_id)
{
	if (keep_number < 0 || (long)n <= (long)keep_number) // upstream compares `points.size() <= keep_number` as size_t: a negative count keeps every point
	{
		std::memset(mask, 1, n);
		return n;
	}
	std::memset(mask, 0, n);
	if (keep_number == 0)
		return 0;
	uint64_t state = seed ^ (0x100000001B3ull * (uint64_t)(cloud_id + 1));
	uint32_t need = (uint32_t)keep_number;
	for (uint32_t i = 0; i < n && need > 0; i++)
	{
		const double u = (double)(splitmix64(state) >> 11) * (1.0 / 9007199254740992.0);
		if (u * (double)(n - i) < (double)need)
		{
			mask[i] = 1;
			need--;
		}
	}
	return (uint32_t)keep_number;
}

// grow-only device / pinned arrays: a batch object can be refilled with new pairs without touching the allocator when
// the previous capacity suffices (mulls_icp / mulls_icp_batch reuse one cached batch per context)
// Device arrays may start `stagger` bytes into their allocation (the per-point arrays of a batch: every large hipMalloc is 2 MiB-aligned, so the same index
// of eight arrays is the same offset into eight 2 MiB pages); staggered_free() finds the allocation again.
inline void staggered_free(void *p)
{
	if (p)
		(void)hipFree(staggered_base(p, true));
}
template <typename T>
int grow(mulls_ctx *ctx, T **p, size_t *cap, size_t need, bool *grew = nullptr, size_t stagger = 0)
{
	if (grew)
		*grew = false;
	if (*p && *cap >= need)
		return MULLS_OK;
	staggered_free(*p);
	*p = nullptr;
	const size_t want = std::max<size_t>(need + need / 4, 64);
	void *base = nullptr;
	HIPCHK(ctx, hipMalloc(&base, want * sizeof(T) + stagger));
	*p = reinterpret_cast<T *>(static_cast<unsigned char *>(base) + stagger);
	if (stagger)
		staggered_note(*p, base);
	*cap = want;
	if (grew)
		*grew = true;
	return MULLS_OK;
}
template <typename T>
int grow_pinned(mulls_ctx *ctx, T **p, size_t *cap, size_t need, unsigned flags)
{
	if (*p && *cap >= need)
		return MULLS_OK;
	if (*p)
		(void)hipHostFree((void *)*p);
	*p = nullptr;
	const size_t want = std::max<size_t>(need + need / 4, 64);
	HIPCHK(ctx, hipHostMalloc((void **)p, want * sizeof(T), flags));
	*cap = want;
	return MULLS_OK;
}

} // namespace

// is [p, p + bytes) inside a live local map's or feature block's device buffers? (mulls_cloud.pts of a device-resident cloud; map.cpp)
bool mulls_is_map_memory(const mulls_ctx *ctx, const void *p, size_t bytes);

// device-resident feature clouds of one scan (mulls_extract_features_resident, ground.cpp): 48-byte records, one buffer
struct mulls_block
{
	unsigned char *buf = nullptr;
	size_t cap = 0;					 // bytes
	size_t off[MULLS_EX_COUNT] = {}; // where cloud k starts
	uint32_t n[MULLS_EX_COUNT] = {}; // its size (RAW and DOWN are not kept: 0)
};

// classify_nground_pts' clouds where the kernels left them (mulls_classify_impl with a ClassifyDev instead of host buffers): valid until the
// context's next classification; the *_down clouds the fixed-number samplers thin on the host come back as host records
struct ClassifyDev
{
	const void *dev[MULLS_CL_COUNT] = {};
	uint32_t n[MULLS_CL_COUNT] = {};
	std::vector<unsigned char> host[MULLS_CL_COUNT]; // non-empty: this cloud's records are here instead
	bool on_host[MULLS_CL_COUNT] = {};
	const void *cloud_in_after = nullptr; // cloud_in as upstream leaves it (device)
	uint32_t n_after = 0;
};
