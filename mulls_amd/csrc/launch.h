// launch.h — host-callable wrappers around the kernels in k_setup / k_grid / k_search / k_reduce .hip (each unit defines its own).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "device_types.h"

#include <hip/hip_vector_types.h>
#include <mutex>

// Launch state that belongs to a DEVICE, not to the process: hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the device that is current at the call, and
// the CU count is the device's.  mulls_create(device) allows contexts on several GPUs of one process (and every host thread may own one), so each launch wrapper
// keeps one DevLaunch per device, set up under a lock by the first launch on that device.  TAG: one table per wrapper.
struct DevLaunch
{
	bool ready = false, ok = false;
	size_t dyn_max[2] = {0, 0}; // dynamic LDS the wrapper's kernels may ask for on this device
	uint32_t n_cu = 256;
};
template <int TAG, class Init>
inline DevLaunch dev_launch(Init init)
{
	static std::mutex mu;
	static DevLaunch table[64];
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
		dev = 0;
	std::lock_guard<std::mutex> lock(mu);
	DevLaunch &D = table[dev];
	if (!D.ready)
	{
		int cus = 0;
		if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
			D.n_cu = (uint32_t)cus;
		D.ok = init(D);
		D.ready = true;
	}
	return D;
}
void launch_clone_src(hipStream_t st, uint32_t njobs, const Job *jobs, const CloudDesc *descs, const PairSetup *setup, const float4 *stage,
					  float4 *tmp_pos, float4 *tmp_nrm, uint32_t *bbox, const RunParams &rp);
void launch_crop(hipStream_t st, uint32_t npairs, CloudDesc *descs, const PairSetup *setup, const uint32_t *bbox, const float4 *stage,
				 const float4 *tmp_pos, const float4 *tmp_nrm, float4 *spos, float4 *snrm, float4 *tpos, float4 *tnrm, uint8_t *flag,
				 int32_t *match, float *wd, const RunParams &rp, GridDesc *grids, uint32_t nbig_segs, const Job *big_segs, uint32_t nbig_clouds,
				 const Job *big_clouds, uint32_t *seg_cnt, uint32_t *big_box);
// n byte ranges in one launch per MULLS_COPY_SEGS of them (device_types.h: CopySeg)
void launch_copy_segs(hipStream_t st, const CopySeg *segs, uint32_t n);
// CFilter::apply_motion_compensation on `n` 48-byte records in device memory (q: w x y z of Tran's rotation, t: its translation)
void launch_motion_comp(hipStream_t st, float4 *recs, uint32_t n, const double q[4], const double t[3], float thre);
void launch_thin(hipStream_t st, uint32_t npairs, CloudDesc *descs, const uint8_t *src_keep, const uint8_t *tgt_keep, float4 *spos, float4 *snrm,
				 float4 *tpos, float4 *tnrm);
// LDS tier with rp.tgt_map: crop + grid build of every target class cloud (<= MULLS_LDS_MAXPTS points) in one pass, no working copy (k_grid.hip)
int launch_tgt_grid(hipStream_t st, uint32_t npairs, CloudDesc *descs, const PairSetup *setup, const uint32_t *bbox, const float4 *stage, const RunParams &rp,
					GridDesc *grids, uint16_t *tmap, uint32_t *cell_start, float4 *tsorted);
// LDS tier without the fused setup (k_crop wrote the cropped copies)
void launch_grid_build_sort(hipStream_t st, uint32_t npairs, const CloudDesc *descs, GridDesc *grids, const RunParams &rp, const float4 *tpos, uint32_t *cell_start,
							float4 *tsorted);
// bitmap grids of the `nl` class clouds lclouds[] (pair * MULLS_NC + class each); tjobs: their 256-point chunks
void launch_bm_build(hipStream_t st, uint32_t nl, const uint32_t *lclouds, uint32_t ntjobs, const Job *tjobs, const CloudDesc *descs, GridDesc *grids, const float4 *tpos,
					 unsigned long long *bm, uint32_t *pf, uint32_t *cnt, uint32_t *cs, float4 *tsorted, uint32_t *rank);
size_t nn_lds_bytes(uint32_t cap, uint32_t maxcells, bool dedup);
int launch_nn_lds(hipStream_t st, uint32_t njobs, const Job *jobs, CloudDesc *descs, const PairState *states, const RunParams &rp, float4 *spos,
				  float4 *snrm, const GridDesc *grids, const uint32_t *cell_start, const float4 *tsorted, uint8_t *flag, int32_t *nn_idx,
				  float *nn_d2, unsigned long long *winner, const float4 *tnrm, int32_t *match, float *wd, const float4 *tpos, int32_t *nn_hint, float4 *mq, uint32_t cap, uint32_t maxcells,
				  uint32_t *wl, uint32_t *wl_ctr, uint32_t parity, bool first = false);
// a small mixed batch: the class clouds of both tiers in one launch (k_cert_mixed); returns 1 when it launched, 0 when the caller has to launch the tiers separately
int launch_cert_mixed(hipStream_t st, uint32_t n_lds, const Job *cjobs, uint32_t n_big, const Job *bjobs, uint32_t max_wgs, uint32_t rounds, CloudDesc *descs, const PairState *states,
					  const RunParams &rp, float4 *spos, float4 *snrm, const GridDesc *grids, const uint32_t *cell_start, const unsigned long long *bm, const uint32_t *pf,
					  const uint32_t *cs, const float4 *tsorted, uint8_t *flag, int32_t *nn_idx, float *nn_d2, unsigned long long *winner, const float4 *tnrm, int32_t *match,
					  float *wd, const float4 *tpos, int32_t *nn_hint, float4 *mq, uint32_t cap, uint32_t maxcells, bool first = false);
// global-memory tier (big_tier.h): class-level (MULLS_JOB_CLASS) and chunk-level jobs; max_wgs: chunk-level jobs are shared by 2 / 4 / 8 / 16 workgroups while
// the launch stays within this many
void launch_cert_big(hipStream_t st, uint32_t njobs, const Job *jobs, uint32_t max_wgs, CloudDesc *descs, const PairState *states, const RunParams &rp, float4 *spos, float4 *snrm,
					 const GridDesc *grids, const unsigned long long *bm, const uint32_t *pf, const uint32_t *cs, const float4 *tsorted, uint8_t *flag, int32_t *nn_idx,
					 float *nn_d2, unsigned long long *winner, const float4 *tpos, const float4 *tnrm, int32_t *nn_hint, int32_t *match, float *wd, float4 *mq);
void launch_nn(hipStream_t st, uint32_t njobs, const Job *jobs, CloudDesc *descs, const PairState *states, const RunParams &rp, float4 *spos,
			   float4 *snrm, const float4 *tpos, const uint8_t *flag, int32_t *nn_idx, float *nn_d2, unsigned long long *winner);
void launch_nn_shoot(hipStream_t st, uint32_t njobs, const Job *jobs, CloudDesc *descs, const PairState *states, const RunParams &rp,
					 float4 *spos, float4 *snrm, const float4 *tpos, const uint8_t *flag, int32_t *nn_idx, float *nn_d2, unsigned long long *winner);
void launch_filter(hipStream_t st, uint32_t njobs, const Job *jobs, CloudDesc *descs, const PairState *states, const RunParams &rp,
				   const float4 *snrm, const float4 *tnrm, uint8_t *flag, const int32_t *nn_idx, const float *nn_d2, int32_t *match, float *wd,
				   const unsigned long long *winner, const float4 *tpos, float4 *mq, bool big = false);
// leaders: job indices of the trip starts, grouped by trip length (split[0..3]: see k_reduce.hip)
void launch_accum(hipStream_t st, const uint32_t *leaders, const uint32_t split[4], const Job *jobs, const CloudDesc *descs, const PairState *states, const RunParams &rp,
				  const float4 *spos, const float4 *mq, const uint8_t *flag, float *wd, double *partial, bool single = false, uint32_t wave_min_trips = 0);
void launch_finish(hipStream_t st, uint32_t npairs, CloudDesc *descs, const PairState *states, const RunParams &rp, const double *partial,
				   PairOut *out, PairOut *out_host, const uint32_t *bbox, uint32_t *ticket, volatile uint32_t *host_epoch, uint32_t epoch,
				   uint32_t pair_base);
void launch_push_states(hipStream_t st, const PairState *host_states, PairState *dev_states, uint32_t npairs);
namespace mulls
{
struct IcpConst;
struct StepState;
} // namespace mulls
// lock-step loop with the O(1) half of the iteration on the device: initial per-pair state and first PairState; k_finish followed by the step (k_step)
void launch_step_init(hipStream_t st, uint32_t npairs, const PairSetup *setup, const mulls::IcpConst &K, mulls::StepState *steps, PairState *states);
void launch_finish_step(hipStream_t st, uint32_t pair_base, uint32_t npairs, CloudDesc *descs, PairState *states, const RunParams &rp, const mulls::IcpConst &K, const double *partial,
						PairOut *out, const uint32_t *bbox, mulls::StepState *steps, IcpOut *results, unsigned long long *host_word, uint32_t epoch, int brute,
						uint32_t *ticket = nullptr, bool sum_step = false); // ticket: two zeroed device words -> finish, step and publication in one launch (small batches);
						// sum_step (large batches): one wave per pair sums and steps (k_sum_step) instead of k_finish + k_step
void launch_transform_aos(hipStream_t st, float4 *recs, uint32_t n, const double *T12);
// the same on `count` (<= 6) clouds in one launch, T12 (host) by value
void launch_transform_clouds(hipStream_t st, float4 *const recs[], const uint32_t n[], int count, const double T12[12]);
void launch_set_corr(hipStream_t st, uint32_t src_off, const int32_t *cs, const int32_t *ct, const float *cd, uint32_t n, uint8_t *flag,
					 int32_t *match, float *wd, uint32_t tgt_off, const float4 *tpos, const float4 *tnrm, float4 *mq);

// device-resident registration loop (k_icp.hip): one launch = every iteration of `npairs` pairs
namespace mulls
{
struct IcpConst;
}
struct mulls_iter_trace;
int launch_icp(hipStream_t st, uint32_t npairs, uint32_t pair_base, const Job *rjobs, const uint32_t *pair_rjob, const uint32_t *order, uint32_t *queue,
			   CloudDesc *descs, const PairSetup *setup, const RunParams &rp, const mulls::IcpConst &K, float4 *spos, float4 *snrm, const GridDesc *grids,
			   const uint32_t *cell_start, const float4 *tsorted, uint8_t *flag, int32_t *nn_idx, float *nn_d2, unsigned long long *winner, const float4 *tnrm,
			   int32_t *match, float *wd, const float4 *tpos, int32_t *nn_hint, float4 *mq, const uint32_t *bbox, uint32_t cap, uint32_t maxcells, IcpOut *outs,
			   mulls_iter_trace *trace, uint32_t trace_cap);
