// device_types.h — plain structs shared by the HIP kernels (k_*.hip) and the host driver (driver.cpp).
// Layouts live in HBM exactly as declared here; see DESIGN.md §"Data layout in HBM".
#pragma once
#include <stdint.h>

#define MULLS_NC 6		  // feature classes, index == used_feature_type character index
#define MULLS_NTERM 27	  // 21 packed normal-matrix terms + 6 right-hand-side terms
#define MULLS_BLOCK 256	  // threads per workgroup = 4 wave64
#define MULLS_SRC_PER_THREAD 2 // source points per lane in the filter / accumulate kernels
#define MULLS_SRC_PER_BLOCK (MULLS_BLOCK * MULLS_SRC_PER_THREAD) // = 512 source points per job
#define MULLS_NN_BLOCK 128	  // correspondence search: 2 wave64 per workgroup ...
#define MULLS_NN_PTS 4		  // ... x 4 register-blocked source points per lane = the same 512 points per job
#define MULLS_TILE 2048		  // target points staged per LDS tile (3 planar float arrays -> 24 KiB)

#define MULLS_MAXCELLS 65536u // most cells of one LDS-tier target grid (what fits next to the points decides, see driver.cpp)
#define MULLS_BM_MAXWORDS (1u << 22) // global-memory tier: most 64-cell occupancy words of one grid (268 M cells)
#define MULLS_BM_TOTALWORDS (1u << 28) // ... and of all grids of a batch together (2 GiB of bitmap + 1 GiB of ranks)
#define MULLS_BM_H0 0.25f // ... smallest cell edge in metres (dense maps); sparse clouds start from their mean point spacing, up to 0.7 m
#define MULLS_LDS_BLOCK 1024		// LDS grid tier: 16 wave64 = 64 sub-groups per workgroup, one 512-point job
#define MULLS_LDS_MAXPTS 9728u // largest target class cloud staged in LDS (14 B per point; the uint16 cell table takes what is left of 160 KiB, >= 4096 cells)
#define MULLS_ACC_BLOCK 128	   // k_accum: two waves per 512-point job, four points per lane
#define MULLS_LDS_QCHUNK 1024u // LDS tier: queries searched between two workgroup barriers (one per lane in the rigid-step phase)
#define MULLS_LDS_AUX (320u + 2u * MULLS_LDS_QCHUNK) // LDS tier: cost histogram and query order of a chunk
#define MULLS_CERT_BLOCK 512	   // k_cert: lanes per class cloud (one source point per lane and trip); several workgroups per CU
#define MULLS_CERT_SMALL 64u   // the device-resident loop searches up to this many uncertified points of a class cloud against the grid in global memory
#define MULLS_CERT_SMALL_LOCKSTEP 512u // ... k_cert up to this many (four workgroups per CU hide the walks' latency: 64 -> 512 took 1.5 ms off a 4096-pair step, profiles/r03_sweeps.txt)
#define MULLS_ICP_STATIC_LDS 9216 // LDS the device-resident loop keeps next to the dynamic block (pair state, class rows, ...; checked at its first launch)
#ifndef MULLS_LDS_GROUP // (4u / 16u: A/B builds, tools/build_variant.sh)
#define MULLS_LDS_GROUP 8u	   // lanes that cooperate on one query in the LDS grid tier (DPP reductions stay inside a 16-lane row)
#endif
#define MULLS_BIG_CLOUD 16384u // class clouds above this size (target or source) are cropped segment-wise (k_crop_big_*): one workgroup walking a 100 k-point
							   // cloud alone took 290 us (profiles/r04_large_base.txt)
#define MULLS_BIG_SRC_SIDE 0x100u // Job::cls flag in the segment tables: the segment belongs to the pair's SOURCE cloud of that class
#define MULLS_SEG 4096u		  // ... in segments of this many points
#define MULLS_GRID_GROUP 16u   // lanes that cooperate on one query in the grid search tier
#define MULLS_BIG_BLOCK 512	   // k_cert_big: lanes per job of the global-memory tier (big_tier.h)
#define MULLS_BIG_CLASS_MAX (3u * MULLS_BIG_BLOCK) // ... and the largest source class cloud one workgroup takes whole (class-level job: three points per lane)
#define MULLS_JOB_CLASS 0x80000000u // Job::count flag of the global-memory tier: the job is a whole source class cloud (the workgroup runs the rejection chain itself)
#define MULLS_SMALL_SRC_MAX 4096u  // auto mode: source class clouds above this size go to the global-memory tier's chunk-level jobs whatever their target's size
// search tier of one (pair, class) cloud pair: CloudDesc::tier
#define MULLS_TIER_BRUTE 0u // LDS-tiled brute force (k_nn)
#define MULLS_TIER_BM 1u	// occupancy-bitmap grid in global memory (k_cert_big)
#define MULLS_TIER_LDS 2u	// dense grid, target cloud staged in LDS (k_cert / k_nn_lds)
#define MULLS_GRID_H0 1.3f	   // preferred cell edge in metres (measured optimum on the bench workload, profiles/r03_sweeps.txt); grows until the cloud's box fits the cell budget

// bits of the per-source-point flag byte
#define MULLS_F_ALIVE 1u // still part of the source cloud (reference: survived every compaction, cregistration.hpp:1755-1792)
#define MULLS_F_VALID 2u // member of Corr_f, i.e. enters the estimation (:1794-1830)

// One feature-class cloud pair of one registration.  Points are SoA float4: pos = (x,y,z,intensity),
// nrm = (nx,ny,nz,curvature).  Offsets index the batch-wide arenas.
struct CloudDesc
{
	uint32_t src_stage; // where this cloud starts in the staged upload, in float4 units; layout per cloud: stage_fmt
	uint32_t tgt_stage;
	uint32_t src_n0; // staged point count
	uint32_t tgt_n0;
	uint32_t src_off; // first element in the working SoA arenas
	uint32_t tgt_off;
	uint32_t src_n; // count after the intersection filter (fixed for the rest of the registration)
	uint32_t tgt_n;
	uint32_t alive_cur;	 // live source points at the start of the current iteration (gates :1727 and :1755)
	uint32_t alive_next; // accumulator filled by k_filter
	uint32_t n_matched;	 // |Corr| of the current iteration (before duplicate removal), filled by k_nn
	uint32_t valid_next; // accumulator filled by k_filter
	uint32_t n_valid;	 // |Corr_f| currently in force (may be stale, SURVEY B-4)
	uint32_t job_begin;	 // this cloud's range in the job table
	uint32_t job_end;
	uint32_t sd_stage; // staged block2->pc_*_down records (motion undistortion regenerates the source from them); may alias src_stage
	uint32_t sd_n0;
	uint32_t src_cap; // slots reserved for this source cloud in the working arenas = max(src_n0, sd_n0)
	uint32_t big_slot; // target clouds beyond MULLS_BIG_CLOUD points are cropped by many workgroups: 1-based slot, 0 = small
	uint32_t src_big_slot; // ... and source clouds (0 = small)
	uint32_t n_search; // LDS tier: queries of this class cloud that went through the grid search in the last iteration (the others were certified)
	uint32_t stage_fmt; // staged layout of the source (bits 0-1), target (2-3) and src_down (4-5) clouds: MULLS_STAGE_*
	uint32_t tier;		// search tier of this cloud pair (MULLS_TIER_*), assigned per run by the host (assign_tiers): decides which setup kernels build its target
						// grid and which search kernel's job table holds it
	uint32_t grid_slot; // its slot in that tier's cell tables (LDS tier: x RunParams::cell_stride entries; bitmap tier: x RunParams::bm_stride words)
};
// staged layouts of one cloud of n points (load_staged, device_util.h)
#define MULLS_STAGE_AOS48 0u  // n x 3 float4: the caller's 48-byte PointXYZINormal records (device-resident map clouds, copied device to device)
#define MULLS_STAGE_PACK32 1u // n float4 (x y z intensity), then n float4 (nx ny nz curvature): what the host gathers out of the caller's records
#define MULLS_STAGE_PACK28 2u // n float4 (x y z intensity), then 3n floats (nx ny nz): no curvature (only motion undistortion reads it)

// Uniform grid over one cropped target-class cloud (exact fixed-radius search tier).  cell id = (cz*ny + cy)*nx + cx,
// x fastest, so the cells cx0..cx1 of one (cy,cz) row are one contiguous range of the cell-sorted target array.
struct GridDesc
{
	float ox, oy, oz; // origin = minimum corner of the cloud
	float inv_h, h;
	uint32_t nx, ny, nz;
	uint32_t ncell;	   // LDS tier: number of cells.  Global-memory tier: number of 64-cell occupancy words (= ny * nz * wpr)
	uint32_t cell_off; // first entry of this cloud in the batch-wide cell table (LDS tier: ncell + 1 entries) / bitmap + rank arrays
	uint32_t wpr;	   // global-memory tier: occupancy words per (cy,cz) row = ceil(nx / 64); bit = ((cz*ny + cy)*wpr)*64 + cx
	uint32_t nocc;	   // global-memory tier: occupied cells (k_bm_scan)
};

// Per-pair state rewritten by the host before every lock-step iteration (one H2D copy for the whole batch).
struct PairState
{
	double T[12];	 // rows of [R|t] applied to the source this iteration (TempTran; identity at i = 0)
	double x[6];	 // last solved step, used by the residual pass
	float thr[MULLS_NC]; // dis_thre per class for this iteration's search
	int32_t iter;	 // iteration number i (feeds the adaptive range weight and the residual-weight gate)
	int32_t active;	 // 1: run search + estimation this iteration
	int32_t want_residual; // 1: run the posterior residual pass instead (pair already converged)
	int32_t pad_[3];	   // sizeof(PairState) = 192, a multiple of 16 (k_push_states moves uint4 words)
};

struct PairSetup // written once per run
{
	double guess[12];	 // rows of the initial guess [R|t]
	double tgt_bound[6]; // block1->local_bound
	double inv_q[4];	 // quaternion (w,x,y,z) of inverse(initial_guess)'s rotation, for motion undistortion (cfilter.hpp:493-516)
	double inv_t[4];	 // its translation (x,y,z,-)
};

// Per-pair output of one lock-step iteration.  k_finish fills it in HBM; k_pull_outs sends the 128-B counter block and
// the sums of the USED classes (224 B each) to pinned host memory as one packed record per pair.
#define MULLS_NTERM_PAD 28 // class rows padded to whole uint4 words
struct PairOut
{
	double sums[MULLS_NC][MULLS_NTERM_PAD]; // per class: 21 packed terms (row-major upper enumeration) + 6 rhs; residual pass: [0]=VTPV [1]=n
	double comb[MULLS_NTERM_PAD];			// rp.pull_comb: the classes combined as the reference assembles them (k_finish) — the only row sent to the host
	uint32_t n_valid[MULLS_NC];
	uint32_t n_alive[MULLS_NC];
	uint32_t src_n[MULLS_NC];
	uint32_t tgt_n[MULLS_NC];
	uint32_t bbox[6]; // ordered-key bounding box of the transformed ground/pillar/facade source clouds (k_clone_src)
	uint32_t pad_[2];
};

// Result of one pair of the device-resident loop (k_icp), read back by the host after the launch.
struct IcpOut
{
	double T[16];	 // Trans1_2, column-major
	double info[36]; // information matrix, column-major
	double sigma2;
	unsigned long long src_pts, tgt_pts, corr_pts; // profile counters: live source points / target points of the searched class clouds / valid correspondences, summed over the iterations
	float ratio;
	int32_t code, iters, singular, trace_len;
	uint32_t ncorr[MULLS_NC], nsrc0[MULLS_NC], ntgt0[MULLS_NC], bbox[6];
	uint32_t pad_;
	unsigned long long t_fused[6]; // ... of the fused class pass by stage: set-up, stage 1, leftovers, stage 3, stage 4, (unused)
	uint32_t t_search_it[24]; // ... and of the search phase of the first 24 iterations
	unsigned long long t_phase[6]; // time of this pair in the loop's phases, 10-ns ticks (wall_clock64): search, counters + count test, normal equations, solve, residual pass, total
	unsigned long long tgt_job_pts, pair_evals; // lock-step loop with the device step only: target points times the workgroups that read them; brute-force pair evaluations
};

// One workgroup's worth of the correspondence search / filter / accumulation.
struct Job
{
	uint32_t pair;
	uint32_t cls;
	uint32_t start; // first source point (relative to the cloud) handled by this workgroup
	uint32_t count; // LDS-tier jobs only: number of consecutive source points this workgroup searches (0 = MULLS_SRC_PER_BLOCK)
};

// One launch moves up to MULLS_COPY_SEGS byte ranges (k_copy_segs): the small tables a fill / a run uploads (their host copies wait in a host-mapped mailbox) and the
// device-resident clouds a pair points to — instead of one copy command each.  Addresses 16-byte aligned, sizes multiples of 4.
#define MULLS_COPY_SEGS 24
struct CopySeg
{
	unsigned long long dst, src;
	uint32_t bytes, pad_;
};
struct CopyArgs
{
	CopySeg seg[MULLS_COPY_SEGS];
};

// Run-wide constants (kernel argument, by value).
struct RunParams
{
	uint8_t used[MULLS_NC];
	uint8_t w_balance, w_resid, w_dist, w_inten; // weight_strategy[0..3]
	uint8_t crop;								 // apply_intersection_filter (forced off while undistorting, cregistration.hpp:1186)
	uint8_t undistort;							 // apply_motion_undistortion_while_registration
	uint8_t normal_shooting;					 // normal_shooting_on: ground / facade / roof use the 10-NN normal-shooting search
	uint8_t faithful;
	uint8_t rej_strict; // CorrespondenceRejectorDistance: 0 keep distance <= max^2 (NaN kept), 1 keep distance < max^2 (mulls_params.rejector_strict)
	float z_xy_ratio;
	float win_pt, win_pl, win_li;
	uint8_t force_class_w; // stage-level API: take class_w_value instead of the balance rule
	uint8_t bm_auto; // bitmap grids: cell edge = clamp(sqrt(dx * dy / n), bm_h0, 2.8 * bm_h0) per cloud instead of bm_h0
	float class_w_value;
	double cos_bearing; // cos(normal_bearing / 180.0 * M_PI) in double, computed on the host
	int32_t resid_from_iter; // residual weighting applies when iter_num > this (2 for mm_lls_icp, cregistration.hpp:1905-1907; -1 for the 3-DoF variant)
	uint32_t debug_stop;	// diagnostics only (env MULLS_DEBUG_STOP): 1 = k_nn_lds returns after the transform, 2 = after staging
	uint32_t cell_stride;	// entries reserved per cloud in the cell tables (multiple of 4: uint4-aligned), >= grid_maxcells + 1
	uint32_t grid_maxcells; // cell budget of the target grids built by k_crop (MULLS_MAXCELLS, or what fits in LDS for the LDS tier)
	float grid_h0;		// LDS tier: preferred cell edge (MULLS_GRID_H0; grows until the cloud's box fits grid_maxcells)
	float bm_h0;		// bitmap grids (MULLS_TIER_BM clouds): smallest cell edge ...
	uint32_t bm_maxwords; // ... most 64-cell occupancy words of one grid ...
	uint32_t bm_stride;	  // ... and the words reserved per grid slot in the bitmap / rank arrays
	uint32_t pull_comb;	// mm_lls_icp loop: k_finish combines the class rows (normal matrix + rhs, or VTPV + count) and k_pull_outs sends that one row
	uint32_t lds_dedup;	// LDS tier with class-level jobs: the duplicate rule is resolved inside k_nn_lds (winner table in LDS), losers get nn_idx = -1
	uint32_t tick_base; // duplicate-table epoch of iteration 0 of this run (see k_nn)
	// LDS tier, certified correspondences (k_nn_lds): a query whose previous nearest target is provably still the nearest skips the
	// search.  The sweep of a searched query is widened by slack = clamp(rate * (distance the point moved), min, max) metres so that
	// the bound it leaves behind survives the next (smaller) step.  cert = 0 switches the whole mechanism off (diagnostics).
	uint32_t cert;
	float cert_slack_min, cert_slack_max, cert_slack_rate;
	// k-candidate certificates (round 5).  A search leaves behind, next to the hint, the nearest targets its OTHER lanes saw (CandRec) and a bound on every
	// target outside that set; a point whose hinted target fails the certificate evaluates the handful of candidates exactly — same distance expression,
	// lowest index on ties — and is certified when the nearest of them beats that bound by the margin the plain certificate asks for.  kcert = 0: off.
	// (The LDS tier is built without it by default — lds_tier.h: MULLS_LDS_KCERT.)
	uint32_t kcert;
	uint32_t kcert_min; // LDS tier: a leftover list gets the look only from this length on (a look is one chain of four round trips for the whole list,
						// a search round of 64 queries about as much: below a few rounds' worth the look costs what it saves)
	uint4 *cand; // per source point: x, y = candidates besides the hinted target (LDS tier: 4 x uint16, 0xffff = none; global-memory tier: 2 x uint32,
				 // 0xffffffff = none), z = float bits of (bound on the targets outside the set) - (bound on the targets other than the hint), w = epoch
				 // (tick_base + iteration of the search that wrote the record: a record of an earlier run, or none, never passes the epoch test)
	// LDS tier without a working copy of the target clouds (k_tgt_grid): record m of a cropped target class cloud is record tgt_map[tgt_off + m] of
	// the staged cloud at tgt_stage (tgt_record(), device_util.h).  Null: the cropped copies tpos / tnrm exist (k_crop).
	const float4 *tgt_stage;
	const uint16_t *tgt_map;
	unsigned long long *dbg_ticks; // diagnostics (MULLS_OPT_DEBUG_STOP = 20): k_cert's one-pass walk adds its phase times here (10-ns ticks; [6] = workgroups)
};
