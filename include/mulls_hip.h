/*
 * mulls_hip.h — C ABI of libmulls_hip.so, the MI355X (gfx950) implementation of the MULLS-ICP hot path.
 *
 * What this boundary replaces (all citations relative to the MULLS reference tree):
 *   CRegistration<PointT>::mm_lls_icp()            include/common/cregistration.hpp:1114-1440
 *   CRegistration<PointT>::determine_corres()      include/common/cregistration.hpp:1701-1835
 *   multi_metrics_lls_tran_estimation() + pt2pl/pt2li/pt2pt summations
 *                                                  include/common/cregistration.hpp:1869-2275
 *   get_multi_metrics_lls_residual()               include/common/cregistration.hpp:2518-2677
 *   intersection_filter()                          include/common/cregistration.hpp:2894-2922
 *
 * The reference has no FFI: mm_lls_icp is a member of a header-only class template.  The drop-in is therefore
 * (i) this C ABI (plain pointers + sizes, no C++/torch types) and (ii) include/cregistration_hip.hpp, which
 * re-creates the member function with its verbatim signature and marshals pcl clouds into the structs below
 * (see INTEGRATION.md).
 *
 * Conventions
 *   - Feature-class index == character index of the reference's `used_feature_type` string
 *     (cregistration.hpp:1196-1201, :1213-1232): 0 ground, 1 pillar, 2 facade, 3 beam, 4 roof, 5 vertex.
 *   - Points are pcl::PointXYZINormal records (48 B): x@0 y@4 z@8 | normal_x@16 normal_y@20 normal_z@24 |
 *     intensity@32 curvature@36.  `stride` is the byte distance between records (48 for PCL clouds).
 *   - Matrices are column-major doubles (Eigen's default), 4x4 -> [16], 6x6 -> [36].
 *   - Every entry point returns 0 on success or a negative MULLS_E_* infrastructure error.  The reference's
 *     registration status (1, -1, -2, -3, 0; cregistration.hpp:1131-1136) is returned in mulls_result.code.
 *     Nothing is thrown across the ABI.
 *   - Caller memory is only borrowed for the duration of a call.
 */
#ifndef MULLS_HIP_H
#define MULLS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MULLS_NCLASS 6
#define MULLS_POINT_BYTES 48

enum mulls_class
{
	MULLS_GROUND = 0, /* point-to-plane */
	MULLS_PILLAR = 1, /* point-to-line  */
	MULLS_FACADE = 2, /* point-to-plane */
	MULLS_BEAM = 3,	  /* point-to-line  */
	MULLS_ROOF = 4,	  /* point-to-plane */
	MULLS_VERTEX = 5  /* point-to-point */
};

enum mulls_error
{
	MULLS_OK = 0,
	MULLS_E_INVALID = -100,	   /* bad argument */
	MULLS_E_HIP = -101,		   /* a HIP runtime call failed (see mulls_last_error) */
	MULLS_E_NO_DEVICE = -102,  /* no gfx950 device / kernels could not be loaded */
	MULLS_E_UNSUPPORTED = -103, /* option not implemented by this build */
	MULLS_E_IO = -104,		   /* file could not be opened / is not in the expected format */
	MULLS_E_NOMEM = -105	   /* a host allocation failed (e.g. sizes that cannot be real) */
};

/* One feature-class cloud, borrowed from the caller (AoS of 48-byte PointXYZINormal records). */
/* pts may also be a device pointer obtained from mulls_map_cloud() (stride 48 only). */
typedef struct mulls_cloud
{
	const void *pts;
	uint32_t n;
	uint32_t stride;
} mulls_cloud;

/* One registration problem = the reference's constraint_t inputs (utility.hpp:561-590).
 *   tgt[c]      block1->pc_{ground,pillar,facade,beam,roof,vertex}                 (cregistration.hpp:1180)
 *   src[c]      block2->pc_*_down, or block2->pc_* when use_more_points; src[5] is always block2->pc_vertex (:1181)
 *   src_down[c] block2->pc_*_down, only read when params.undistort (cregistration.hpp:1251-1253); may be all-zero
 *   tgt_bound   block1->local_bound as {min_x,min_y,min_z,max_x,max_y,max_z}       (cregistration.hpp:2916)
 *   init_guess  the by-value Eigen::Matrix4d initial_guess, column-major           (cregistration.hpp:1120) */
typedef struct mulls_pair
{
	mulls_cloud tgt[MULLS_NCLASS];
	mulls_cloud src[MULLS_NCLASS];
	mulls_cloud src_down[MULLS_NCLASS];
	double tgt_bound[6];
	double init_guess[16];
} mulls_pair;

/* The positional arguments of mm_lls_icp (cregistration.hpp:1114-1123), in order, plus ABI-only knobs at the end. */
typedef struct mulls_params
{
	int32_t max_iter_num;			 /* 20 */
	float dis_thre_unit;			 /* 1.5 */
	float converge_translation;		 /* 0.002 */
	float converge_rotation_d;		 /* 0.01 */
	float dis_thre_min;				 /* 0.4 */
	float dis_thre_update_rate;		 /* 1.1 */
	char used_feature_type[8];		 /* "111110" */
	char weight_strategy[8];		 /* "1101" */
	float z_xy_balanced_ratio;		 /* 1.0 */
	float pt2pt_residual_window;	 /* 0.1 */
	float pt2pl_residual_window;	 /* 0.1 */
	float pt2li_residual_window;	 /* 0.1 */
	uint8_t apply_intersection_filter; /* true */
	uint8_t apply_motion_undistortion; /* false */
	uint8_t normal_shooting_on;		   /* false */
	uint8_t use_more_points;		   /* false (only decides which clouds the adapter passes as src[]) */
	float normal_bearing;			 /* 45.0 */
	uint8_t keep_less_source_points; /* false */
	uint8_t faithful;				 /* ABI-only. 1 = reproduce reference quirks (pt2li off-diagonals dropped,
										vertex residual weighted by d^2); 0 = mathematically intended version */
	uint8_t rejector_strict;		 /* ABI-only. 1 (default): pcl::registration::CorrespondenceRejectorDistance::getRemainingCorrespondences as PCL 1.7-1.12 ship it
										(registration/src/correspondence_rejection_distance.cpp: `original_correspondences[i].distance < max_distance_`,
										max_distance_ = the float square set by setMaximumDistance; a NaN distance is dropped).  0: `distance <= max^2` (NaN kept),
										the reading SURVEY.md A.4-3 wrote down.  The two differ only on correspondences whose float distance equals the float
										threshold exactly (tests/test_pcl_operators.py counts them: none on the bench workload) */
	uint8_t reserved_;
	float sigma_thre;				 /* 0.5 */
	float min_neccessary_corr_ratio; /* 0.03 */
	float max_bearable_rotation_d;	 /* 45.0 */
	uint64_t rng_seed;				 /* ABI-only. keep_less_source_points thins with an order-preserving selection sampling (Knuth's
										Algorithm S driven by splitmix64(rng_seed ^ cloud id)); the reference uses pcl::RandomSample
										seeded with time(NULL) (cfilter.hpp:620), i.e. a different subset on every run */
} mulls_params;

/* Optional per-iteration record (debug / parity triage).  Filled when mulls_result.trace != NULL. */
typedef struct mulls_iter_trace
{
	int32_t iter;
	uint32_t ncorr[MULLS_NCLASS]; /* correspondences that entered the estimation (after all rejectors) */
	uint32_t nsrc[MULLS_NCLASS];  /* live source points of the class after this iteration's search */
	float thr[MULLS_NCLASS];	  /* dis_thre used by this iteration's search */
	double atpa[36];			  /* normal matrix after the mirror step (cregistration.hpp:1924-1938) */
	double atpb[6];
	double x[6];				  /* tx ty tz roll pitch yaw of this step */
} mulls_iter_trace;

typedef struct mulls_result
{
	int32_t code;		/* 1 ok, -1 step too large, -2 too few correspondences, -3 sigma too large, 0 loop never ran */
	int32_t iters;		/* iterations whose correspondence search ran */
	double T[16];		/* constraint_t::Trans1_2, column-major (cregistration.hpp:1405) */
	double info[36];	/* constraint_t::information_matrix, column-major (:1418) */
	float sigma;		/* constraint_t::sigma (:1419) */
	float confidence;	/* constraint_t::confidence (:1420) */
	uint32_t ncorr[MULLS_NCLASS]; /* correspondences per class at the last search */
	uint32_t nsrc0[MULLS_NCLASS]; /* source points per class after the intersection filter */
	uint32_t ntgt0[MULLS_NCLASS]; /* target points per class after the intersection filter */
	int32_t singular;			  /* ABI-only: 1 if a non-finite solve was observed (reference does not check, B-11) */
	int32_t cropped;			  /* 1 if the intersection filter ran (cregistration.hpp:1186-1188) */
	double crop_box[6];			  /* its box {min_x,min_y,min_z,max_x,max_y,max_z} (utility.hpp:857-865); the adapter re-creates
									 the reference's kd-tree side effect on block1 from it */
	float ms_total;				  /* wall time of this registration inside the library (batch: batch time / n) */
	mulls_iter_trace *trace;	  /* in: caller array or NULL */
	int32_t trace_cap;			  /* in: capacity of trace[] */
	int32_t trace_len;			  /* out */
} mulls_result;

/* Per-kernel device time of the last mulls_batch_run, measured with hipEvents on the library's stream
 * when mulls_set_profiling(ctx, 1) is on (adds one event pair per launch group and waits on events; keep off for throughput runs).
 * mulls_set_profiling(ctx, 2): only the correspondence search is bracketed (ms_nn, launches_nn and the counters; the other ms_* stay 0) —
 * two events per iteration, the iteration hand-over as without profiling. */
typedef struct mulls_profile
{
	double ms_setup;	  /* clone + initial guess + intersection filter kernels */
	double ms_nn;		  /* correspondence-search kernel, summed over launches */
	double ms_filter;	  /* duplicate / distance / direction rejection kernel */
	double ms_accum;	  /* normal-equation accumulation kernel (+ partial finish) */
	double ms_residual;	  /* posterior residual kernel */
	int32_t launches_nn;  /* number of correspondence-search launches in the run */
	int32_t iterations;	  /* lock-step iterations executed by the run */
	uint64_t nn_pair_evals; /* source-target distance evaluations issued by those launches */
	uint64_t nn_src_pts;	/* live source points searched, summed over launches */
	uint64_t nn_tgt_pts;	/* target points streamed into LDS (once per 512-source job), summed over launches */
	double ms_host_step;	/* host time spent in the per-iteration algebra (6x6 solves, tests, next states), summed over iterations; 0 when the
							   loop steps on the device (every run without per-iteration traces) */
	double ms_host_wait;	/* host time spent waiting for the device epoch, summed over iterations */
	double ms_host_launch;	/* host time spent enqueueing the launch set, summed over iterations */
	uint64_t nn_tgt_unique; /* target points of the searched class clouds (once per cloud), summed over launches */
	uint64_t nn_corr_pts;	/* device-resident loop: correspondences that entered the estimation, summed over the iterations */
	double icp_fused_ms[6];	/* device-resident loop: the fused class pass by stage (set-up, rigid step + certificates, leftover queries, rejection
							   chain, normal-equation terms), summed over the pairs */
	double icp_search_ms[24]; /* device-resident loop: the search phase by iteration (summed over the pairs) */
	double icp_phase_ms[6]; /* device-resident loop: workgroup time summed over the pairs, by phase: search, counters + count test, normal
							   equations, solve + step tests, residual pass, whole loop (one workgroup per CU: divide by the CUs for wall time) */
	double ms_stage;		/* mulls_icp / mulls_icp_batch: wall time of staging the caller's clouds (host gather into pinned memory + upload), */
	double ms_stage_pack;	/* ... of which the host gather, */
	uint64_t stage_bytes;	/* ... and the bytes that crossed PCIe */
} mulls_profile;

typedef struct mulls_ctx mulls_ctx;		/* one per host thread / HIP stream */
typedef struct mulls_batch mulls_batch; /* device-resident set of pairs */

void mulls_default_params(mulls_params *p);

int mulls_create(int device, mulls_ctx **out);
void mulls_destroy(mulls_ctx *ctx);
const char *mulls_last_error(const mulls_ctx *ctx);
int mulls_set_profiling(mulls_ctx *ctx, int on);
int mulls_get_profile(const mulls_ctx *ctx, mulls_profile *out);
/* correspondence-search tier: 0 = auto, 1 = LDS-tiled brute force, 2 = uniform grid in global memory, 3 = uniform grid staged in LDS with lock-step
 * launches (MULLS_E_INVALID when a searched target class cloud exceeds 9728 points), 4 = device-resident loop (k_icp: one launch iterates every pair
 * of the batch to the end; the lock-step LDS tier where the loop does not apply: normal shooting, source class clouds above 16384 points).
 * Auto: searched target class clouds of <= 9728 points -> the grid staged in LDS, stepped by lock-step launch sets (the O(1) half of the iteration
 * on the device too; the device-resident loop only inside the window MULLS_OPT_RESIDENT_MIN_PAIRS .. _MAX_PAIRS, empty by default); larger targets ->
 * the grid in global memory.  All tiers are exact and return bit-identical results (tests/test_gpu_stages.py, test_gpu_icp.py). */
int mulls_set_nn_mode(mulls_ctx *ctx, int mode);
/* Execution options of a context (none of them changes a result: every path returns the same bits).  mulls_create presets each from the environment
 * variable named after it (MULLS_OPT_HOST_STEP <- MULLS_HOST_STEP=1, ...: diagnostics and the A/B scripts under tools/); nothing reads the
 * environment after that. */
enum mulls_option
{
	MULLS_OPT_HOST_STEP = 0,			  /* [0] 1: the lock-step loop is stepped by the host (what per-iteration traces switch on anyway) */
	MULLS_OPT_RESIDENT_MIN_PAIRS = 1,	  /* [1] auto mode runs the device-resident loop (one persistent workgroup per pair) for batches of MIN .. MAX pairs; MAX < MIN (the  */
	MULLS_OPT_RESIDENT_MAX_PAIRS = 2,	  /* [0] default): never — since round 3 the lock-step path is at least as fast at every size (profiles/r03_modes.txt: 256 pairs
											 139 k both, 320 pairs 152 k against 100 k).  nn_mode 4 asks for the loop at any size */
	MULLS_OPT_FEW_LAUNCHES_MAX_PAIRS = 3, /* [640; 384 until round 6: +1 % at 384 - 512 pairs] lock-step loop: batches up to this size run 3 - 4 launches per iteration instead of 7 (one accumulation launch;
											 finish + step + publication as one kernel; light and heavy pass of the search as one launch while there are at most
											 two class clouds per CU) — small batches are bound by the launch count */
	MULLS_OPT_SUBBATCHES = 4,			  /* [0 = by batch size] host-stepped loop: sub-batches in flight (1 or 2) */
	MULLS_OPT_TWO_STREAMS = 5,			  /* [0] host-stepped loop: the second sub-batch on a second stream */
	MULLS_OPT_CERTIFICATES = 6,			  /* [1] LDS tier: certified correspondences (0: every point is searched every iteration; MULLS_NO_CERT=1) */
	MULLS_OPT_CERT_SLACK_MIN = 7,		  /* [0.02 m] how much farther than the hinted target a searched query sweeps: */
	MULLS_OPT_CERT_SLACK_MAX = 8,		  /* [0.10 m]   clamp(rate * distance moved, min, max)                          */
	MULLS_OPT_CERT_SLACK_RATE = 9,		  /* [1.0] */
	MULLS_OPT_LDS_DEDUP = 10,			  /* [1] LDS tier: duplicate rule and rejection chain inside the search kernels (0: k_filter; MULLS_NO_LDS_DEDUP=1) */
	MULLS_OPT_GRID_H0 = 11,				  /* [0 = 1.3 m] LDS tier: preferred cell edge */
	MULLS_OPT_BM_H0 = 12,				  /* [0 = from the point spacing] global-memory tier: one fixed cell edge for every cloud */
	MULLS_OPT_LEAN_STAGING = 13,		  /* [0] mulls_icp / mulls_icp_batch stage only what the registration reads: the classes of used_feature_type (plus the
											 source ground / pillar / facade clouds the intersection box is taken from).  mulls_result.nsrc0 / ntgt0 of the
											 classes left out report 0; every output of the reference's interface is unchanged.  The C++ bridge switches it on. */
	MULLS_OPT_DEBUG_STOP = 14,			  /* [0] kernel bring-up switches (tools/gpu_time_nn.py) */
	MULLS_OPT_DEBUG_TICK = 15,			  /* [0] tests: start a fresh batch's duplicate-table epoch counter here */
	MULLS_OPT_SPLIT_MIN_PAIRS = 16,		  /* [96]    lock-step loop stepped on the device: batches of MIN .. MAX pairs iterate as two sub-batches on two streams, */
	MULLS_OPT_SPLIT_MAX_PAIRS = 17,		  /* [2^30]  so that one half's kernels fill the gaps of the other's (MAX < MIN: never; not while profiling: +3 % at */
										  /*         128 - 4096 pairs, nothing below 96, profiles/r03_modes.txt) */
	MULLS_OPT_FUSED_TGT_SETUP = 18,		  /* [1] LDS tier: the target class clouds are cropped and their grids built in one pass without a cropped working */
										  /*     copy (k_tgt_grid); 0 = k_crop + k_grid_build_sort.  Same results */
	MULLS_OPT_STAGGER = 19,				  /* [4352] bytes by which the k-th per-point array of a batch starts into its 2 MiB-aligned allocation (k x this): the same index of a dozen
										     arrays is then not the same offset into a dozen pages (+1.3 % at 4096 pairs, profiles/r03_sweeps.txt) */
	MULLS_OPT_STEP_LAUNCH_MAX_PAIRS = 20, /* [640] lock-step loop: batches up to this size run finish + step + publication as one launch (k_finish_step) even when they are
											 above FEW_LAUNCHES_MAX_PAIRS (which implies it): +2 % at 512 pairs, -2 % at 1024, -7 % at 4096 (one atomic per pair on one word) */
	MULLS_OPT_MIXED_TIERS = 21,			  /* [1] auto mode picks the search tier per (pair, class) cloud: the LDS tier for the down-sampled clouds, the global-memory
											 tier (certified correspondences on an occupancy-bitmap grid) for larger ones, both in one launch set; 0 = one tier per
											 batch, decided by its largest searched target cloud (rounds 1 - 3) */
	MULLS_OPT_BIG_EARLY_SETS = 22,		  /* [5; 2 until round 6: 32 scans against a 961 k-point map 5 940 -> 6 420 /s, 64 scans against 20 000-point maps unchanged] mixed batches: the first iterations run every global-memory-tier cloud as chunk-level jobs shared by several workgroups
											 (+ k_filter) — while most points still need a search that beats one workgroup per class cloud; from this iteration on the
											 down-sampled source clouds are class-level jobs (certificates, leftovers, rejection chain in one workgroup, no k_filter) */
	MULLS_OPT_KCERT = 23,				  /* [1] k-candidate certificates: a point whose hinted target fails the certificate evaluates the few nearest targets its last search
											 saw before it is searched again (0: round 4's certificates only).  In force on the global-memory tier (+4 ... 8 % there); the LDS tier's
											 kernels are built without them by default (keeping the records costs more than the look returns, lds_tier.h: MULLS_LDS_KCERT) */
	MULLS_OPT_KCERT_MIN = 24,			  /* [64] LDS tier, builds with MULLS_LDS_KCERT=1 only: leftover lists shorter than this skip the look (it costs one chain of round trips whatever the length) */
	MULLS_OPT_ACCUM_WAVE_MIN_TRIPS = 25,  /* [768; 2048 until round 6, when the kernel lost its memo and two thirds of its divisions: 512 pairs 185 -> 201 k/s, 768 pairs 211 -> 235 k/s] lock-step loop: from this many 1024-slot trips per launch on the normal equations are summed by one wave per trip (k_accum_wave:
											 the slots of a lane in sequence, the running sums in registers — no term buffer, no barriers; same bits; -0.26 ms of a 17.8 ms
											 step at 4096 pairs, profiles/r05_experiments.txt); 0 = always one workgroup per trip (k_accum).  Read when a batch is filled
											 and at every launch */
	MULLS_OPT_FIRST_DIRECT = 26,		  /* [1] LDS tier, lock-step loop: the setup applies iteration 0's rigid step (the identity) where it writes the cropped source clouds, and
											 iteration 0 runs no light pass — without hints it could only list every point and hand the class clouds over, or search a small
											 cloud unhinted against the grid in global memory: every called class cloud goes straight to the staged search.  0 = light pass
											 first, as in every other iteration.  Same bits */
	MULLS_OPT_SUM_STEP = 27,			  /* [1] lock-step loop stepped on the device, batches beyond STEP_LAUNCH_MAX_PAIRS: one wave per pair sums the pair's trip partials AND steps it
											 (k_sum_step) instead of k_finish followed by k_step — one launch less per iteration; pairs with more than MULLS_SUM_STEP_TRIPS
											 trips in a class keep the two kernels.  0 = k_finish + k_step.  Same bits */
	MULLS_OPT_COUNT = 28
};
int mulls_set_option(mulls_ctx *ctx, int option, double value);
int mulls_get_option(const mulls_ctx *ctx, int option, double *value);
/* raw hipStream_t the library launches on (so callers can bracket it with their own events) */
void *mulls_stream(mulls_ctx *ctx);

/* replaces one mm_lls_icp call: upload, register, release */
int mulls_icp(mulls_ctx *ctx, const mulls_pair *pair, const mulls_params *params, mulls_result *result);

/* n independent pairs advanced in lock-step (one launch set + one host sync per ICP iteration for the batch) */
int mulls_icp_batch(mulls_ctx *ctx, const mulls_pair *pairs, int n, const mulls_params *params, mulls_result *results);

/* ---- independent scan pairs over several contexts of one process ("one host thread + one HIP stream set per GPU") ----
 * The reference has one kind of caller: a C++ program that holds its clouds in host memory (test/mulls_slam.cpp).  These two forms give such a caller
 * what bench.py's torch.distributed launcher gives the benchmark, without a second process:
 *
 * mulls_icp_batch_sharded: the n pairs block-partitioned over n_ctx contexts — pair p goes to context floor(p * n_ctx / n), as mulls_amd/shard.py does —
 *   one per GPU of the node (mulls_create(device k)), or several on one GPU; the calling thread drives the first shard, one host thread each of the others;
 *   results[p] is pair p's, the very bits mulls_icp_batch returns on one context.  No collective: the shards share nothing.  Returns the first failing
 *   shard's code (its context has the text).  The contexts must be distinct and not in use by another thread. */
int mulls_icp_batch_sharded(mulls_ctx *const *ctxs, int n_ctx, const mulls_pair *pairs, int n, const mulls_params *params, mulls_result *results);

/* mulls_pipe: calls from host buffers in flight on `depth` alternating contexts of one device, so that call k + 1's host gather and PCIe upload run under
 *   call k's kernels (a serial mulls_icp_batch caller waits for 9 ms of staging in front of 3.7 ms of kernels per 1024 pairs).
 *   mulls_icp_batch_begin returns a ticket (>= 0) at once — or, when the lane it falls on (ticket mod depth) is still running its previous call, as soon as
 *   that call has finished; a negative value is an error code.  pairs, the clouds they point to and results belong to the call until mulls_icp_batch_end(ticket)
 *   has returned its code (MULLS_OK: results[] are written).  A ticket can be waited for until its lane is given another call.
 *   mulls_pipe_set_option applies a mulls_option to every lane (it waits for running calls); mulls_pipe_ctx hands out a lane's context (profile, error text). */
typedef struct mulls_pipe mulls_pipe;
int mulls_pipe_create(int device, int depth, mulls_pipe **out);
void mulls_pipe_destroy(mulls_pipe *pipe);
int mulls_pipe_depth(const mulls_pipe *pipe);
mulls_ctx *mulls_pipe_ctx(mulls_pipe *pipe, int lane);
int mulls_pipe_set_option(mulls_pipe *pipe, int option, double value);
int mulls_icp_batch_begin(mulls_pipe *pipe, const mulls_pair *pairs, int n, const mulls_params *params, mulls_result *results);
int mulls_icp_batch_end(mulls_pipe *pipe, int ticket);

/* the n result records as rows of 56 doubles — T (16, column-major), information matrix (36), code, iterations, sigma, confidence — the table bench.py's ranks
 * gather over RCCL (mulls_amd/shard.py: RECORD); host code, no context */
void mulls_pack_results(const mulls_result *results, int n, double *table);

/* device-resident form: stage once, run many times (each run re-clones the staged clouds like
 * cloudblock_t::clone_feature does, utility.hpp:524-550) */
int mulls_batch_create(mulls_ctx *ctx, const mulls_pair *pairs, int n, mulls_batch **out);
int mulls_batch_run(mulls_ctx *ctx, mulls_batch *batch, const mulls_params *params, mulls_result *results);
void mulls_batch_destroy(mulls_ctx *ctx, mulls_batch *batch);

/* ---- variants of the path ---- */

/* lls_icp_3dof_ground (cregistration.hpp:1443-1582): ground class only, unknowns (roll, pitch, z).  Reads the like-named
 * fields of mulls_params: max_iter_num, dis_thre_unit, converge_translation, converge_rotation_d, dis_thre_min,
 * dis_thre_update_rate, weight_strategy[1..3], keep_less_source_points (+ rng_seed), max_bearable_rotation_d (the
 * reference's default for it is 10).  Only result->T and result->code are outputs of this variant (the reference
 * returns the code cast to bool). */
int mulls_icp_3dof_ground(mulls_ctx *ctx, const mulls_pair *pair, const mulls_params *params, mulls_result *result);
int mulls_icp_3dof_ground_batch(mulls_ctx *ctx, const mulls_pair *pairs, int n, const mulls_params *params, mulls_result *results);

/* mm_lls_icp_4dof_global (cregistration.hpp:1584-1681): sweeps the heading of the source about `station`
 * (block2->local_station) in steps of heading_step_d degrees; every trial is a full mm_lls_icp with classes "111110" and
 * weights "1001", all trials run as one lock-step batch; the trial maximising confidence / sigma wins.  pair->init_guess
 * is ignored.  converge_rotation_d and max_bearable_rotation_d are accepted for signature fidelity; the reference does
 * not use them (it passes converge_translation twice).  result->iters returns the number of trials.  *success: 0 = no trial
 * succeeded (the reference returns false), 1 = `result` is the best succeeded trial, 2 = trials succeeded but none scored above 0 (a NaN
 * sigma): the reference returns true and leaves the constraint untouched — `result` is then a placeholder (identity, sigma FLT_MAX). */
int mulls_icp_4dof_global(mulls_ctx *ctx, const mulls_pair *pair, float heading_step_d, const double station[3], int max_iter_num,
						  float dis_thre_unit, float converge_translation, float converge_rotation_d, float dis_thre_min,
						  float dis_thre_update_rate, float max_bearable_rotation_d, mulls_result *result, int *success,
						  float *best_heading_d);

/* ---- device-resident local map ----
 * MapManager::update_local_map (src/map_manager.cpp:18-140) with the six undown class clouds of `local_map` kept in HBM
 * between frames: the scan-to-map target is never re-uploaded (mulls_map_cloud() yields device clouds that mulls_pair.tgt
 * accepts), and map-based dynamic-object removal (map_manager.cpp:149-256) runs as an exact nearest-neighbour pass on the
 * device instead of querying the kd-trees mm_lls_icp left on block1. */
typedef struct mulls_map mulls_map;

typedef struct mulls_map_params
{
	/* the positional arguments of update_local_map after the two blocks, in order (map_manager.h:21-31) */
	float local_map_radius;			   /* 80 */
	int max_num_pts;				   /* 20000 */
	int kept_vertex_num;			   /* 800 */
	float last_frame_reliable_radius;  /* 60; unused by the reference too */
	int map_based_dynamic_removal_on;  /* false */
	char used_feature_type[8];		   /* "111110" */
	float dynamic_removal_center_radius; /* 30.0 */
	float dynamic_dist_thre_min;	   /* 0.3 */
	float dynamic_dist_thre_max;	   /* 3.0 */
	float near_dist_thre;			   /* 0.03 */
	int recalculate_feature_on;		   /* true: principal directions of the map's pillar / beam points recomputed from their own
										  neighbourhoods, points that are not linear / steep / flat enough dropped (map_manager.cpp:98-118, :258-292) */
	/* not in the reference's signature */
	uint64_t rng_seed; /* random_downsample_pcl: the ABI's seeded selection sampling (see mulls_params.rng_seed) */
	int tree_mode;	   /* what block1->tree_* held after the last mm_lls_icp against this map (cregistration.hpp:1209-1232):
						* 0 = nothing (dynamic removal is skipped), 1 = the whole class clouds, 2 = the class clouds cropped to
						* tree_box (mulls_result.crop_box of that registration) */
	char tree_used[8]; /* used_feature_type of that registration: a class without a tree is left alone */
	double tree_box[6];
} mulls_map_params;

typedef struct mulls_map_report
{
	uint32_t n[6];			/* class cloud sizes of the map after the update */
	uint32_t frame_n[6];	/* sizes of the frame's clouds as appended (pc_*_down after dynamic removal; [5] = pc_vertex) */
	int feature_point_num;	/* ground + facade + roof + pillar + beam */
	int dynamic_removal_ran;
	double local_bound[6];	/* local_map->local_bound (min xyz, max xyz) */
	double bound[6];		/* local_map->bound: the same points moved by pose_lo */
	float ms_total;
} mulls_map_report;

void mulls_map_default_params(mulls_map_params *p);
int mulls_map_create(mulls_ctx *ctx, mulls_map **out);
/* a map belongs to its context: mulls_destroy(ctx) also destroys the maps still alive (their handles become invalid) */
void mulls_map_destroy(mulls_ctx *ctx, mulls_map *map);
/* (re)initialise the map from host clouds (e.g. the first frame's undown features) and its pose_lo (column-major) */
int mulls_map_set(mulls_ctx *ctx, mulls_map *map, const mulls_cloud clouds[6], const double pose_lo[16]);
/* update_local_map(local_map = map, last_target_cblock = {frame_down, frame_pose_lo}).  frame_down[0..4] = pc_*_down,
 * frame_down[5] = pc_vertex, all in the frame's own coordinates; they are not modified (the reference transforms and
 * filters last_target_cblock->pc_*_down in place: mulls_map_frame_download returns that state). */
int mulls_map_update(mulls_ctx *ctx, mulls_map *map, const mulls_cloud frame_down[6], const double frame_pose_lo[16],
					 const mulls_map_params *params, mulls_map_report *report);
/* class cloud c of the map as a device cloud (48-B records); valid until the next mulls_map_set / _update / _destroy */
int mulls_map_cloud(mulls_ctx *ctx, const mulls_map *map, int cls, mulls_cloud *out);
int mulls_map_pose(mulls_ctx *ctx, const mulls_map *map, double pose_lo[16]);
/* copy class cloud c back to the host (48-B records); *n receives its size, at most cap records are written */
int mulls_map_download(mulls_ctx *ctx, const mulls_map *map, int cls, void *pts, uint32_t cap, uint32_t *n);
int mulls_map_frame_download(mulls_ctx *ctx, const mulls_map *map, int cls, void *pts, uint32_t cap, uint32_t *n);

/* ---- on-disk formats either side of the path; host code, no context needed ----
 * Readers follow the two-call pattern: *n receives the number of points, at most `cap` 48-byte records are written. */
/* DataIo::read_bin_file (dataio.hpp:357-378): KITTI x y z intensity float32 quadruples, intensity * 255; like the
 * reference the cloud ends with one extra all-zero point (its read loop tests the stream after pushing). */
int mulls_io_read_kitti_bin(const char *path, void *pts, uint32_t cap, uint32_t *n);
/* DataIo::read_pcd_file (dataio.hpp:279-287) = pcl::io::loadPCDFile<PointXYZINormal>: PCD v0.7, DATA ascii or binary,
 * float32 fields matched by name (x y z intensity normal_x normal_y normal_z curvature), others ignored / left 0. */
int mulls_io_read_pcd(const char *path, void *pts, uint32_t cap, uint32_t *n);
/* DataIo::write_pcd_file (dataio.hpp:288-312): WIDTH 1, HEIGHT n, eight float32 fields, binary or ascii */
int mulls_io_write_pcd(const char *path, const void *pts, uint32_t n, uint32_t stride, int as_binary);
/* DataIo::write_lo_pose_overwrite / _append (dataio.hpp:1896-1926): the top three rows of T (column-major), 8 significant digits */
int mulls_io_write_pose(const char *path, const double T[16], int append);

/* ---- feature extraction, first stage: CFilter::fast_ground_filter (include/common/cfilter.hpp:1658-2036),
 * the two-threshold grid ground filter that splits a (voxel-down-sampled) scan into ground and non-ground points ---- */

/* its positional parameters (cfilter.hpp:1663-1672; values of script/config/lo_gflag_list_kitti_urban.txt in brackets) */
typedef struct mulls_ground_params
{
	int32_t min_grid_pt_num;			   /* gf_grid_min_pt_num [6] */
	float grid_resolution;				   /* gf_grid_size [2.5] */
	float max_height_difference;		   /* gf_in_grid_h_thre [0.25] */
	float neighbor_height_diff;			   /* gf_neigh_grid_h_thre [1.5] */
	float max_ground_height;			   /* gf_max_h [2.0] */
	int32_t ground_random_down_rate;	   /* gf_ground_down_rate [12] */
	int32_t ground_random_down_down_rate;  /* gf_down_down_rate [3] */
	int32_t nonground_random_down_rate;	   /* gf_nonground_down_rate [3] */
	int32_t reliable_neighbor_grid_num_thre; /* gf_reliable_neighbor_grid_thre [0] */
	int32_t estimate_ground_normal_method; /* 0: (0,0,1).  1 / 2: pcl::NormalEstimationOMP over cloud_ground with every neighbour within normal_estimation_radius /
											  with the 2 * min_grid_pt_num nearest (pca.hpp:66-119), non-finite normals replaced by 0.577 (check_normal, :462-475).
											  3 [the shipped configs and extract_semantic_pts' default]: one PCL plane RANSAC per grid cell (cfilter.hpp:1909,
											  :2038-2056 -> cprocessing.hpp:67-106: SACSegmentation, 20 iterations, threshold 0.3 * max_height_difference, refit to the
											  inliers); the cell's ground points are the refined plane's inliers, every ground_random_down_rate-th of them kept with the
											  plane's normal if abs(normal_z) > 0.8.  PCL is not in this image: what it computes inside these calls is restated —
											  PCL's own deterministic sample sequence (boost::mt19937 seeded 12345, draw = output / 2, partial shuffles carried from
											  draw to draw), its float expressions, the neighbours of methods 1 / 2 ascending by (distance, index); the plane through
											  the inliers / neighbours is the smallest eigenvector of PCL's float covariance by Jacobi rotations in double instead of
											  pcl::eigen33's closed form (its largest component positive for method 3; turned towards the sensor for 1 / 2, as PCL
											  does) — "parity unpinned" for that part, everything MULLS wrote around the calls is pinned */
	int32_t distance_weight_downsampling_method; /* dist_inverse_sampling_method: 0 off, 1 linear, 2 quadratic [2].  Upstream the per-cell rates of 1 / 2
											  go through a variable shared by the threads of an OpenMP loop (cfilter.hpp:1829-1840: a data race); here
											  every cell uses its own value, i.e. the loop's sequential semantics */
	float standard_distance;			   /* [15.0] */
	uint8_t fixed_num_downsampling;		   /* ground_down by a fixed number instead of every down_down_rate-th point */
	uint8_t apply_grid_wise_outlier_filter; /* extract_semantic_pts passes apply_scanner_filter here (cfilter.hpp:2361) */
	uint8_t reserved_[2];
	int32_t down_ground_fixed_num;		   /* ground_down_fixed_num [800] */
	float intensity_thre;				   /* intensity_thre_nonground [150]; FLT_MAX disables */
	float outlier_std_scale;			   /* 3.0 */
	float normal_estimation_radius;		   /* [2.0] estimate_ground_normal_method 1 only (cfilter.hpp:1669) */
	uint32_t reserved2_;
	uint64_t rng_seed;					   /* ABI-only: fixed_num_downsampling thins with the seeded order-preserving selection of mulls_params.rng_seed
											  (upstream: pcl::RandomSample seeded with time(NULL)) */
} mulls_ground_params;

void mulls_ground_default_params(mulls_ground_params *p);

/* fast_ground_filter on `n` points (48-byte records, `stride` bytes apart).  Outputs (48-byte records, host memory, capacities in
 * points): ground = cloud_ground (normal (0,0,1)), ground_down = cloud_ground_down, unground = cloud_unground with data[3] (the
 * float at byte offset 12) = height above ground as the reference stores it.  n_out[3] = the three sizes; a cloud larger than its
 * capacity is truncated to it (its size is still reported).  The input is not modified (upstream writes normals and data[3] into
 * cloud_in: the copies handed out carry them).  One workgroup per scan; grids of more than 65536 cells or scans of more than 500000 points -> MULLS_E_UNSUPPORTED. */
int mulls_ground_filter(mulls_ctx *ctx, const void *pts, uint32_t n, uint32_t stride, const mulls_ground_params *params, void *ground, uint32_t cap_ground,
						void *ground_down, uint32_t cap_ground_down, void *unground, uint32_t cap_unground, uint32_t n_out[3]);

/* ---- feature extraction, second stage: CFilter::classify_nground_pts (include/common/cfilter.hpp:2058-2290) ---- */

/* its positional parameters (cfilter.hpp:2068-2082); defaults = what extract_semantic_pts (:2301-2318) passes when test/mulls_reg.cpp
 * calls it with script/run_mulls_reg.sh's flags */
typedef struct mulls_classify_params
{
	float neighbor_searching_radius; /* pca_neighbor_radius [1.0] */
	int32_t neighbor_k;				 /* pca_neighbor_count [50]; at most 64 */
	int32_t neigh_k_min;			 /* pca_neighbor_k_min [8] */
	int32_t pca_down_rate;			 /* [1]: every pca_down_rate-th point is a query, all are neighbours */
	float edge_thre, planar_thre;	 /* linearity_thre, planarity_thre [0.65, 0.65] */
	float edge_thre_down, planar_thre_down; /* [0.75, 0.75]; only read when sharpen_with_nms = 0 */
	int32_t extract_vertex_points_method;	/* [2]; 0 = no promotion of high-curvature points to pillar / beam */
	float curvature_thre;					/* [0.10] */
	float vertex_curvature_non_max_radius;	/* unused upstream too (1.5 * radius) */
	float linear_vertical_sin_high_thre, linear_vertical_sin_low_thre; /* [0.94, 0.17]: pillar above, beam below */
	float planar_vertical_sin_high_thre, planar_vertical_sin_low_thre; /* [0.98, 0.34]: roof above, facade below */
	uint8_t fixed_num_downsampling;			/* [0] */
	uint8_t sharpen_with_nms;				/* [1] */
	uint8_t use_distance_adaptive_pca;		/* [0] */
	uint8_t reserved_;
	int32_t pillar_down_fixed_num, facade_down_fixed_num, beam_down_fixed_num, roof_down_fixed_num, unground_down_fixed_num; /* [200, 800, 200, 200, 20000] */
	float beam_height_max, roof_height_min; /* [FLT_MAX, 0.0] */
	float feature_pts_ratio_guess;			/* [0.3] */
	uint64_t rng_seed;						/* ABI-only: the fixed-number down-samplings use the seeded order-preserving selection of mulls_params.rng_seed */
} mulls_classify_params;

void mulls_classify_default_params(mulls_classify_params *p);

enum mulls_classify_cloud
{
	MULLS_CL_PILLAR = 0,
	MULLS_CL_BEAM = 1,
	MULLS_CL_FACADE = 2,
	MULLS_CL_ROOF = 3,
	MULLS_CL_PILLAR_DOWN = 4,
	MULLS_CL_BEAM_DOWN = 5,
	MULLS_CL_FACADE_DOWN = 6,
	MULLS_CL_ROOF_DOWN = 7,
	MULLS_CL_VERTEX = 8, /* the key points this call appends to cloud_vertex */
	MULLS_CL_COUNT = 9
};

/* classify_nground_pts on the `n` non-ground points of a scan (48-byte records, `stride` bytes apart; normally mulls_ground_filter's
 * `unground`).  out[k] / cap[k] / n_out[k], k = enum mulls_classify_cloud: host buffers of 48-byte records, capacities in points, sizes; a
 * cloud larger than its capacity is truncated to it (its size is still reported); out[k] may be NULL with cap[k] = 0.
 * cloud_in_after (NULL, or room for n records) / n_cloud_in_after (NULL or a count): cloud_in as upstream leaves it — the function writes the
 * estimated normals into the cloud it is given, and thins it first when fixed_num_downsampling is on.  `pts` itself is not modified.
 * Neighbourhoods are exact (the neighbor_k nearest within the radius, by (distance, index)); what pcl::PCA / Eigen compute is restated (float
 * covariance in neighbour order, Jacobi in double; a direction's largest component is positive) — see DESIGN.md section 11 for what that means
 * for points within ~1e-6 of a threshold.  non_max_suppress's visiting order (and the order the class clouds are left in) is std::sort's by
 * normal[3] descending, ties included: the keys are sorted by the host's std::sort, the one order upstream's own build would produce. */
int mulls_classify_nground(mulls_ctx *ctx, const void *pts, uint32_t n, uint32_t stride, const mulls_classify_params *params, void *const out[MULLS_CL_COUNT],
						   const uint32_t cap[MULLS_CL_COUNT], uint32_t n_out[MULLS_CL_COUNT], void *cloud_in_after, uint32_t *n_cloud_in_after);

/* ---- feature extraction, the whole chain: CFilter::extract_semantic_pts (include/common/cfilter.hpp:2294-2413) ---- */

/* The per-frame front end of test/mulls_slam.cpp:359-365 / :404-421 in one call: dist_filter (:806-832, --apply_dist_filter) -> scanner filter
 * (:2338-2346, :914-929) -> voxel_downsample (:83-160, off below 0.001 m as in every shipped configuration: cloud_down_res 0) ->
 * fast_ground_filter -> classify_nground_pts.  The scan goes up once and the clouds stay on the device between the stages.  Not part of it: the
 * semantic-mask filters (semantic_assisted), the adaptive parameter update (host arithmetic on the cloud sizes, :2416-2444). */
typedef struct mulls_extract_params
{
	mulls_ground_params ground;
	mulls_classify_params classify;
	uint8_t apply_scanner_filter; /* extract_semantic_pts passes the same flag to fast_ground_filter as apply_grid_wise_outlier_filter: set ground.* yourself */
	uint8_t apply_dist_filter;	  /* keep min_dist_used^2 < x^2 + y^2 < max_dist_used^2 (float range, double limits), ahead of everything else */
	uint8_t reserved_[2];
	float self_ring_radius;			/* [1.75] */
	float ghost_radius;				/* [20.0] */
	float z_min;					/* -approx_scanner_height - 4.0: ghost points below it within ghost_radius go */
	float z_min_min;				/* -approx_scanner_height + underground_thre: everything below it goes */
	float vf_downsample_resolution; /* [0.0] voxel edge in metres; < 0.001: pc_down = pc_raw */
	double min_dist_used;			/* [1.0]  */
	double max_dist_used;			/* [120.0] */
} mulls_extract_params;

void mulls_extract_default_params(mulls_extract_params *p);

enum mulls_extract_cloud
{
	MULLS_EX_RAW = 0,		  /* pc_raw after the distance and scanner filters */
	MULLS_EX_GROUND = 1,	  /* pc_ground */
	MULLS_EX_GROUND_DOWN = 2, /* pc_ground_down */
	MULLS_EX_UNGROUND = 3,	  /* pc_unground as classify_nground_pts leaves it (written only if its capacity holds the whole cloud) */
	MULLS_EX_PILLAR = 4,	  /* ... followed by the nine clouds of enum mulls_classify_cloud in that order */
	MULLS_EX_VERTEX = 12,
	MULLS_EX_DOWN = 13, /* pc_down: pc_raw after voxel_downsample (the same cloud when that is off; ask for it with a capacity only when it is on) */
	MULLS_EX_COUNT = 14
};

/* out[k] / cap[k] / n_out[k], k = enum mulls_extract_cloud: host buffers of 48-byte records, capacities in points, sizes (a cloud larger than its
 * capacity is truncated, its size still reported; out[k] may be NULL with cap[k] = 0).  Results are those of mulls_ground_filter followed by
 * mulls_classify_nground on its `unground`. */
int mulls_extract_features(mulls_ctx *ctx, const void *scan, uint32_t n, uint32_t stride, const mulls_extract_params *params, void *const out[MULLS_EX_COUNT],
						   const uint32_t cap[MULLS_EX_COUNT], uint32_t n_out[MULLS_EX_COUNT]);

/* ---- device-resident feature block: the cloudblock_t of one scan kept in HBM ----
 * The per-frame loop of test/mulls_slam.cpp (:359-442, :642-693) hands one cloudblock_t from extract_semantic_pts to mm_lls_icp to update_local_map.
 * mulls_extract_features_resident is mulls_extract_features with the feature clouds left on the device: mulls_block_cloud() yields device clouds
 * (48-byte records) that mulls_pair.src[] / tgt[] and mulls_map_update's frame_down[] accept the way they accept mulls_map_cloud() — raw scan in, pose
 * out, nothing but the scan, a few selection indices and the 4x4 crossing PCIe (the *_down clouds the fixed-number samplers thin on the host — a few
 * thousand points — make one round trip when fixed_num_downsampling is on).  A block holds the clouds of enum mulls_extract_cloud except MULLS_EX_RAW and
 * MULLS_EX_DOWN; they are valid until the block is extracted into again or destroyed (mulls_destroy destroys the blocks still alive). */
typedef struct mulls_block mulls_block;
int mulls_block_create(mulls_ctx *ctx, mulls_block **out);
void mulls_block_destroy(mulls_ctx *ctx, mulls_block *block);
int mulls_extract_features_resident(mulls_ctx *ctx, const void *scan, uint32_t n, uint32_t stride, const mulls_extract_params *params, mulls_block *block,
									uint32_t n_out[MULLS_EX_COUNT]);
int mulls_block_cloud(mulls_ctx *ctx, const mulls_block *block, int which, mulls_cloud *out);
/* copy cloud `which` back to the host (48-byte records); *n receives its size, at most cap records are written */
int mulls_block_download(mulls_ctx *ctx, const mulls_block *block, int which, void *pts, uint32_t cap, uint32_t *n);

/* ---- motion compensation of a frame after its registration (test/mulls_slam.cpp:703-712, on in the 32- and 128-beam configurations) ----
 * mulls_motion_compensate = CFilter::apply_motion_compensation(pc_in_out, Tran, s_ambigous_thre) (cfilter.hpp:470-491), in place on `n` 48-byte records: every
 * point whose time stamp t (the curvature field, pts time-stamp ratio in the frame) lies in [thre, 1 - thre] is moved by the fraction t of Tran — slerp of the
 * rotation from the identity, t times the translation, double arithmetic, float store; directions are left as they are.  Tran: 4 x 4, column-major.
 * pts: host memory (one upload, one download) or a device-resident cloud (mulls_block_cloud, mulls_map_cloud: in place, nothing crosses PCIe).
 * mulls_block_motion_compensate = the two batch_apply_motion_compensation calls of mulls_slam.cpp:706-710 on a device-resident feature block: its ground /
 * pillar / facade / beam / roof clouds and their *_down clouds (the vertex cloud only with undistort_keypoints, which no caller of the reference sets). */
int mulls_motion_compensate(mulls_ctx *ctx, void *pts, uint32_t n, uint32_t stride, const double Tran[16], float s_ambiguous_thre);
int mulls_block_motion_compensate(mulls_ctx *ctx, mulls_block *block, const double Tran[16], int undistort_keypoints);

/* CFilter::voxel_downsample (cfilter.hpp:83-160): one point per occupied voxel of edge voxel_size, voxels in increasing index
 * ((vx * ny + vy) * nz + vz from the cloud's minimum corner), the point of a voxel being the one std::sort leaves first among that voxel's
 * (voxel, index) pairs, exactly as upstream (bounding box and voxel indices on the device, that one sort on the host).  voxel_size < 0.001
 * copies the cloud.  MULLS_E_INVALID for a non-finite coordinate, MULLS_E_UNSUPPORTED beyond 2^21 voxels along an axis or 500000 points.
 * out: host buffer of cap 48-byte records; *n_out = size of pc_down (truncated to cap when larger). */
int mulls_voxel_downsample(mulls_ctx *ctx, const void *pts, uint32_t n, uint32_t stride, float voxel_size, void *out, uint32_t cap, uint32_t *n_out);

/* ---- stage-level entry points (used by the parity tests; same kernels the driver launches) ---- */

/* batch_transform_feature_points (cregistration.hpp:1685-1696): in place on a host cloud via the device kernel */
int mulls_stage_transform(mulls_ctx *ctx, void *pts, uint32_t n, uint32_t stride, const double T[16]);

/* determine_corres (cregistration.hpp:1701-1835) on already-transformed clouds.
 * Outputs, each sized n_src: match[s] = target index or -1, d2[s] = squared NN distance (float),
 * flags[s] bit0 = survives compaction (always 1 when n_src < 500), bit1 = final correspondence. */
int mulls_stage_correspond(mulls_ctx *ctx, const mulls_cloud *src, const mulls_cloud *tgt, float dis_thre,
						   int normal_check, float angle_thre_degree, int32_t *match, float *d2, uint8_t *flags);

/* one class' contribution to ATPA (21 packed upper/lower terms, row-major-upper enumeration) and ATPb (6):
 * metric 0 = pt2pl (:2066-2156), 1 = pt2li (:2160-2275), 2 = pt2pt (:1976-2063).  out27 = 21 + 6 doubles;
 * weight_out[k] (may be NULL) = value the reference leaves in pcl::Correspondence::weight. */
int mulls_stage_accumulate(mulls_ctx *ctx, int metric, const mulls_cloud *src, const mulls_cloud *tgt,
						   const int32_t *corr_src, const int32_t *corr_tgt, const float *corr_d2, uint32_t ncorr,
						   int iter_num, float class_weight, int dist_w, int resid_w, int inten_w, float window,
						   double *out27, float *weight_out);

#ifdef __cplusplus
}
#endif
#endif /* MULLS_HIP_H */
