// classify_launch.h — host-callable launchers of k_classify.hip (CFilter::classify_nground_pts on the device)
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include <stdint.h>

#include "../../include/mulls_hip.h"

#define MULLS_CL_MAX_K 64u			// neighbor_k limit (list slots per query lane)
#define MULLS_CL_MAX_CELLS (1u << 22) // cells of the search grid; a finer grid is coarsened by powers of two

// device-resident description of the search grid (written by k_cl_setup)
struct ClGrid
{
	uint32_t keys[6]; // ordered-uint min xyz, max xyz
	uint32_t nonfinite;
	uint32_t dim[3];
	uint32_t ncell;
	float cell;
	float lo[3];
};

// every device array of one call (SoA, n = points after the optional fixed-number thinning; K = neighbor_k)
struct ClArrays
{
	float4 *recs;	 // [3n] work copy of cloud_in: normals are written into it as the reference writes them
	float4 *sorted;	 // [n] x y z | original index, cell-sorted
	uint32_t *cellof; // [n]
	uint32_t *cell_start, *cell_fill; // [MAX_CELLS + 1]
	uint32_t *seg_sum;				  // [1024 + 1]
	uint32_t *nbr;					  // [K][n] neighbour indices, nearest first
	unsigned long long *closebits;	  // [n] bit k: neighbour k is closer than sqrt(0.64) * radius
	int32_t *f_cnt;					  // [n] pt_num
	float *cov;						  // [6][n] xx xy xz yy yz zz of the neighbourhood's covariance
	double *f_curv, *f_lin, *f_pla;	  // [n] curvature, linear_2, planar_2
	float4 *f_pd, *f_nd;			  // [n] principal / normal direction
	uint8_t *lab, *plab, *cstate, *cand, *down; // [n] each
	uint8_t *mask;								// [11][n]
	float4 *vtx;								// [3n] key-point records by input index
	uint32_t *round_cnt;						// [64] undecided candidates left after promotion round r
	ClGrid *grid;
};

struct ClParams // classify parameters as the kernels use them
{
	uint32_t n, K;
	int32_t down_rate, k_min;
	float radius;
	int32_t adaptive;
	float unit_distance;
	float edge_thre, planar_thre, edge_thre_down, planar_thre_down;
	float lin_high, lin_low, pla_high, pla_low;
	float beam_height_max, roof_height_min;
	int32_t nms, vertex_method;
	float curvature_thre, vertex_ratio_thre, min_curvature;
	int32_t min_neighbor_feature_pts;
};

void launch_cl_grid(hipStream_t st, const ClArrays &A, const ClParams &P);
void launch_cl_pca(hipStream_t st, const ClArrays &A, const ClParams &P);
void launch_cl_label(hipStream_t st, const ClArrays &A, const ClParams &P);
void launch_cl_promote_round(hipStream_t st, const ClArrays &A, const ClParams &P, uint32_t round);
void launch_cl_encode_and_masks(hipStream_t st, const ClArrays &A, const ClParams &P);

#define MULLS_CL_NMS_CAP 32u
struct ClNmsArgs
{
	const float4 *recs[4]; // class clouds in visiting order
	uint32_t n[4];		   // 0 = skip
	uint8_t *keep[4];	   // state while the rounds run, keep mask once they have settled
	uint32_t *list[4];	   // [MULLS_CL_NMS_CAP][n[c]] earlier neighbours within the radius
	uint32_t *cnt[4];	   // [n[c]] how many there are (may exceed the list)
	uint32_t *off[4];	   // [n[c]] where a longer list starts in the pool, ~0 if it did not fit
	uint32_t *wcur[4];	   // [n[c]] fill cursor of a pool list
	uint32_t *pool;				// pool of pool_cap entries shared by the four classes
	unsigned long long *pool_used; // entries requested so far; 0 before launch_cl_nms_lists
	uint32_t pool_cap;
	float r2;
};
void launch_cl_nms_lists(hipStream_t st, const ClNmsArgs &a, uint32_t *round_cnt); // round_cnt: 64 counters, zeroed here with the lists' own arrays
void launch_cl_nms_round(hipStream_t st, const ClNmsArgs &a, uint32_t *round_cnt); // *round_cnt += points still undecided
// keys[i] = normal[3] of record i
void launch_cl_keys(hipStream_t st, const float4 *recs, uint32_t n, float *keys);
// out[i] = in[perm[i]], whole 48-byte records
void launch_cl_gather(hipStream_t st, const float4 *in, const uint32_t *perm, float4 *out, uint32_t n);
