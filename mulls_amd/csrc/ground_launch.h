// ground_launch.h — host-callable launchers of k_ground.hip (CFilter::fast_ground_filter and the per-point filters / voxel down-sampling ahead of
// it on the device)
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/mulls_hip.h"

struct GfOut // head of the device-side state (k_ground.hip: GfState)
{
	uint32_t n_ground, n_unground, n_high, error;
	uint32_t row, col;
	float mean_height;
	uint32_t n_cand;
};
// scratch of estimate_ground_normal_method 3 (the per-cell plane RANSAC, k_gf_ransac); all null for the other methods
struct GfRansac
{
	uint32_t *gg;		 // [n] the cells' grid_ground members (sorted-entry indices), cell by cell
	uint32_t *perm;		 // [n] shuffled_indices_ of the cells' sample consensus models
	uint32_t *inl;		 // [n] inlier lists
	float4 *gxyz;		 // [n] the members' x y z data[3]
	float4 *cell_nrm;	 // [MULLS_GF_MAXCELLS] the refined plane's normal per cell
	const uint32_t *rnd; // [MULLS_GF_RND] PCL's sample sequence: boost::mt19937(12345u) outputs / 2
};
#define MULLS_GF_RND 63008u // draws one cell can take at most: 21 iterations x 1000 sample checks x 3 (+ padding)
int launch_ground_filter(hipStream_t st, const float4 *pts, uint32_t n, const mulls_ground_params &P, uint32_t *ids, uint16_t *cellof, uint8_t *code, float *d3v,
						 float4 *ground, float4 *unground, void *aux, const GfRansac &R);
size_t ground_filter_aux_bytes(uint32_t n);
#define MULLS_GF_MAXCELLS 65536u

// estimate_ground_normal_method 1 / 2: pcl::NormalEstimationOMP over the n_ground records at `ground` (radius > 0: every neighbour within it,
// else the k nearest), check_normal's 0.577 where fewer than 3 neighbours exist.  grid_mem: ground_normals_bytes(n_ground) bytes of scratch.
// *error (device word) |= 4 when a neighbourhood exceeds the 1024 entries the kernel buffers.
size_t ground_normals_bytes(uint32_t n_ground);
void launch_ground_normals(hipStream_t st, float4 *ground, uint32_t n_ground, float radius, int k, void *grid_mem, uint32_t *error);

struct RawMaskArgs // dist_filter (cfilter.hpp:806-832) and scanner_filter (cfilter.hpp:914-929) as one keep mask
{
	int32_t dist_on, scanner_on;
	double dist_min_sq, dist_max_sq; // xy_dist_min * xy_dist_min, xy_dist_max * xy_dist_max as the reference forms them (doubles)
	float self_radius, ghost_radius, z_min_ghost, z_min_global;
};
void launch_raw_mask(hipStream_t st, const float4 *pts, uint32_t n, const RawMaskArgs &a, uint8_t *mask);

// voxel_downsample (cfilter.hpp:83-160): box = 7 device words (ordered keys of min xyz, max xyz; non-finite flag), keys[i] = voxel index of point i
int launch_vox_bbox(hipStream_t st, const float4 *pts, uint32_t n, uint32_t *box);
void launch_vox_keys(hipStream_t st, const float4 *pts, uint32_t n, const float min_p[3], float inverse_voxel_size, unsigned long long mul_vx,
					 unsigned long long mul_vy, unsigned long long *keys);
