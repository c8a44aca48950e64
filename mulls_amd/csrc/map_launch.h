// map_launch.h — host-callable launchers of map_kernels.hip (device-resident local map)
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include <stdint.h>

// one class cloud of 48-B records for the multi-cloud kernels (blockIdx.x = slot)
struct MapCloudArg
{
	const float4 *in; // 3 float4 per record
	float4 *out;
	const uint8_t *mask; // compaction by mask (mode 0); unused otherwise
	uint32_t n;
	uint32_t pad_;
};
struct MapCompactArgs
{
	MapCloudArg cloud[6];
	uint32_t *out_n; // [6]
	int mode;		 // 0 = keep where mask != 0; 1 = CFilter::dist_filter(cloud, radius) (cfilter.hpp:834-871)
	double radius;
};
struct MapBoxArgs
{
	const float4 *recs[6];
	uint32_t n[6];
	double pose[12]; // rows of [R|t] (local_map->pose_lo)
	uint32_t *keys;	 // [12] ordered-uint min xyz / max xyz of the points, then of the posed points
};

// seg_scratch: device array of at least map_compact_segments(a) uint32 (one counter per 4096-record segment)
uint32_t map_compact_segments(const MapCompactArgs &a);
void launch_map_compact(hipStream_t st, const MapCompactArgs &a, uint32_t *seg_scratch);
void launch_map_bbox(hipStream_t st, const MapBoxArgs &a);
// nearest tree point of every frame point: best[i] = float bits of the smallest squared distance (0x7f800000 if the tree
// is empty); tree = map class cloud, optionally restricted to the open box (strict inequalities, float vs double)
void launch_map_nn(hipStream_t st, const float4 *frame, uint32_t n_frame, const float4 *tree, uint32_t n_tree, int use_box, const double box[6],
				   uint32_t *best);
void launch_map_keep(hipStream_t st, const float4 *frame, uint32_t n_frame, const uint32_t *best, float center_radius, float dmin, float dmax,
					 float near, uint8_t *keep);

// PCA refresh of a linear-feature cloud (MapManager::update_cloud_vectors): writes direction / linearity into the records of
// the points it keeps and keep[i] = 0/1 for every point; max_k <= 24
struct MapPcaArgs
{
	float4 *recs;
	uint32_t n;
	float radius;
	int max_k, min_k;
	float sin_low, sin_high, min_linearity;
	uint8_t *keep;
};
void launch_map_pca(hipStream_t st, const MapPcaArgs &a);
