// map_kernels.hip — device side of the local map (MapManager::update_local_map, src/map_manager.cpp:18-140; dynamic
// object removal :149-268).  Clouds stay 48-B PointXYZINormal records (3 float4: x y z -, nx ny nz -, intensity
// curvature - -) so that a map class cloud can be handed to mulls_icp as a device-resident target unchanged.
#include <hip/hip_runtime.h>

#include "map_launch.h"
#include "pca_device.h"

namespace
{
__device__ __forceinline__ uint32_t float_ord(float f)
{
	const uint32_t u = __float_as_uint(f);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
} // namespace

// Stable compaction of up to six clouds in three steps, so that a 400 k-point class cloud is not left to one CU's memory
// bandwidth: (1) every 256-lane workgroup counts the survivors of one 4096-record segment, (2) one workgroup per cloud
// turns the segment counts into segment bases (and the cloud's new size), (3) every workgroup rewrites its segment behind
// its base.  The order of the survivors is the input order (the reference pushes them back one by one).
#define MAP_SEG 4096u
namespace
{
__device__ __forceinline__ bool map_keep(const MapCompactArgs &a, const MapCloudArg &c, uint32_t i, const float4 &r0, double r2)
{
	if (a.mode == 0)
		return c.mask[i] != 0;
	const double dis_square = (double)(r0.x * r0.x + r0.y * r0.y); // float products and sum, then widened (cfilter.hpp:845)
	return dis_square < r2 && r0.z < 1.7976931348623157e308 && r0.z > -1.7976931348623157e308;
}
// which cloud and which of its segments a workgroup owns
__device__ __forceinline__ bool map_segment(const MapCompactArgs &a, uint32_t b, uint32_t &cloud, uint32_t &seg)
{
	for (cloud = 0; cloud < 6; cloud++)
	{
		const uint32_t ns = (a.cloud[cloud].n + MAP_SEG - 1u) / MAP_SEG;
		if (b < ns)
		{
			seg = b;
			return true;
		}
		b -= ns;
	}
	return false;
}
} // namespace

__global__ __launch_bounds__(256) void k_map_seg_count(MapCompactArgs a, uint32_t *__restrict__ seg_cnt)
{
	__shared__ uint32_t wave_cnt[4];
	uint32_t cloud, seg;
	if (!map_segment(a, blockIdx.x, cloud, seg))
		return;
	const MapCloudArg c = a.cloud[cloud];
	const double r2 = a.radius * a.radius;
	uint32_t mine = 0;
	for (uint32_t k = 0; k < MAP_SEG; k += 256)
	{
		const uint32_t i = seg * MAP_SEG + k + threadIdx.x;
		if (i < c.n)
			mine += map_keep(a, c, i, c.in[(size_t)i * 3], r2) ? 1u : 0u;
	}
	for (int off = 32; off > 0; off >>= 1)
		mine += __shfl_down(mine, off);
	if ((threadIdx.x & 63) == 0)
		wave_cnt[threadIdx.x >> 6] = mine;
	__syncthreads();
	if (threadIdx.x == 0)
		seg_cnt[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

// exclusive scan of each cloud's segment counts (in place), cloud totals to out_n; segments per cloud are few (<= a few hundred)
__global__ __launch_bounds__(64) void k_map_seg_scan(MapCompactArgs a, uint32_t *__restrict__ seg_cnt)
{
	uint32_t first = 0;
	for (uint32_t c = 0; c < blockIdx.x; c++)
		first += (a.cloud[c].n + MAP_SEG - 1u) / MAP_SEG;
	const uint32_t ns = (a.cloud[blockIdx.x].n + MAP_SEG - 1u) / MAP_SEG;
	uint32_t running = 0;
	for (uint32_t base = 0; base < ns; base += 64)
	{
		const uint32_t s = base + threadIdx.x;
		const uint32_t v = s < ns ? seg_cnt[first + s] : 0u;
		uint32_t incl = v;
		for (int off = 1; off < 64; off <<= 1)
		{
			const uint32_t o = __shfl_up(incl, off);
			if ((int)threadIdx.x >= off)
				incl += o;
		}
		if (s < ns)
			seg_cnt[first + s] = running + incl - v;
		running += __shfl(incl, 63);
	}
	if (threadIdx.x == 0)
		a.out_n[blockIdx.x] = running;
}

__global__ __launch_bounds__(256) void k_map_seg_scatter(MapCompactArgs a, const uint32_t *__restrict__ seg_base)
{
	__shared__ uint32_t wave_cnt[4];
	uint32_t cloud, seg;
	if (!map_segment(a, blockIdx.x, cloud, seg))
		return;
	const MapCloudArg c = a.cloud[cloud];
	const double r2 = a.radius * a.radius;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t running = seg_base[blockIdx.x];
	for (uint32_t k = 0; k < MAP_SEG; k += 256)
	{
		const uint32_t i = seg * MAP_SEG + k + threadIdx.x;
		bool keep = false;
		float4 r0, r1, rr;
		if (i < c.n)
		{
			r0 = c.in[(size_t)i * 3];
			r1 = c.in[(size_t)i * 3 + 1];
			rr = c.in[(size_t)i * 3 + 2];
			keep = map_keep(a, c, i, r0, r2);
		}
		const unsigned long long b = __ballot(keep);
		const uint32_t before = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
		__syncthreads();
		if (lane == 0)
			wave_cnt[wave] = (uint32_t)__popcll(b);
		__syncthreads();
		uint32_t wbase = 0, total = 0;
		for (int w = 0; w < 4; w++)
		{
			if (w < wave)
				wbase += wave_cnt[w];
			total += wave_cnt[w];
		}
		if (keep)
		{
			const size_t o = (size_t)(running + wbase + before) * 3;
			c.out[o] = r0;
			c.out[o + 1] = r1;
			c.out[o + 2] = rr;
		}
		running += total;
	}
}

// get_cloud_bbx (utility.hpp:817-848) over the six class clouds, and over the same points moved by pose_lo
// (pcl::transformPointCloud: double arithmetic, float store) — ordered-uint atomics; NaN coordinates never win a
// comparison in the reference and are skipped here.
__global__ __launch_bounds__(256) void k_map_bbox(MapBoxArgs a)
{
	const uint32_t cls = blockIdx.y;
	const float4 *recs = a.recs[cls];
	float lo[6], hi[6];
	for (int k = 0; k < 6; k++)
	{
		lo[k] = __builtin_inff();
		hi[k] = -__builtin_inff();
	}
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n[cls]; i += gridDim.x * blockDim.x)
	{
		const float4 p = recs[(size_t)i * 3];
		const double x = p.x, y = p.y, z = p.z;
		const float v[6] = {p.x, p.y, p.z, (float)(a.pose[0] * x + a.pose[1] * y + a.pose[2] * z + a.pose[3]),
							(float)(a.pose[4] * x + a.pose[5] * y + a.pose[6] * z + a.pose[7]),
							(float)(a.pose[8] * x + a.pose[9] * y + a.pose[10] * z + a.pose[11])};
		for (int k = 0; k < 6; k++)
		{
			lo[k] = fminf(lo[k], v[k]);
			hi[k] = fmaxf(hi[k], v[k]);
		}
	}
	for (int k = 0; k < 6; k++)
	{
		for (int off = 32; off > 0; off >>= 1)
		{
			lo[k] = fminf(lo[k], __shfl_down(lo[k], off));
			hi[k] = fmaxf(hi[k], __shfl_down(hi[k], off));
		}
		if ((threadIdx.x & 63) == 0)
		{
			// slot layout: [0..2] min xyz, [3..5] max xyz, [6..8] posed min, [9..11] posed max
			const int base = k < 3 ? 0 : 6, ax = k % 3;
			if (lo[k] <= hi[k])
			{
				atomicMin(&a.keys[base + ax], float_ord(lo[k]));
				atomicMax(&a.keys[base + 3 + ax], float_ord(hi[k]));
			}
		}
	}
}

// Exact nearest tree point (FLANN L2_Simple<float> distance) of every frame point, brute force: blockIdx.y splits the tree
// into chunks of MAP_NN_CHUNK points staged through LDS, atomicMin on the float bits combines the chunks (distances are >= 0).
#define MAP_NN_CHUNK 2048u // (a down-sampled frame against a 20 000-point map: 16384 made 8 workgroups of it, 284 us; 2048: ~40 workgroups)
#define MAP_NN_TILE 1024u
__global__ __launch_bounds__(256) void k_map_nn(const float4 *__restrict__ frame, uint32_t n_frame, const float4 *__restrict__ tree, uint32_t n_tree,
												 int use_box, double b0, double b1, double b2, double b3, double b4, double b5,
												 uint32_t *__restrict__ best)
{
	__shared__ float4 tile[MAP_NN_TILE];
	const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
	float qx = 0, qy = 0, qz = 0;
	if (q < n_frame)
	{
		const float4 p = frame[(size_t)q * 3];
		qx = p.x, qy = p.y, qz = p.z;
	}
	float d_best = __builtin_inff();
	const uint32_t t0 = blockIdx.y * MAP_NN_CHUNK, t1 = min(n_tree, t0 + MAP_NN_CHUNK);
	for (uint32_t base = t0; base < t1; base += MAP_NN_TILE)
	{
		__syncthreads();
		for (uint32_t k = threadIdx.x; k < MAP_NN_TILE; k += blockDim.x)
		{
			float4 t = make_float4(0, 0, 0, 0); // w = 1: member of the tree
			if (base + k < t1)
			{
				t = tree[(size_t)(base + k) * 3];
				const bool in = !use_box || (t.x > b0 && t.x < b3 && t.y > b1 && t.y < b4 && t.z > b2 && t.z < b5); // bbx_filter, cfilter.hpp:950-981
				t.w = in ? 1.0f : 0.0f;
			}
			tile[k] = t;
		}
		__syncthreads();
		const uint32_t cnt = min(MAP_NN_TILE, t1 - base);
		for (uint32_t k = 0; k < cnt; k++)
		{
			const float4 t = tile[k];
			float diff = qx - t.x;
			float d = diff * diff;
			diff = qy - t.y;
			d += diff * diff;
			diff = qz - t.z;
			d += diff * diff;
			if (t.w != 0.0f && d < d_best)
				d_best = d;
		}
	}
	if (q < n_frame && d_best < __builtin_inff())
		atomicMin(&best[q], __float_as_uint(d_best));
}

// map_scan_feature_pts_distance_removal's keep rule (map_manager.cpp:246-256), all in float as written there
__global__ void k_map_keep(const float4 *__restrict__ frame, uint32_t n_frame, const uint32_t *__restrict__ best, float center_radius, float dmin,
						   float dmax, float near, uint8_t *__restrict__ keep)
{
	const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= n_frame)
		return;
	const float4 p = frame[(size_t)q * 3];
	bool k = true;
	if (!(p.x * p.x + p.y * p.y > center_radius * center_radius))
	{
		const uint32_t bits = best[q];
		if (bits != 0x7f800000u) // an empty tree leaves the cloud alone (see oracle)
		{
			const float d2 = __uint_as_float(bits);
			k = (d2 > near * near && d2 < dmin * dmin) || d2 > dmax * dmax;
		}
	}
	keep[q] = k ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// MapManager::update_cloud_vectors (src/map_manager.cpp:258-292): every point of a linear-feature cloud gets the principal direction
// of its radius-limited k-neighbourhood (pca.hpp:209-290, :392-437) and survives only if the neighbourhood is linear enough and its
// direction steep (pillars) or flat (beams) enough.  One lane per point; the cloud streams through LDS in 256-point tiles and every
// lane keeps its `max_k` nearest neighbours as a sorted list in LDS (slot-major, so that neighbouring lanes touch neighbouring banks).
// A cloud of a few thousand points is all this ever sees (the map's pillar / beam share of max_num_pts): n^2 distances is microseconds.
//   neighbourhood   squared L2_Simple distance (((dx*dx)+dy*dy)+dz*dz, float) < radius^2, the max_k nearest in (distance, index) order,
//                   the point itself included — pcl::KdTreeFLANN::radiusSearch with max_nn
//   pcl::PCA        float centroid and float demeaned covariance summed in the neighbours' order, scaled by 1/(n-1); the
//                   eigen-decomposition of that float matrix in double (cyclic Jacobi), eigenvalues rounded to float
#define MAP_PCA_K 24 // list slots per lane (max_k <= 24: 48 KB of lists per workgroup)
__global__ __launch_bounds__(256) void k_map_pca(MapPcaArgs a)
{
	__shared__ float tx[256], ty[256], tz[256];
	__shared__ float ld[MAP_PCA_K * 256];
	__shared__ uint32_t li[MAP_PCA_K * 256];
	const uint32_t t = threadIdx.x, i = blockIdx.x * 256u + t;
	const bool live = i < a.n;
	float qx = 0, qy = 0, qz = 0;
	if (live)
	{
		const float4 r0 = a.recs[(size_t)i * 3];
		qx = r0.x, qy = r0.y, qz = r0.z;
	}
	const float r2 = a.radius * a.radius;
	const uint32_t K = (uint32_t)a.max_k;
	uint32_t m = 0;	   // list length
	float worst = r2; // a candidate enters while its distance is below this
	for (uint32_t base = 0; base < a.n; base += 256u)
	{
		__syncthreads();
		if (base + t < a.n)
		{
			const float4 r0 = a.recs[(size_t)(base + t) * 3];
			tx[t] = r0.x, ty[t] = r0.y, tz[t] = r0.z;
		}
		__syncthreads();
		const uint32_t cnt = min(256u, a.n - base);
		if (!live)
			continue;
		for (uint32_t j = 0; j < cnt; j++)
		{
			const float dx = qx - tx[j], dy = qy - ty[j], dz = qz - tz[j];
			const float d = dx * dx + dy * dy + dz * dz;
			if (!(d < worst))
				continue;
			// behind every entry with distance <= d: candidates arrive in index order, so ties stay in index order
			uint32_t pos = m < K ? m : K - 1u;
			while (pos > 0 && ld[(pos - 1u) * 256u + t] > d)
			{
				ld[pos * 256u + t] = ld[(pos - 1u) * 256u + t];
				li[pos * 256u + t] = li[(pos - 1u) * 256u + t];
				pos--;
			}
			ld[pos * 256u + t] = d;
			li[pos * 256u + t] = base + j;
			if (m < K)
				m++;
			if (m == K)
				worst = ld[(K - 1u) * 256u + t];
		}
	}
	if (!live)
		return;
	uint8_t keep = 0;
	if (m > 3u && (int)m >= a.min_k)
	{
		float mx = 0, my = 0, mz = 0;
		for (uint32_t k = 0; k < m; k++)
		{
			const float4 r0 = a.recs[(size_t)li[k * 256u + t] * 3];
			mx += r0.x, my += r0.y, mz += r0.z;
		}
		mx /= (float)m, my /= (float)m, mz /= (float)m;
		float s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0;
		for (uint32_t k = 0; k < m; k++)
		{
			const float4 r0 = a.recs[(size_t)li[k * 256u + t] * 3];
			const float dx = r0.x - mx, dy = r0.y - my, dz = r0.z - mz;
			s0 += dx * dx, s1 += dx * dy, s2 += dx * dz, s3 += dy * dy, s4 += dy * dz, s5 += dz * dz;
		}
		const float alpha = 1.f / ((float)m - 1.f);
		const mulls_pca::Eig E = mulls_pca::eigen3(alpha * s0, alpha * s1, alpha * s2, alpha * s3, alpha * s4, alpha * s5);
		const float e1 = E.e1, e2 = E.e2;
		float dx = E.px, dy = E.py, dz = E.pz;
		mulls_pca::normalize3(dx, dy, dz); // Vector3f::normalize()
		const double linear_2 = ((double)e1 - (double)e2) / (double)e1; // pca_feature_t keeps eigenvalues and ratios as doubles (pca.hpp:18-40)
		if (linear_2 > (double)a.min_linearity && (fabsf(dz) > a.sin_high || fabsf(dz) < a.sin_low))
		{
			keep = 1;
			a.recs[(size_t)i * 3 + 1] = make_float4(dx, dy, dz, (float)linear_2); // assign_normal(pt, feature, false)
			float4 r2v = a.recs[(size_t)i * 3 + 2];
			r2v.y = (float)linear_2; // curvature <- linearity (:280)
			a.recs[(size_t)i * 3 + 2] = r2v;
		}
	}
	a.keep[i] = keep;
}
void launch_map_pca(hipStream_t st, const MapPcaArgs &a)
{
	if (a.n)
		hipLaunchKernelGGL(k_map_pca, dim3((a.n + 255u) / 256u), dim3(256), 0, st, a);
}

uint32_t map_compact_segments(const MapCompactArgs &a)
{
	uint32_t ns = 0;
	for (int c = 0; c < 6; c++)
		ns += (a.cloud[c].n + MAP_SEG - 1u) / MAP_SEG;
	return ns;
}
void launch_map_compact(hipStream_t st, const MapCompactArgs &a, uint32_t *seg_scratch)
{
	const uint32_t ns = map_compact_segments(a);
	if (ns)
		hipLaunchKernelGGL(k_map_seg_count, dim3(ns), dim3(256), 0, st, a, seg_scratch);
	hipLaunchKernelGGL(k_map_seg_scan, dim3(6), dim3(64), 0, st, a, seg_scratch);
	if (ns)
		hipLaunchKernelGGL(k_map_seg_scatter, dim3(ns), dim3(256), 0, st, a, seg_scratch);
}
void launch_map_bbox(hipStream_t st, const MapBoxArgs &a) { hipLaunchKernelGGL(k_map_bbox, dim3(64, 6), dim3(256), 0, st, a); }
void launch_map_nn(hipStream_t st, const float4 *frame, uint32_t n_frame, const float4 *tree, uint32_t n_tree, int use_box, const double box[6],
				   uint32_t *best)
{
	if (!n_frame || !n_tree)
		return;
	hipLaunchKernelGGL(k_map_nn, dim3((n_frame + 255) / 256, (n_tree + MAP_NN_CHUNK - 1) / MAP_NN_CHUNK), dim3(256), 0, st, frame, n_frame, tree,
					   n_tree, use_box, box[0], box[1], box[2], box[3], box[4], box[5], best);
}
void launch_map_keep(hipStream_t st, const float4 *frame, uint32_t n_frame, const uint32_t *best, float center_radius, float dmin, float dmax,
					 float near, uint8_t *keep)
{
	if (n_frame)
		hipLaunchKernelGGL(k_map_keep, dim3((n_frame + 255) / 256), dim3(256), 0, st, frame, n_frame, best, center_radius, dmin, dmax, near, keep);
}
