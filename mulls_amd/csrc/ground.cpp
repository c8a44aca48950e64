// ground.cpp — host side of mulls_ground_filter and mulls_extract_features (include/mulls_hip.h): CFilter::fast_ground_filter on the device
// (k_ground.hip), and extract_semantic_pts' chain scanner filter -> ground filter -> classify_nground_pts with the clouds staying on the device.
// Upload the scan, one launch, download the two clouds; cloud_ground_down is a subset of cloud_ground by index (every
// ground_random_down_down_rate-th point, or the ABI's seeded fixed-number selection: cfilter.hpp:1955-1968) and is taken on the way out.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

#include "../../include/mulls_hip.h"
#include "ctx.h"
#include "device_types.h"
#include "map_launch.h"

#include "classify_launch.h"
#include "ground_launch.h"

extern "C" __attribute__((visibility("hidden"))) int mulls_classify_impl(mulls_ctx *ctx, const void *pts, bool pts_on_device, uint32_t n_in, uint32_t stride, const mulls_classify_params *P,
								   void *const out[MULLS_CL_COUNT], const uint32_t cap[MULLS_CL_COUNT], uint32_t n_out[MULLS_CL_COUNT], void *cloud_in_after,
								   uint32_t *n_cloud_in_after, ClassifyDev *dev_out);

extern "C"
{
	void mulls_ground_default_params(mulls_ground_params *p)
	{
		if (!p)
			return;
		std::memset(p, 0, sizeof(*p));
		// extract_semantic_pts' defaults (cfilter.hpp:2296-2316) where it has them, the flag defaults of test/mulls_slam.cpp otherwise;
		// normal method 0 / distance-inverse sampling 0: the deterministic skeleton (see the header)
		p->min_grid_pt_num = 8;
		p->grid_resolution = 3.0f;
		p->max_height_difference = 0.3f;
		p->neighbor_height_diff = 1.5f;
		p->max_ground_height = 2.0f;
		p->ground_random_down_rate = 10;
		p->ground_random_down_down_rate = 2;
		p->nonground_random_down_rate = 3;
		p->reliable_neighbor_grid_num_thre = 0;
		p->estimate_ground_normal_method = 0;
		p->distance_weight_downsampling_method = 0;
		p->standard_distance = 15.0f;
		p->fixed_num_downsampling = 0;
		p->apply_grid_wise_outlier_filter = 0;
		p->down_ground_fixed_num = 500;
		p->intensity_thre = 3.402823466e+38f;
		p->outlier_std_scale = 3.0f;
		p->normal_estimation_radius = 2.0f;
		p->rng_seed = 0;
	}

	// ---- shared by mulls_ground_filter and mulls_extract_features -------------------------------------------------------------------
	struct GfArena // one device arena: scan | ground | unground | ids | d3v | cellof | code | state, counters, per-cell tables | (filters / voxels ahead:) scan copy, mask, scratch, voxel keys, order
	{
		size_t o_ground, o_unground, o_ids, o_d3, o_cell, o_code, o_aux, o_alt, o_mask, o_seg, o_keys, o_perm, o_ransac, o_normals, total;
	};
	// prefilter: dist_filter / scanner_filter ahead of the ground filter; voxels: voxel_downsample ahead of it (either needs a second scan-sized buffer)
	static GfArena gf_layout(uint32_t n, bool prefilter, bool voxels, int normal_method = 0)
	{
		GfArena a;
		const size_t rec = (size_t)n * MULLS_POINT_BYTES;
		a.o_ground = rec, a.o_unground = 2 * rec, a.o_ids = 3 * rec, a.o_d3 = a.o_ids + (size_t)n * 4, a.o_cell = a.o_d3 + (size_t)n * 4;
		a.o_code = a.o_cell + (size_t)n * 2;
		a.o_aux = (a.o_code + n + 255) & ~(size_t)255;
		a.o_alt = (a.o_aux + ground_filter_aux_bytes(n) + 255) & ~(size_t)255;
		a.o_mask = a.o_alt + (prefilter || voxels ? rec : 0);
		a.o_seg = (a.o_mask + (prefilter ? n : 0) + 255) & ~(size_t)255;
		a.o_keys = (a.o_seg + (prefilter ? ((size_t)n / 4096 + 32) * 4 * 8 : 0) + 255) & ~(size_t)255;
		a.o_perm = a.o_keys + (voxels ? (size_t)n * 8 + 256 : 0);
		// normal method 3: gg | perm | inl (uint32 each) | gxyz (float4) | cell_nrm (float4 per cell); methods 1 / 2: the neighbour grid of the ground cloud
		a.o_ransac = (a.o_perm + (voxels ? (size_t)n * 4 : 0) + 255) & ~(size_t)255;
		a.o_normals = a.o_ransac + (normal_method == 3 ? (size_t)n * 28 + (size_t)MULLS_GF_MAXCELLS * 16 + 256 : 0);
		a.total = a.o_normals + ((normal_method == 1 || normal_method == 2) ? ground_normals_bytes(n) : 0);
		return a;
	}
	static int gf_check(mulls_ctx *ctx, const mulls_ground_params *P, uint32_t n)
	{
		if (P->estimate_ground_normal_method < 0 || P->estimate_ground_normal_method > 3)
		{
			ctx->err = "mulls_ground_filter: estimate_ground_normal_method must be 0 .. 3";
			return MULLS_E_INVALID;
		}
		if (P->estimate_ground_normal_method == 1 && !(P->normal_estimation_radius > 0.0f))
		{
			ctx->err = "mulls_ground_filter: normal method 1 needs a positive normal_estimation_radius";
			return MULLS_E_INVALID;
		}
		if (P->estimate_ground_normal_method == 2 && 2 * (long)P->min_grid_pt_num > 64)
		{
			ctx->err = "mulls_ground_filter: normal method 2 searches 2 * min_grid_pt_num neighbours, at most 64";
			return MULLS_E_UNSUPPORTED;
		}
		if (P->min_grid_pt_num < 1 || !(P->grid_resolution > 0.0f) || P->ground_random_down_rate < 1 || P->ground_random_down_down_rate < 1 ||
			P->nonground_random_down_rate < 1 || P->distance_weight_downsampling_method < 0 || P->distance_weight_downsampling_method > 2)
		{
			ctx->err = "mulls_ground_filter: min_grid_pt_num, grid_resolution and the down-sampling rates must be positive";
			return MULLS_E_INVALID;
		}
		if (n > 500000u)
		{
			ctx->err = "mulls_ground_filter: more than 500000 points in one scan";
			return MULLS_E_UNSUPPORTED;
		}
		return MULLS_OK;
	}
	static int gf_reserve(mulls_ctx *ctx, const GfArena &a)
	{
		if (ctx->gf_cap < a.total)
		{
			if (ctx->gf_buf)
				(void)hipFree(ctx->gf_buf);
			ctx->gf_buf = nullptr;
			ctx->gf_cap = 0;
			HIPCHK(ctx, hipMalloc(&ctx->gf_buf, a.total + a.total / 4));
			ctx->gf_cap = a.total + a.total / 4;
		}
		return MULLS_OK;
	}
	// PCL's sample sequence for the plane RANSAC of normal method 3: SACSegmentation(random = false) seeds boost::mt19937 with 12345u for every
	// model and draws through boost::uniform_int<>(0, INT_MAX), i.e. output / 2 — the same draws for every grid cell, so one table serves all
	static int gf_rnd_table(mulls_ctx *ctx)
	{
		if (ctx->gf_rnd)
			return MULLS_OK;
		std::vector<uint32_t> t(MULLS_GF_RND);
		std::mt19937 eng(12345u);
		for (uint32_t &v : t)
			v = (uint32_t)(eng() >> 1);
		HIPCHK(ctx, hipMalloc(&ctx->gf_rnd, t.size() * sizeof(uint32_t)));
		HIPCHK(ctx, hipMemcpy(ctx->gf_rnd, t.data(), t.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
		return MULLS_OK;
	}
	// the filter on the n records at `scan` (device, inside the arena); the stream is idle afterwards
	static int gf_run(mulls_ctx *ctx, const GfArena &a, const float4 *scan, uint32_t n, const mulls_ground_params *P, GfOut &out)
	{
		unsigned char *base = static_cast<unsigned char *>(ctx->gf_buf);
		hipStream_t st = ctx->stream;
		GfRansac R = {};
		if (P->estimate_ground_normal_method == 3)
		{
			if (gf_rnd_table(ctx) != MULLS_OK)
				return MULLS_E_HIP;
			unsigned char *r = base + a.o_ransac;
			R.gg = reinterpret_cast<uint32_t *>(r);
			R.perm = R.gg + n;
			R.inl = R.perm + n;
			R.gxyz = reinterpret_cast<float4 *>((reinterpret_cast<uintptr_t>(R.inl + n) + 15) & ~(uintptr_t)15);
			R.cell_nrm = R.gxyz + n;
			R.rnd = static_cast<const uint32_t *>(ctx->gf_rnd);
		}
		if (launch_ground_filter(st, scan, n, *P, reinterpret_cast<uint32_t *>(base + a.o_ids), reinterpret_cast<uint16_t *>(base + a.o_cell), base + a.o_code,
								 reinterpret_cast<float *>(base + a.o_d3), reinterpret_cast<float4 *>(base + a.o_ground), reinterpret_cast<float4 *>(base + a.o_unground),
								 base + a.o_aux, R) != 0)
		{
			ctx->err = "mulls_ground_filter: launch failed";
			return MULLS_E_HIP;
		}
		HIPCHK(ctx, hipMemcpyAsync(&out, base + a.o_aux, sizeof(out), hipMemcpyDeviceToHost, st));
		HIPCHK(ctx, hipStreamSynchronize(st));
		if (out.error)
		{
			ctx->err = "mulls_ground_filter: the grid has too many cells (more than 65536, or more than 64 M table entries: grid_resolution too fine for this scan's extent)";
			return MULLS_E_UNSUPPORTED;
		}
		if ((P->estimate_ground_normal_method == 1 || P->estimate_ground_normal_method == 2) && out.n_ground)
		{
			// normals of cloud_ground from its own neighbourhoods (:1943-1954); the error word of the state reports a neighbourhood beyond the buffer
			uint32_t *err_word = reinterpret_cast<uint32_t *>(base + a.o_aux) + 3;
			launch_ground_normals(st, reinterpret_cast<float4 *>(base + a.o_ground), out.n_ground, P->estimate_ground_normal_method == 1 ? P->normal_estimation_radius : 0.0f,
								  2 * P->min_grid_pt_num, base + a.o_normals, err_word);
			uint32_t e = 0;
			HIPCHK(ctx, hipMemcpyAsync(&e, err_word, sizeof(e), hipMemcpyDeviceToHost, st));
			HIPCHK(ctx, hipStreamSynchronize(st));
			if (e)
			{
				ctx->err = "mulls_ground_filter: more than 1024 ground points within normal_estimation_radius of one point";
				return MULLS_E_UNSUPPORTED;
			}
		}
		return MULLS_OK;
	}
	// CFilter::voxel_downsample (cfilter.hpp:83-160) on the n records at `in` (device): pc_down is gathered to `down` (device, room for n records),
	// *n_down = its size.  Bounding box and voxel indices come from the device; the order of the (voxel, index) pairs is std::sort's on the host,
	// as upstream's is: which point of a voxel comes first is decided by that sort (the comparison sees the voxel only, :42), and it depends on
	// nothing but the indices' values and their count, so the same sort over the same indices picks the same points.  The stream is idle afterwards.
	static int vox_run(mulls_ctx *ctx, const GfArena &a, const float4 *in, uint32_t n, float voxel_size, float4 *down, uint32_t *n_down)
	{
		unsigned char *base = static_cast<unsigned char *>(ctx->gf_buf);
		hipStream_t st = ctx->stream;
		unsigned long long *keys = reinterpret_cast<unsigned long long *>(base + a.o_keys);
		uint32_t *box = reinterpret_cast<uint32_t *>(base + a.o_keys + (size_t)n * 8);
		uint32_t *perm = reinterpret_cast<uint32_t *>(base + a.o_perm);
		*n_down = 0;
		if (n == 0)
			return MULLS_OK;
		uint32_t hb[7];
		if (launch_vox_bbox(st, in, n, box) != 0)
		{
			ctx->err = "mulls_voxel_downsample: launch failed";
			return MULLS_E_HIP;
		}
		HIPCHK(ctx, hipMemcpyAsync(hb, box, sizeof(hb), hipMemcpyDeviceToHost, st));
		HIPCHK(ctx, hipStreamSynchronize(st));
		if (hb[6])
		{
			ctx->err = "mulls_voxel_downsample: non-finite coordinate";
			return MULLS_E_INVALID;
		}
		float min_p[3], max_p[3];
		for (int k = 0; k < 3; k++)
		{
			const uint32_t lo = hb[k], hi = hb[3 + k]; // ordered keys back to floats
			const uint32_t ul = (lo & 0x80000000u) ? (lo & 0x7fffffffu) : ~lo, uh = (hi & 0x80000000u) ? (hi & 0x7fffffffu) : ~hi;
			std::memcpy(&min_p[k], &ul, 4);
			std::memcpy(&max_p[k], &uh, 4);
		}
		const float inverse_voxel_size = 1.0f / voxel_size;
		const float gap_p[3] = {max_p[0] - min_p[0], max_p[1] - min_p[1], max_p[2] - min_p[2]};
		for (int k = 0; k < 3; k++)
			if (!(std::ceil((double)(gap_p[k] * inverse_voxel_size)) + 1 < 2097152.0))
			{
				ctx->err = "mulls_voxel_downsample: more than 2^21 voxels along an axis";
				return MULLS_E_UNSUPPORTED;
			}
		const unsigned long long max_vy = (unsigned long long)(std::ceil(gap_p[1] * inverse_voxel_size) + 1);
		const unsigned long long max_vz = (unsigned long long)(std::ceil(gap_p[2] * inverse_voxel_size) + 1);
		launch_vox_keys(st, in, n, min_p, inverse_voxel_size, max_vy * max_vz, max_vz, keys);
		std::vector<unsigned long long> hk(n);
		HIPCHK(ctx, hipMemcpyAsync(hk.data(), keys, (size_t)n * 8, hipMemcpyDeviceToHost, st));
		HIPCHK(ctx, hipStreamSynchronize(st));
		struct IdPair
		{
			unsigned long long voxel_idx;
			int idx;
			bool operator<(const IdPair &o) const { return voxel_idx < o.voxel_idx; }
		};
		std::vector<IdPair> id_pairs(n);
		for (uint32_t i = 0; i < n; i++)
			id_pairs[i].voxel_idx = hk[i], id_pairs[i].idx = (int)i;
		std::sort(id_pairs.begin(), id_pairs.end());
		std::vector<uint32_t> first;
		first.reserve(n);
		for (size_t b = 0; b < id_pairs.size();)
		{
			first.push_back((uint32_t)id_pairs[b].idx);
			size_t c = b + 1;
			while (c < id_pairs.size() && id_pairs[c].voxel_idx == id_pairs[b].voxel_idx)
				c++;
			b = c;
		}
		HIPCHK(ctx, hipMemcpyAsync(perm, first.data(), first.size() * 4, hipMemcpyHostToDevice, st));
		launch_cl_gather(st, in, perm, down, (uint32_t)first.size()); // whole 48-byte records
		HIPCHK(ctx, hipStreamSynchronize(st));
		*n_down = (uint32_t)first.size();
		return MULLS_OK;
	}
	// which points of cloud_ground make cloud_ground_down (cfilter.hpp:1955-1968): every ground_random_down_down_rate-th, or the seeded fixed-number selection
	static void gf_ground_down_indices(const mulls_ground_params *P, uint32_t n_ground, std::vector<uint32_t> &idx)
	{
		idx.clear();
		if (!P->fixed_num_downsampling)
		{
			for (uint32_t i = 0; i < n_ground; i++)
				if ((int)i % P->ground_random_down_down_rate == 0)
					idx.push_back(i);
		}
		else
		{
			std::vector<uint8_t> mask(std::max<uint32_t>(n_ground, 1));
			thin_mask(mask.data(), n_ground, P->down_ground_fixed_num, P->rng_seed, 12);
			for (uint32_t i = 0; i < n_ground; i++)
				if (mask[i])
					idx.push_back(i);
		}
	}
	// ... taken on the host from the downloaded cloud (n_ground records at g)
	static uint32_t gf_ground_down(const mulls_ground_params *P, const unsigned char *g, uint32_t n_ground, void *ground_down, uint32_t cap_ground_down)
	{
		std::vector<uint32_t> idx;
		gf_ground_down_indices(P, n_ground, idx);
		unsigned char *gd = static_cast<unsigned char *>(ground_down);
		for (size_t k = 0; k < idx.size() && k < cap_ground_down; k++)
			std::memcpy(gd + k * MULLS_POINT_BYTES, g + (size_t)idx[k] * MULLS_POINT_BYTES, MULLS_POINT_BYTES);
		return (uint32_t)idx.size();
	}

	int mulls_ground_filter(mulls_ctx *ctx, const void *pts, uint32_t n, uint32_t stride, const mulls_ground_params *P, void *ground, uint32_t cap_ground,
							void *ground_down, uint32_t cap_ground_down, void *unground, uint32_t cap_unground, uint32_t n_out[3])
	try
	{
		if (!ctx || !P || !n_out || (n && !pts) || stride < MULLS_POINT_BYTES || (cap_ground && !ground) || (cap_ground_down && !ground_down) ||
			(cap_unground && !unground))
			return MULLS_E_INVALID;
		n_out[0] = n_out[1] = n_out[2] = 0;
		const int chk = gf_check(ctx, P, n);
		if (chk != MULLS_OK)
			return chk;
		if (n == 0)
			return MULLS_OK; // (the reference divides by a zero sample count here)
		HIPCHK(ctx, hipSetDevice(ctx->device));
		hipStream_t st = ctx->stream;
		const mulls::StreamDrain drain{st}; // error returns below leave copies into `unground` / `g` queued
		const GfArena a = gf_layout(n, false, false, P->estimate_ground_normal_method);
		if (gf_reserve(ctx, a) != MULLS_OK)
			return MULLS_E_HIP;
		unsigned char *base = static_cast<unsigned char *>(ctx->gf_buf);
		const size_t rec = (size_t)n * MULLS_POINT_BYTES;
		if (stride == MULLS_POINT_BYTES)
			HIPCHK(ctx, hipMemcpyAsync(base, pts, rec, hipMemcpyHostToDevice, st));
		else
			HIPCHK(ctx, hipMemcpy2DAsync(base, MULLS_POINT_BYTES, pts, stride, MULLS_POINT_BYTES, n, hipMemcpyHostToDevice, st));
		GfOut out;
		const int rc = gf_run(ctx, a, reinterpret_cast<const float4 *>(base), n, P, out);
		if (rc != MULLS_OK)
			return rc;
		std::vector<unsigned char> g((size_t)out.n_ground * MULLS_POINT_BYTES);
		if (out.n_ground)
			HIPCHK(ctx, hipMemcpyAsync(g.data(), base + a.o_ground, g.size(), hipMemcpyDeviceToHost, st));
		const uint32_t ku = std::min(out.n_unground, cap_unground);
		if (ku)
			HIPCHK(ctx, hipMemcpyAsync(unground, base + a.o_unground, (size_t)ku * MULLS_POINT_BYTES, hipMemcpyDeviceToHost, st));
		HIPCHK(ctx, hipStreamSynchronize(st));
		const uint32_t kg = std::min(out.n_ground, cap_ground);
		if (kg)
			std::memcpy(ground, g.data(), (size_t)kg * MULLS_POINT_BYTES);
		n_out[0] = out.n_ground;
		n_out[1] = gf_ground_down(P, g.data(), out.n_ground, ground_down, cap_ground_down);
		n_out[2] = out.n_unground;
		return MULLS_OK;
	}
	catch (...)
	{
		return mulls::abi_caught(ctx); // nothing is thrown across the ABI
	}

	void mulls_extract_default_params(mulls_extract_params *p)
	{
		if (!p)
			return;
		std::memset(p, 0, sizeof(*p));
		mulls_ground_default_params(&p->ground);
		mulls_classify_default_params(&p->classify);
		// extract_semantic_pts' scanner filter (cfilter.hpp:2338-2346) with approx_scanner_height 2.0, underground_thre -7.0
		p->apply_scanner_filter = 0;
		p->self_ring_radius = 1.75f;
		p->ghost_radius = 20.0f;
		p->z_min = -2.0f - 4.0f;
		p->z_min_min = -2.0f + -7.0f;
		// test/mulls_slam.cpp:51-53 (apply_dist_filter false, min_dist_used 1.0, max_dist_used 120.0), :40 (cloud_down_res 0.0)
		p->apply_dist_filter = 0;
		p->min_dist_used = 1.0;
		p->max_dist_used = 120.0;
		p->vf_downsample_resolution = 0.0f;
	}

	int mulls_voxel_downsample(mulls_ctx *ctx, const void *pts, uint32_t n, uint32_t stride, float voxel_size, void *out, uint32_t cap, uint32_t *n_out)
	try
	{
		if (!ctx || !n_out || (n && !pts) || (cap && !out) || stride < MULLS_POINT_BYTES)
			return MULLS_E_INVALID;
		*n_out = 0;
		if (voxel_size < 0.001) // `cloud_out = cloud_in` (:90-97)
		{
			*n_out = n;
			for (uint32_t i = 0; i < std::min(n, cap); i++)
				std::memcpy(static_cast<unsigned char *>(out) + (size_t)i * MULLS_POINT_BYTES, static_cast<const unsigned char *>(pts) + (size_t)i * stride, MULLS_POINT_BYTES);
			return MULLS_OK;
		}
		if (n == 0)
			return MULLS_OK;
		if (n > 500000u)
		{
			ctx->err = "mulls_voxel_downsample: more than 500000 points in one scan";
			return MULLS_E_UNSUPPORTED;
		}
		HIPCHK(ctx, hipSetDevice(ctx->device));
		hipStream_t st = ctx->stream;
		const mulls::StreamDrain drain{st};
		const GfArena a = gf_layout(n, false, true);
		if (gf_reserve(ctx, a) != MULLS_OK)
			return MULLS_E_HIP;
		unsigned char *base = static_cast<unsigned char *>(ctx->gf_buf);
		if (stride == MULLS_POINT_BYTES)
			HIPCHK(ctx, hipMemcpyAsync(base + a.o_alt, pts, (size_t)n * MULLS_POINT_BYTES, hipMemcpyHostToDevice, st));
		else
			HIPCHK(ctx, hipMemcpy2DAsync(base + a.o_alt, MULLS_POINT_BYTES, pts, stride, MULLS_POINT_BYTES, n, hipMemcpyHostToDevice, st));
		uint32_t nd = 0;
		const int rc = vox_run(ctx, a, reinterpret_cast<const float4 *>(base + a.o_alt), n, voxel_size, reinterpret_cast<float4 *>(base), &nd);
		if (rc != MULLS_OK)
			return rc;
		*n_out = nd;
		if (std::min(nd, cap))
		{
			HIPCHK(ctx, hipMemcpyAsync(out, base, (size_t)std::min(nd, cap) * MULLS_POINT_BYTES, hipMemcpyDeviceToHost, st));
			HIPCHK(ctx, hipStreamSynchronize(st));
		}
		return MULLS_OK;
	}
	catch (...)
	{
		return mulls::abi_caught(ctx); // nothing is thrown across the ABI
	}

	// blk: keep the feature clouds on the device (mulls_extract_features_resident) instead of copying them to out[]
	static int extract_impl(mulls_ctx *ctx, const void *scan, uint32_t n_in, uint32_t stride, const mulls_extract_params *X, void *const out[MULLS_EX_COUNT],
							const uint32_t cap[MULLS_EX_COUNT], uint32_t n_out[MULLS_EX_COUNT], mulls_block *blk)
	{
		if (!ctx || !X || !out || !cap || !n_out || (n_in && !scan) || stride < MULLS_POINT_BYTES)
			return MULLS_E_INVALID;
		if (blk)
			for (int k = 0; k < MULLS_EX_COUNT; k++)
				blk->n[k] = 0;
		for (int k = 0; k < MULLS_EX_COUNT; k++)
		{
			n_out[k] = 0;
			if (cap[k] && !out[k])
				return MULLS_E_INVALID;
		}
		const mulls_ground_params *P = &X->ground;
		const int chk = gf_check(ctx, P, n_in);
		if (chk != MULLS_OK)
			return chk;
		if (n_in == 0)
			return MULLS_OK;
		HIPCHK(ctx, hipSetDevice(ctx->device));
		hipStream_t st = ctx->stream;
		const mulls::StreamDrain drain{st}; // error returns below leave copies into out[] / `g` queued
		const bool scanner = X->apply_scanner_filter != 0, dist = X->apply_dist_filter != 0, prefilter = scanner || dist;
		const bool voxels = !(X->vf_downsample_resolution < 0.001); // voxel_downsample hands the cloud on below 0.001 m (:90-97)
		const GfArena a = gf_layout(n_in, prefilter, voxels, P->estimate_ground_normal_method);
		if (gf_reserve(ctx, a) != MULLS_OK)
			return MULLS_E_HIP;
		unsigned char *base = static_cast<unsigned char *>(ctx->gf_buf);
		const size_t rec_in = (size_t)n_in * MULLS_POINT_BYTES;
		// two scan-sized buffers (base, alt): every stage ahead of the ground filter reads one and writes the other
		unsigned char *cur = (prefilter != voxels) ? base + a.o_alt : base, *other = (prefilter != voxels) ? base : base + a.o_alt;
		if (stride == MULLS_POINT_BYTES)
		{
			// A pageable scan of a few MB is pinned and unpinned by the runtime around its copy (0.15 ms in front of a 0.1 ms transfer on the frame path): the
			// process's host pool moves it into the context's pinned scratch instead, and the transfer starts from there.  A caller's pinned buffer goes up as it is.
			const void *up = scan;
			hipPointerAttribute_t at;
			// (an unregistered host pointer is "invalid value" to the query, or of type "unregistered": only that is staged — pinned, managed or device memory goes as it is)
			const hipError_t qe = hipPointerGetAttributes(&at, scan);
			const bool pageable = qe != hipSuccess || at.type == hipMemoryTypeUnregistered;
			if (qe != hipSuccess)
				(void)hipGetLastError(); // not an error of this call
			if (pageable && rec_in >= ((size_t)1 << 20))
			{
				if (grow_pinned(ctx, &ctx->scan_pin, &ctx->scan_pin_cap, rec_in, hipHostMallocDefault) != MULLS_OK)
					return MULLS_E_HIP;
				const size_t chunk = (size_t)256 << 10;
				const long nchunk = (long)((rec_in + chunk - 1) / chunk);
				unsigned char *dst = ctx->scan_pin;
				const unsigned char *srcb = static_cast<const unsigned char *>(scan);
				const std::function<void(long)> move = [&](long k) {
					const size_t o = (size_t)k * chunk;
					std::memcpy(dst + o, srcb + o, std::min(chunk, rec_in - o));
				};
				shared_host_pool().parallel_for(0, nchunk, 1, move);
				up = dst;
			}
			HIPCHK(ctx, hipMemcpyAsync(cur, up, rec_in, hipMemcpyHostToDevice, st));
		}
		else
			HIPCHK(ctx, hipMemcpy2DAsync(cur, MULLS_POINT_BYTES, scan, stride, MULLS_POINT_BYTES, n_in, hipMemcpyHostToDevice, st));
		uint32_t n = n_in;
		if (prefilter)
		{
			// dist_filter (test/mulls_slam.cpp:359-360 -> cfilter.hpp:806-832), then scanner_filter (cfilter.hpp:2338-2346 -> :914-929)
			uint32_t *counts = reinterpret_cast<uint32_t *>(base + a.o_seg);
			RawMaskArgs ma;
			std::memset(&ma, 0, sizeof(ma));
			ma.dist_on = dist, ma.scanner_on = scanner;
			ma.dist_min_sq = X->min_dist_used * X->min_dist_used, ma.dist_max_sq = X->max_dist_used * X->max_dist_used;
			ma.self_radius = X->self_ring_radius, ma.ghost_radius = X->ghost_radius, ma.z_min_ghost = X->z_min, ma.z_min_global = X->z_min_min;
			launch_raw_mask(st, reinterpret_cast<const float4 *>(cur), n_in, ma, base + a.o_mask);
			MapCompactArgs ca;
			std::memset(&ca, 0, sizeof(ca));
			ca.cloud[0].in = reinterpret_cast<const float4 *>(cur);
			ca.cloud[0].out = reinterpret_cast<float4 *>(other);
			ca.cloud[0].mask = base + a.o_mask;
			ca.cloud[0].n = n_in;
			ca.out_n = counts;
			ca.mode = 0;
			launch_map_compact(st, ca, counts + 8);
			uint32_t c6[6];
			HIPCHK(ctx, hipMemcpyAsync(c6, counts, sizeof(c6), hipMemcpyDeviceToHost, st));
			HIPCHK(ctx, hipStreamSynchronize(st));
			n = c6[0];
			std::swap(cur, other);
		}
		n_out[MULLS_EX_RAW] = n;
		n_out[MULLS_EX_DOWN] = n;
		if (n == 0)
			return MULLS_OK;
		const uint32_t kr = std::min(n, cap[MULLS_EX_RAW]);
		if (kr)
			HIPCHK(ctx, hipMemcpyAsync(out[MULLS_EX_RAW], cur, (size_t)kr * MULLS_POINT_BYTES, hipMemcpyDeviceToHost, st));
		if (voxels)
		{
			uint32_t nd = 0;
			const int rcv = vox_run(ctx, a, reinterpret_cast<const float4 *>(cur), n, X->vf_downsample_resolution, reinterpret_cast<float4 *>(other), &nd);
			if (rcv != MULLS_OK)
				return rcv;
			std::swap(cur, other);
			n = nd;
			n_out[MULLS_EX_DOWN] = n;
		}
		// cur == base here: one stage (or none) when exactly one of the two ran from alt, two stages from base
		const uint32_t kd = std::min(n, cap[MULLS_EX_DOWN]);
		if (kd)
			HIPCHK(ctx, hipMemcpyAsync(out[MULLS_EX_DOWN], cur, (size_t)kd * MULLS_POINT_BYTES, hipMemcpyDeviceToHost, st));
		GfOut go;
		const int rc = gf_run(ctx, a, reinterpret_cast<const float4 *>(cur), n, P, go);
		if (rc != MULLS_OK)
			return rc;
		if (blk)
		{
			// ---- device-resident block: nothing but selection indices, the clouds the host-side samplers thin and the sizes cross PCIe -------------
			ClassifyDev cd;
			void *no_out[MULLS_CL_COUNT] = {};
			uint32_t no_cap[MULLS_CL_COUNT] = {};
			const int rc2 = mulls_classify_impl(ctx, base + a.o_unground, true, go.n_unground, MULLS_POINT_BYTES, &X->classify, no_out, no_cap, n_out + MULLS_EX_PILLAR, nullptr,
												&n_out[MULLS_EX_UNGROUND], &cd);
			if (rc2 != MULLS_OK)
				return rc2;
			std::vector<uint32_t> gd_idx;
			gf_ground_down_indices(P, go.n_ground, gd_idx);
			n_out[MULLS_EX_GROUND] = go.n_ground;
			n_out[MULLS_EX_GROUND_DOWN] = (uint32_t)gd_idx.size();
			uint32_t cnt[MULLS_EX_COUNT] = {};
			cnt[MULLS_EX_GROUND] = go.n_ground, cnt[MULLS_EX_GROUND_DOWN] = (uint32_t)gd_idx.size(), cnt[MULLS_EX_UNGROUND] = cd.n_after;
			for (int k = 0; k < MULLS_CL_COUNT; k++)
				cnt[MULLS_EX_PILLAR + k] = cd.n[k];
			size_t need = 0, off[MULLS_EX_COUNT];
			for (int k = 0; k < MULLS_EX_COUNT; k++)
			{
				off[k] = need;
				need += ((size_t)cnt[k] * MULLS_POINT_BYTES + 255) & ~(size_t)255;
			}
			need += (size_t)gd_idx.size() * 4 + 256; // the selection indices ride at the end
			if (blk->cap < need)
			{
				if (blk->buf)
					(void)hipFree(blk->buf);
				blk->buf = nullptr, blk->cap = 0;
				HIPCHK(ctx, hipMalloc((void **)&blk->buf, need + need / 4));
				blk->cap = need + need / 4;
			}
			auto put = [&](int k, const void *src, hipMemcpyKind kind) -> hipError_t {
				return cnt[k] ? hipMemcpyAsync(blk->buf + off[k], src, (size_t)cnt[k] * MULLS_POINT_BYTES, kind, st) : hipSuccess;
			};
			HIPCHK(ctx, put(MULLS_EX_GROUND, base + a.o_ground, hipMemcpyDeviceToDevice));
			if (!gd_idx.empty())
			{
				uint32_t *d_idx = reinterpret_cast<uint32_t *>(blk->buf + need - (gd_idx.size() * 4 + 128));
				HIPCHK(ctx, hipMemcpyAsync(d_idx, gd_idx.data(), gd_idx.size() * 4, hipMemcpyHostToDevice, st));
				launch_cl_gather(st, reinterpret_cast<const float4 *>(base + a.o_ground), d_idx, reinterpret_cast<float4 *>(blk->buf + off[MULLS_EX_GROUND_DOWN]), (uint32_t)gd_idx.size());
			}
			HIPCHK(ctx, put(MULLS_EX_UNGROUND, cd.cloud_in_after, hipMemcpyDeviceToDevice));
			for (int k = 0; k < MULLS_CL_COUNT; k++)
				HIPCHK(ctx, cd.on_host[k] ? put(MULLS_EX_PILLAR + k, cd.host[k].data(), hipMemcpyHostToDevice) : put(MULLS_EX_PILLAR + k, cd.dev[k], hipMemcpyDeviceToDevice));
			HIPCHK(ctx, hipStreamSynchronize(st)); // gd_idx, cd.host are host vectors; the arenas the copies read are reused by the next call
			for (int k = 0; k < MULLS_EX_COUNT; k++)
				blk->off[k] = off[k], blk->n[k] = cnt[k];
			return MULLS_OK;
		}
		std::vector<unsigned char> g((size_t)go.n_ground * MULLS_POINT_BYTES);
		if (go.n_ground)
			HIPCHK(ctx, hipMemcpyAsync(g.data(), base + a.o_ground, g.size(), hipMemcpyDeviceToHost, st));
		// classify_nground_pts on the non-ground cloud where the filter left it
		const int rc2 = mulls_classify_impl(ctx, base + a.o_unground, true, go.n_unground, MULLS_POINT_BYTES, &X->classify, out + MULLS_EX_PILLAR, cap + MULLS_EX_PILLAR,
											n_out + MULLS_EX_PILLAR, cap[MULLS_EX_UNGROUND] >= go.n_unground ? out[MULLS_EX_UNGROUND] : nullptr, &n_out[MULLS_EX_UNGROUND], nullptr);
		if (rc2 != MULLS_OK)
			return rc2;
		HIPCHK(ctx, hipStreamSynchronize(st));
		const uint32_t kg = std::min(go.n_ground, cap[MULLS_EX_GROUND]);
		if (kg)
			std::memcpy(out[MULLS_EX_GROUND], g.data(), (size_t)kg * MULLS_POINT_BYTES);
		n_out[MULLS_EX_GROUND] = go.n_ground;
		n_out[MULLS_EX_GROUND_DOWN] = gf_ground_down(P, g.data(), go.n_ground, out[MULLS_EX_GROUND_DOWN], cap[MULLS_EX_GROUND_DOWN]);
		return MULLS_OK;
	}

	int mulls_extract_features(mulls_ctx *ctx, const void *scan, uint32_t n_in, uint32_t stride, const mulls_extract_params *X, void *const out[MULLS_EX_COUNT],
							   const uint32_t cap[MULLS_EX_COUNT], uint32_t n_out[MULLS_EX_COUNT])
	try
	{
		return extract_impl(ctx, scan, n_in, stride, X, out, cap, n_out, nullptr);
	}
	catch (...)
	{
		return mulls::abi_caught(ctx); // nothing is thrown across the ABI
	}

	// ---- device-resident feature blocks ------------------------------------------------------------------------------------------------
	int mulls_block_create(mulls_ctx *ctx, mulls_block **out)
	try
	{
		if (!ctx || !out)
			return MULLS_E_INVALID;
		*out = new mulls_block();
		ctx->blocks.push_back(*out);
		return MULLS_OK;
	}
	catch (...)
	{
		return mulls::abi_caught(ctx);
	}
	void mulls_block_destroy(mulls_ctx *ctx, mulls_block *b)
	{
		if (!b)
			return;
		if (ctx)
		{
			(void)hipSetDevice(ctx->device);
			ctx->blocks.erase(std::remove(ctx->blocks.begin(), ctx->blocks.end(), b), ctx->blocks.end());
		}
		if (b->buf)
			(void)hipFree(b->buf);
		delete b;
	}
	int mulls_extract_features_resident(mulls_ctx *ctx, const void *scan, uint32_t n, uint32_t stride, const mulls_extract_params *params, mulls_block *block,
										uint32_t n_out[MULLS_EX_COUNT])
	try
	{
		if (!block)
			return MULLS_E_INVALID;
		void *no_out[MULLS_EX_COUNT] = {};
		uint32_t no_cap[MULLS_EX_COUNT] = {};
		return extract_impl(ctx, scan, n, stride, params, no_out, no_cap, n_out, block);
	}
	catch (...)
	{
		return mulls::abi_caught(ctx);
	}
	int mulls_block_cloud(mulls_ctx *ctx, const mulls_block *b, int which, mulls_cloud *out)
	{
		if (!ctx || !b || !out || which < 0 || which >= MULLS_EX_COUNT || which == MULLS_EX_RAW || which == MULLS_EX_DOWN)
			return MULLS_E_INVALID;
		out->pts = b->n[which] ? b->buf + b->off[which] : nullptr;
		out->n = b->n[which];
		out->stride = MULLS_POINT_BYTES;
		return MULLS_OK;
	}
	int mulls_block_download(mulls_ctx *ctx, const mulls_block *b, int which, void *pts, uint32_t cap, uint32_t *n)
	try
	{
		if (!ctx || !b || !n || which < 0 || which >= MULLS_EX_COUNT || (cap && !pts))
			return MULLS_E_INVALID;
		*n = b->n[which];
		const uint32_t k = std::min(cap, b->n[which]);
		if (k)
		{
			HIPCHK(ctx, hipSetDevice(ctx->device));
			HIPCHK(ctx, hipMemcpy(pts, b->buf + b->off[which], (size_t)k * MULLS_POINT_BYTES, hipMemcpyDeviceToHost));
		}
		return MULLS_OK;
	}
	catch (...)
	{
		return mulls::abi_caught(ctx); // nothing is thrown across the ABI
	}
}
