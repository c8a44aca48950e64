// ground.cpp — host side of mulls_ground_filter and mulls_extract_features (include/mulls_hip.h): CFilter::fast_ground_filter on the device
// (k_ground.hip), and extract_semantic_pts' chain scanner filter -> ground filter -> classify_nground_pts with the clouds staying on the device.
// Upload the scan, one launch, download the two clouds; cloud_ground_down is a subset of cloud_ground by index (every
// ground_random_down_down_rate-th point, or the ABI's seeded fixed-number selection: cfilter.hpp:1955-1968) and is taken on the way out.
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/mulls_hip.h"
#include "ctx.h"
#include "device_types.h"
#include "map_launch.h"

#include <hip/hip_vector_types.h>

struct GfOut // head of the device-side state (k_ground.hip: GfState)
{
	uint32_t n_ground, n_unground, n_high, error;
	uint32_t row, col;
	float mean_height;
	uint32_t n_cand;
};
int launch_ground_filter(hipStream_t st, const float4 *pts, uint32_t n, const mulls_ground_params &P, uint32_t *ids, uint16_t *cellof, uint8_t *code, float *d3v,
						 float4 *ground, float4 *unground, void *aux);
size_t ground_filter_aux_bytes(uint32_t n);
void launch_scanner_mask(hipStream_t st, const float4 *pts, uint32_t n, float self_radius, float ghost_radius, float z_min_ghost, float z_min_global, uint8_t *mask);
extern "C" __attribute__((visibility("hidden"))) int mulls_classify_impl(mulls_ctx *ctx, const void *pts, bool pts_on_device, uint32_t n_in, uint32_t stride, const mulls_classify_params *P,
								   void *const out[MULLS_CL_COUNT], const uint32_t cap[MULLS_CL_COUNT], uint32_t n_out[MULLS_CL_COUNT], void *cloud_in_after,
								   uint32_t *n_cloud_in_after);

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
		p->rng_seed = 0;
	}

	// ---- shared by mulls_ground_filter and mulls_extract_features -------------------------------------------------------------------
	struct GfArena // one device arena: scan | ground | unground | ids | d3v | cellof | code | state, counters, per-cell tables | (scanner filter:) scan copy, mask, scratch
	{
		size_t o_ground, o_unground, o_ids, o_d3, o_cell, o_code, o_aux, o_alt, o_mask, o_seg, total;
	};
	static GfArena gf_layout(uint32_t n, bool scanner)
	{
		GfArena a;
		const size_t rec = (size_t)n * MULLS_POINT_BYTES;
		a.o_ground = rec, a.o_unground = 2 * rec, a.o_ids = 3 * rec, a.o_d3 = a.o_ids + (size_t)n * 4, a.o_cell = a.o_d3 + (size_t)n * 4;
		a.o_code = a.o_cell + (size_t)n * 2;
		a.o_aux = (a.o_code + n + 255) & ~(size_t)255;
		a.o_alt = (a.o_aux + ground_filter_aux_bytes(n) + 255) & ~(size_t)255;
		a.o_mask = a.o_alt + (scanner ? rec : 0);
		a.o_seg = (a.o_mask + (scanner ? n : 0) + 255) & ~(size_t)255;
		a.total = a.o_seg + (scanner ? ((size_t)n / 4096 + 32) * 4 * 8 : 0);
		return a;
	}
	static int gf_check(mulls_ctx *ctx, const mulls_ground_params *P, uint32_t n)
	{
		if (P->estimate_ground_normal_method != 0)
		{
			ctx->err = "mulls_ground_filter: only estimate_ground_normal_method 0 is built (1 / 2: PCA normals, 3: PCL RANSAC per cell)";
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
	// the filter on the n records at `scan` (device, inside the arena); the stream is idle afterwards
	static int gf_run(mulls_ctx *ctx, const GfArena &a, const float4 *scan, uint32_t n, const mulls_ground_params *P, GfOut &out)
	{
		unsigned char *base = static_cast<unsigned char *>(ctx->gf_buf);
		hipStream_t st = ctx->stream;
		if (launch_ground_filter(st, scan, n, *P, reinterpret_cast<uint32_t *>(base + a.o_ids), reinterpret_cast<uint16_t *>(base + a.o_cell), base + a.o_code,
								 reinterpret_cast<float *>(base + a.o_d3), reinterpret_cast<float4 *>(base + a.o_ground), reinterpret_cast<float4 *>(base + a.o_unground),
								 base + a.o_aux) != 0)
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
		return MULLS_OK;
	}
	// cloud_ground (n_ground records at g) -> cloud_ground_down (cfilter.hpp:1955-1968), taken on the host from the downloaded cloud
	static uint32_t gf_ground_down(const mulls_ground_params *P, const unsigned char *g, uint32_t n_ground, void *ground_down, uint32_t cap_ground_down)
	{
		uint32_t nd = 0;
		unsigned char *gd = static_cast<unsigned char *>(ground_down);
		auto take = [&](uint32_t i) {
			if (nd < cap_ground_down)
				std::memcpy(gd + (size_t)nd * MULLS_POINT_BYTES, g + (size_t)i * MULLS_POINT_BYTES, MULLS_POINT_BYTES);
			nd++;
		};
		if (!P->fixed_num_downsampling)
		{
			for (uint32_t i = 0; i < n_ground; i++)
				if ((int)i % P->ground_random_down_down_rate == 0)
					take(i);
		}
		else
		{
			std::vector<uint8_t> mask(std::max<uint32_t>(n_ground, 1));
			thin_mask(mask.data(), n_ground, P->down_ground_fixed_num, P->rng_seed, 12);
			for (uint32_t i = 0; i < n_ground; i++)
				if (mask[i])
					take(i);
		}
		return nd;
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
		const GfArena a = gf_layout(n, false);
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
	}

	int mulls_extract_features(mulls_ctx *ctx, const void *scan, uint32_t n_in, uint32_t stride, const mulls_extract_params *X, void *const out[MULLS_EX_COUNT],
							   const uint32_t cap[MULLS_EX_COUNT], uint32_t n_out[MULLS_EX_COUNT])
	try
	{
		if (!ctx || !X || !out || !cap || !n_out || (n_in && !scan) || stride < MULLS_POINT_BYTES)
			return MULLS_E_INVALID;
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
		const bool scanner = X->apply_scanner_filter != 0;
		const GfArena a = gf_layout(n_in, scanner);
		if (gf_reserve(ctx, a) != MULLS_OK)
			return MULLS_E_HIP;
		unsigned char *base = static_cast<unsigned char *>(ctx->gf_buf);
		const size_t rec_in = (size_t)n_in * MULLS_POINT_BYTES;
		unsigned char *up = scanner ? base + a.o_alt : base; // the scanner filter compacts from the copy into the scan's place
		if (stride == MULLS_POINT_BYTES)
			HIPCHK(ctx, hipMemcpyAsync(up, scan, rec_in, hipMemcpyHostToDevice, st));
		else
			HIPCHK(ctx, hipMemcpy2DAsync(up, MULLS_POINT_BYTES, scan, stride, MULLS_POINT_BYTES, n_in, hipMemcpyHostToDevice, st));
		uint32_t n = n_in;
		if (scanner)
		{
			// scanner_filter (cfilter.hpp:914-929): ego-vehicle ring and underground ghost points
			uint32_t *counts = reinterpret_cast<uint32_t *>(base + a.o_seg);
			launch_scanner_mask(st, reinterpret_cast<const float4 *>(up), n_in, X->self_ring_radius, X->ghost_radius, X->z_min, X->z_min_min, base + a.o_mask);
			MapCompactArgs ca;
			std::memset(&ca, 0, sizeof(ca));
			ca.cloud[0].in = reinterpret_cast<const float4 *>(up);
			ca.cloud[0].out = reinterpret_cast<float4 *>(base);
			ca.cloud[0].mask = base + a.o_mask;
			ca.cloud[0].n = n_in;
			ca.out_n = counts;
			ca.mode = 0;
			launch_map_compact(st, ca, counts + 8);
			uint32_t c6[6];
			HIPCHK(ctx, hipMemcpyAsync(c6, counts, sizeof(c6), hipMemcpyDeviceToHost, st));
			HIPCHK(ctx, hipStreamSynchronize(st));
			n = c6[0];
		}
		n_out[MULLS_EX_RAW] = n;
		if (n == 0)
			return MULLS_OK;
		const uint32_t kr = std::min(n, cap[MULLS_EX_RAW]);
		if (kr)
			HIPCHK(ctx, hipMemcpyAsync(out[MULLS_EX_RAW], base, (size_t)kr * MULLS_POINT_BYTES, hipMemcpyDeviceToHost, st));
		GfOut go;
		const int rc = gf_run(ctx, a, reinterpret_cast<const float4 *>(base), n, P, go);
		if (rc != MULLS_OK)
			return rc;
		std::vector<unsigned char> g((size_t)go.n_ground * MULLS_POINT_BYTES);
		if (go.n_ground)
			HIPCHK(ctx, hipMemcpyAsync(g.data(), base + a.o_ground, g.size(), hipMemcpyDeviceToHost, st));
		// classify_nground_pts on the non-ground cloud where the filter left it
		const int rc2 = mulls_classify_impl(ctx, base + a.o_unground, true, go.n_unground, MULLS_POINT_BYTES, &X->classify, out + MULLS_EX_PILLAR, cap + MULLS_EX_PILLAR,
											n_out + MULLS_EX_PILLAR, cap[MULLS_EX_UNGROUND] >= go.n_unground ? out[MULLS_EX_UNGROUND] : nullptr, &n_out[MULLS_EX_UNGROUND]);
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
	catch (...)
	{
		return mulls::abi_caught(ctx); // nothing is thrown across the ABI
	}
}
