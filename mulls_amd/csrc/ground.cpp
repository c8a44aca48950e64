// ground.cpp — host side of mulls_ground_filter (include/mulls_hip.h): CFilter::fast_ground_filter on the device (k_ground.hip).
// Upload the scan, one launch, download the two clouds; cloud_ground_down is a subset of cloud_ground by index (every
// ground_random_down_down_rate-th point, or the ABI's seeded fixed-number selection: cfilter.hpp:1955-1968) and is taken on the way out.
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/mulls_hip.h"
#include "ctx.h"
#include "device_types.h"

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

	int mulls_ground_filter(mulls_ctx *ctx, const void *pts, uint32_t n, uint32_t stride, const mulls_ground_params *P, void *ground, uint32_t cap_ground,
							void *ground_down, uint32_t cap_ground_down, void *unground, uint32_t cap_unground, uint32_t n_out[3])
	try
	{
		if (!ctx || !P || !n_out || (n && !pts) || stride < MULLS_POINT_BYTES || (cap_ground && !ground) || (cap_ground_down && !ground_down) ||
			(cap_unground && !unground))
			return MULLS_E_INVALID;
		n_out[0] = n_out[1] = n_out[2] = 0;
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
		if (n == 0)
			return MULLS_OK; // (the reference divides by a zero sample count here)
		if (n > 500000u)
		{
			ctx->err = "mulls_ground_filter: more than 500000 points in one scan";
			return MULLS_E_UNSUPPORTED;
		}
		HIPCHK(ctx, hipSetDevice(ctx->device));
		hipStream_t st = ctx->stream;
		// one arena: scan | ground | unground | ids | d3v | cellof | code | state, block / segment counters, per-cell tables
		const size_t rec = (size_t)n * MULLS_POINT_BYTES;
		const size_t o_ground = rec, o_unground = 2 * rec, o_ids = 3 * rec, o_d3 = o_ids + (size_t)n * 4, o_cell = o_d3 + (size_t)n * 4,
					 o_code = o_cell + (size_t)n * 2, o_aux = (o_code + n + 255) & ~(size_t)255, total = o_aux + ground_filter_aux_bytes(n);
		if (ctx->gf_cap < total)
		{
			if (ctx->gf_buf)
				(void)hipFree(ctx->gf_buf);
			ctx->gf_buf = nullptr;
			ctx->gf_cap = 0;
			HIPCHK(ctx, hipMalloc(&ctx->gf_buf, total + total / 4));
			ctx->gf_cap = total + total / 4;
		}
		unsigned char *base = static_cast<unsigned char *>(ctx->gf_buf);
		if (stride == MULLS_POINT_BYTES)
			HIPCHK(ctx, hipMemcpyAsync(base, pts, rec, hipMemcpyHostToDevice, st));
		else
			HIPCHK(ctx, hipMemcpy2DAsync(base, MULLS_POINT_BYTES, pts, stride, MULLS_POINT_BYTES, n, hipMemcpyHostToDevice, st));
		if (launch_ground_filter(st, reinterpret_cast<const float4 *>(base), n, *P, reinterpret_cast<uint32_t *>(base + o_ids), reinterpret_cast<uint16_t *>(base + o_cell),
								 base + o_code, reinterpret_cast<float *>(base + o_d3), reinterpret_cast<float4 *>(base + o_ground),
								 reinterpret_cast<float4 *>(base + o_unground), base + o_aux) != 0)
		{
			ctx->err = "mulls_ground_filter: launch failed";
			return MULLS_E_HIP;
		}
		GfOut out;
		HIPCHK(ctx, hipMemcpyAsync(&out, base + o_aux, sizeof(out), hipMemcpyDeviceToHost, st));
		HIPCHK(ctx, hipStreamSynchronize(st));
		if (out.error)
		{
			ctx->err = "mulls_ground_filter: the grid has too many cells (more than 65536, or more than 64 M table entries: grid_resolution too fine for this scan's extent)";
			return MULLS_E_UNSUPPORTED;
		}
		std::vector<unsigned char> g((size_t)out.n_ground * MULLS_POINT_BYTES);
		if (out.n_ground)
			HIPCHK(ctx, hipMemcpyAsync(g.data(), base + o_ground, g.size(), hipMemcpyDeviceToHost, st));
		const uint32_t ku = std::min(out.n_unground, cap_unground);
		if (ku)
			HIPCHK(ctx, hipMemcpyAsync(unground, base + o_unground, (size_t)ku * MULLS_POINT_BYTES, hipMemcpyDeviceToHost, st));
		HIPCHK(ctx, hipStreamSynchronize(st));
		const uint32_t kg = std::min(out.n_ground, cap_ground);
		if (kg)
			std::memcpy(ground, g.data(), (size_t)kg * MULLS_POINT_BYTES);
		// cloud_ground_down (cfilter.hpp:1955-1968)
		uint32_t nd = 0;
		unsigned char *gd = static_cast<unsigned char *>(ground_down);
		auto take = [&](uint32_t i) {
			if (nd < cap_ground_down)
				std::memcpy(gd + (size_t)nd * MULLS_POINT_BYTES, g.data() + (size_t)i * MULLS_POINT_BYTES, MULLS_POINT_BYTES);
			nd++;
		};
		if (!P->fixed_num_downsampling)
		{
			for (uint32_t i = 0; i < out.n_ground; i++)
				if ((int)i % P->ground_random_down_down_rate == 0)
					take(i);
		}
		else
		{
			std::vector<uint8_t> mask(std::max<uint32_t>(out.n_ground, 1));
			thin_mask(mask.data(), out.n_ground, P->down_ground_fixed_num, P->rng_seed, 12);
			for (uint32_t i = 0; i < out.n_ground; i++)
				if (mask[i])
					take(i);
		}
		n_out[0] = out.n_ground;
		n_out[1] = nd;
		n_out[2] = out.n_unground;
		return MULLS_OK;
	}
	catch (...)
	{
		return mulls::abi_caught(ctx); // nothing is thrown across the ABI
	}
}
