// crop_grid.h — the intersection box and the grid descriptor of a cropped target class cloud: shared by k_crop (k_setup.hip) and the fused
// crop + grid build of the LDS tier (k_tgt_grid, k_grid.hip)
#pragma once
#include "device_util.h"

// the intersection box of get_cloud_pair_intersection (cfilter.hpp:2613-2655): union box of the transformed source clouds
// (ordered keys from k_clone_src) against block1->local_bound, padded by 1 m
namespace
{
__device__ __forceinline__ void crop_box(uint32_t pair, const uint32_t *__restrict__ bbox, const PairSetup *__restrict__ setup, double lo[3],
										  double hi[3])
{
	for (int k = 0; k < 3; k++)
	{
		uint32_t kmin = bbox[pair * 6 + k], kmax = bbox[pair * 6 + 3 + k];
		// an empty union keeps (+DBL_MAX, -DBL_MAX) like CloudUtility::merge_bbx (utility.hpp:867-884)
		double mmin = (kmin == 0xffffffffu && kmax == 0u) ? 1.7976931348623157e308 : (double)ord2f(kmin);
		double mmax = (kmin == 0xffffffffu && kmax == 0u) ? -1.7976931348623157e308 : (double)ord2f(kmax);
		double b1min = setup[pair].tgt_bound[k], b1max = setup[pair].tgt_bound[3 + k];
		const float pad = 1.0f;
		lo[k] = ((b1min > mmin) ? b1min : mmin) - pad; // get_intersection_bbx, utility.hpp:857-865
		hi[k] = ((b1max < mmax) ? b1max : mmax) + pad;
	}
}
// strict inequalities, float coordinate promoted to double (cfilter.hpp:959-961)
__device__ __forceinline__ bool crop_keep(const float4 &p, const double lo[3], const double hi[3])
{
	return (double)p.x > lo[0] && (double)p.x < hi[0] && (double)p.y > lo[1] && (double)p.y < hi[1] && (double)p.z > lo[2] && (double)p.z < hi[2];
}
// cells along one axis for an extent and a cell edge — the same float expression as grid_cell(), so that the largest
// coordinate lands in the last cell; an absurd extent (the bounding box only sees coordinates within 1e18 m)
// saturates instead of overflowing the conversion, and the caller then grows the cell edge until the grid fits
__device__ __forceinline__ uint32_t grid_dim(float extent, float inv_h)
{
	const float c = floorf(extent * inv_h);
	return c >= 0.0f ? (uint32_t)fminf(c, 4.0e9f) + 1u : 1u; // also false for NaN
}
// grid descriptor of a cropped target cloud with bounding box [lo3, hi3] and `running` points, of the kind its tier asks for (CloudDesc::tier)
__device__ __forceinline__ GridDesc make_grid(const float lo3[3], const float hi3[3], uint32_t running, const RunParams &rp, const CloudDesc &d, uint32_t cls)
{
	GridDesc g;
	g.ox = lo3[0], g.oy = lo3[1], g.oz = lo3[2];
	g.h = rp.grid_h0;
	g.nx = g.ny = g.nz = 1;
	g.ncell = 0;
	g.wpr = 1;
	g.nocc = 0;
	uint32_t nwords = 1;
	const bool bitmap = d.tier == MULLS_TIER_BM;
	if (!rp.used[cls])
		running = 0; // a class the run does not search gets no grid (its cloud is cropped for the sizes the reference reports)
	if (running > 0 && bitmap)
	{
		// global-memory tier: occupancy bitmap over fine cells; rows are padded to whole 64-cell words
		g.h = rp.bm_h0;
		if (rp.bm_auto) // points lie on surfaces: mean spacing ~ sqrt(footprint / count); measured optimum 0.25 m (1 M points) .. 0.7 m (5 k)
			g.h = fminf(fmaxf(sqrtf((hi3[0] - lo3[0]) * (hi3[1] - lo3[1]) / (float)running), rp.bm_h0), 2.8f * rp.bm_h0);
		for (;;)
		{
			g.inv_h = 1.0f / g.h;
			g.nx = grid_dim(hi3[0] - g.ox, g.inv_h);
			g.ny = grid_dim(hi3[1] - g.oy, g.inv_h);
			g.nz = grid_dim(hi3[2] - g.oz, g.inv_h);
			g.wpr = (g.nx + 63u) >> 6;
			if ((unsigned long long)g.ny * g.nz * g.wpr <= (unsigned long long)rp.bm_maxwords || g.ny * g.nz * g.wpr <= 1u)
				break;
			g.h *= 1.25f;
		}
		nwords = g.ny * g.nz * g.wpr;
	}
	else if (running > 0)
		for (;;)
		{
			g.inv_h = 1.0f / g.h;
			// same float expression as grid_cell() so that the largest coordinate lands in the last cell
			g.nx = grid_dim(hi3[0] - g.ox, g.inv_h);
			g.ny = grid_dim(hi3[1] - g.oy, g.inv_h);
			g.nz = grid_dim(hi3[2] - g.oz, g.inv_h);
			if ((unsigned long long)g.nx * g.ny * g.nz <= (unsigned long long)rp.grid_maxcells || g.nx * g.ny * g.nz <= 1u)
				break;
			g.h *= 1.25f;
		}
	else
		g.inv_h = 1.0f;
	// cell tables exist for the USED classes only; the host handed out their slots (assign_tiers)
	g.ncell = running > 0 ? (bitmap ? nwords : g.nx * g.ny * g.nz) : 0u;
	g.cell_off = d.grid_slot * (bitmap ? rp.bm_stride : rp.cell_stride);
	return g;
}
} // namespace
