// classify.cpp — host side of mulls_classify_nground (include/mulls_hip.h): CFilter::classify_nground_pts (cfilter.hpp:2058-2290) on the
// device (k_classify.hip; stable compactions: map_kernels.hip).  What the host does besides moving data is selection bookkeeping: the seeded
// fixed-number selections (upstream: pcl::RandomSample), the sector buckets of xy_normal_balanced_downsample on the few hundred points that
// reach it, and the one step upstream hands to this toolchain's std::sort — non_max_suppress's visiting order — for which the keys come
// back, are sorted by the same std::sort (so that equal keys fall as they fall upstream), and return as a permutation.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/mulls_hip.h"
#include "classify_launch.h"
#include "ctx.h"
#include "map_launch.h"

namespace
{
const size_t REC = MULLS_POINT_BYTES;
typedef std::vector<unsigned char> Bytes;

struct Bump // carve device arrays out of one arena
{
	unsigned char *base;
	size_t off;
	template <typename T>
	T *take(size_t count)
	{
		off = (off + 255u) & ~(size_t)255u;
		T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
		off += count * sizeof(T);
		return p;
	}
};

struct Layout
{
	ClArrays A;
	float4 *cls_a[6];	  // pillar, pillar (promoted), beam, beam (promoted), facade, roof: compaction targets, pillar / beam become the class clouds
	float4 *cls_sorted[4]; // class clouds in visiting order
	float4 *down[4], *vertex;
	uint32_t *nms_list[4], *nms_cnt[4], *nms_off[4], *nms_wcur[4], *nms_pool;
	unsigned long long *nms_pool_used;
	uint32_t nms_pool_cap;
	uint8_t *keep[4];
	float *keys;	// [4][n]
	uint32_t *perm; // [4][n]
	uint32_t *counts; // [16]
	uint32_t *seg;	  // compaction scratch
	size_t seg_cap;
	size_t total;
};

// n: points the passes work on; n_in >= n: points of the cloud as handed in (the device-side thinning needs a mask and compaction scratch of that size)
Layout carve(unsigned char *base, uint32_t n, uint32_t K, uint32_t n_in)
{
	Layout L;
	Bump b{base, 0};
	ClArrays &A = L.A;
	A.recs = b.take<float4>((size_t)n * 3);
	A.sorted = b.take<float4>(n);
	A.cellof = b.take<uint32_t>(n);
	A.cell_start = b.take<uint32_t>((size_t)MULLS_CL_MAX_CELLS + 1u);
	A.cell_fill = b.take<uint32_t>((size_t)MULLS_CL_MAX_CELLS + 1u);
	A.seg_sum = b.take<uint32_t>(1032);
	A.nbr = b.take<uint32_t>((size_t)n * K);
	A.closebits = b.take<unsigned long long>(n);
	A.f_cnt = b.take<int32_t>(n);
	A.cov = b.take<float>((size_t)n * 6);
	A.f_curv = b.take<double>(n);
	A.f_lin = b.take<double>(n);
	A.f_pla = b.take<double>(n);
	A.f_pd = b.take<float4>(n);
	A.f_nd = b.take<float4>(n);
	A.lab = b.take<uint8_t>(n);
	A.plab = b.take<uint8_t>(n);
	A.cstate = b.take<uint8_t>(n);
	A.cand = b.take<uint8_t>(n);
	A.down = b.take<uint8_t>(n);
	A.mask = b.take<uint8_t>(std::max<size_t>((size_t)n * 11, n_in));
	A.vtx = b.take<float4>((size_t)n * 3);
	A.round_cnt = b.take<uint32_t>(64);
	A.grid = b.take<ClGrid>(1);
	for (int k = 0; k < 6; k++)
		L.cls_a[k] = b.take<float4>((size_t)n * 3);
	for (int k = 0; k < 4; k++)
	{
		L.cls_sorted[k] = b.take<float4>((size_t)n * 3);
		L.down[k] = b.take<float4>((size_t)n * 3);
		L.nms_list[k] = b.take<uint32_t>((size_t)n * MULLS_CL_NMS_CAP);
		L.nms_cnt[k] = b.take<uint32_t>(n);
		L.nms_off[k] = b.take<uint32_t>(n);
		L.nms_wcur[k] = b.take<uint32_t>(n);
		L.keep[k] = b.take<uint8_t>(n);
	}
	L.vertex = b.take<float4>((size_t)n * 3);
	L.nms_pool_cap = (uint32_t)std::min<size_t>((size_t)n * 64u, 0x7fffffffu);
	L.nms_pool = b.take<uint32_t>(L.nms_pool_cap);
	L.nms_pool_used = b.take<unsigned long long>(1);
	L.keys = b.take<float>((size_t)n * 4);
	L.perm = b.take<uint32_t>((size_t)n * 4);
	L.counts = b.take<uint32_t>(16);
	L.seg_cap = (size_t)6 * ((std::max(n, n_in) + 4095u) / 4096u + 1u) + 16;
	L.seg = b.take<uint32_t>(L.seg_cap);
	L.total = b.off + 256;
	return L;
}

// random_downsample_pcl (cfilter.hpp:606-628) on raw records, with the ABI's seeded selection
void random_downsample(Bytes &c, int keep_number, uint64_t seed, int cloud_id)
{
	const uint32_t n = (uint32_t)(c.size() / REC);
	if (keep_number < 0 || (long)n <= (long)keep_number) // size_t comparison upstream (:608): a negative count keeps every point
		return;
	std::vector<uint8_t> mask(n);
	thin_mask(mask.data(), n, keep_number, seed, cloud_id);
	size_t w = 0;
	for (uint32_t i = 0; i < n; i++)
		if (mask[i])
		{
			if (w != i)
				std::memmove(c.data() + w * REC, c.data() + (size_t)i * REC, REC);
			w++;
		}
	c.resize(w * REC);
}
// xy_normal_balanced_downsample (cfilter.hpp:551-602): sectors by the normal's azimuth (std::atan2 of this platform's libm, as upstream)
void xy_normal_balanced_downsample(Bytes &c, int keep_number_per_sector, int sector_num, uint64_t seed, int cloud_id)
{
	const uint32_t n = (uint32_t)(c.size() / REC);
	if (keep_number_per_sector < 0 || (long)n <= (long)keep_number_per_sector) // size_t comparison upstream (:555)
		return;
	std::vector<Bytes> sectors(sector_num);
	const double angle_per_sector = 360.0 / sector_num;
	for (uint32_t i = 0; i < n; i++)
	{
		float nrm[2];
		std::memcpy(nrm, c.data() + (size_t)i * REC + 16, sizeof(nrm));
		double ang = std::atan2(nrm[1], nrm[0]);
		if (ang < 0)
			ang += 2 * M_PI;
		ang *= (180.0 / M_PI);
		int sector_id = (int)(ang / angle_per_sector);
		if (sector_id >= sector_num) // -tiny + 2 pi rounds to 2 pi: upstream indexes past the last sector
			sector_id = sector_num - 1;
		if (sector_id < 0)
			sector_id = 0;
		sectors[sector_id].insert(sectors[sector_id].end(), c.data() + (size_t)i * REC, c.data() + (size_t)(i + 1) * REC);
	}
	c.clear();
	for (int j = 0; j < sector_num; j++)
	{
		random_downsample(sectors[j], keep_number_per_sector, seed, cloud_id + j);
		c.insert(c.end(), sectors[j].begin(), sectors[j].end());
	}
}

struct KeyIdx
{
	float key;
	uint32_t idx;
};
// Rounds of a settle-until-nothing-is-undecided loop (the promotion of vertex candidates, the suppression rounds): launch(slot) runs one round that adds the points
// it leaves undecided to round_cnt[slot].  A round over a settled state changes nothing, so rounds are launched in batches and the counters read once per batch — the
// first batch as long as the previous call's loop turned out to be (+ 2: consecutive frames need about the same), further ones `step` rounds: one wait per loop instead
// of one per `step` rounds (28 us each on the frame path).  All 64 counters come down, so the round that settled is known and the hint follows the data both ways.
template <class Launch>
int run_rounds(mulls_ctx *ctx, hipStream_t st, uint32_t *round_cnt, uint32_t step, uint32_t *hint, const Launch &launch, const char *what)
{
	uint32_t h[64];
	// a multiple of `step` (4 or 8; 32 is one of both), so that `round` stays one and the reset below meets every multiple of 64
	const uint32_t first = std::min(32u, std::max(step, (*hint + 2u + step - 1u) / step * step));
	for (uint32_t round = 0;;)
	{
		const uint32_t lo = round, batch = round == 0 ? first : step;
		for (uint32_t r = 0; r < batch; r++, round++)
			launch(round & 63u);
		HIPCHK(ctx, hipMemcpyAsync(h, round_cnt, sizeof(h), hipMemcpyDeviceToHost, st));
		HIPCHK(ctx, hipStreamSynchronize(st));
		if (h[(round - 1u) & 63u] == 0)
		{
			uint32_t needed = round;
			for (uint32_t r = lo; r < round; r++)
				if (h[r & 63u] == 0)
				{
					needed = r + 1u;
					break;
				}
			*hint = needed;
			return MULLS_OK;
		}
		if ((round & 63u) == 0)
			HIPCHK(ctx, hipMemsetAsync(round_cnt, 0, 64 * 4, st));
		if (round > (1u << 22))
		{
			ctx->err = what;
			return MULLS_E_HIP;
		}
	}
}
} // namespace

extern "C"
{
	void mulls_classify_default_params(mulls_classify_params *p)
	{
		if (!p)
			return;
		std::memset(p, 0, sizeof(*p));
		// what extract_semantic_pts (cfilter.hpp:2301-2318) passes for test/mulls_reg.cpp with script/run_mulls_reg.sh's flags
		p->neighbor_searching_radius = 1.0f;
		p->neighbor_k = 50;
		p->neigh_k_min = 8;
		p->pca_down_rate = 1;
		p->edge_thre = 0.65f;
		p->planar_thre = 0.65f;
		p->edge_thre_down = 0.75f;
		p->planar_thre_down = 0.75f;
		p->extract_vertex_points_method = 2;
		p->curvature_thre = 0.10f;
		p->vertex_curvature_non_max_radius = 1.5f;
		p->linear_vertical_sin_high_thre = 0.94f;
		p->linear_vertical_sin_low_thre = 0.17f;
		p->planar_vertical_sin_high_thre = 0.98f;
		p->planar_vertical_sin_low_thre = 0.34f;
		p->sharpen_with_nms = 1;
		p->pillar_down_fixed_num = 200;
		p->facade_down_fixed_num = 800;
		p->beam_down_fixed_num = 200;
		p->roof_down_fixed_num = 200;
		p->unground_down_fixed_num = 20000;
		p->beam_height_max = FLT_MAX;
		p->roof_height_min = 0.0f;
		p->feature_pts_ratio_guess = 0.3f;
	}

	// pts_on_device: `pts` is device memory of this context's device holding packed 48-byte records (mulls_extract_features)
	// dev_out: leave the clouds on the device (ClassifyDev, ctx.h) instead of copying them to out[] (which may then hold nulls)
	__attribute__((visibility("hidden"))) int mulls_classify_impl(mulls_ctx *ctx, const void *pts, bool pts_on_device, uint32_t n_in, uint32_t stride, const mulls_classify_params *P,
							void *const out[MULLS_CL_COUNT], const uint32_t cap[MULLS_CL_COUNT], uint32_t n_out[MULLS_CL_COUNT], void *cloud_in_after,
							uint32_t *n_cloud_in_after, ClassifyDev *dev_out)
	{
		if (!ctx || !P || !out || !cap || !n_out || (n_in && !pts) || stride < MULLS_POINT_BYTES || (pts_on_device && stride != MULLS_POINT_BYTES))
			return MULLS_E_INVALID;
		for (int k = 0; k < MULLS_CL_COUNT; k++)
		{
			n_out[k] = 0;
			if (cap[k] && !out[k])
				return MULLS_E_INVALID;
		}
		if (P->pca_down_rate < 1 || P->neighbor_k < 1 || P->neighbor_k > (int)MULLS_CL_MAX_K || !(P->neighbor_searching_radius > 0.0f))
		{
			ctx->err = "mulls_classify_nground: pca_down_rate >= 1, 1 <= neighbor_k <= 64 and a positive radius are required";
			return MULLS_E_INVALID;
		}
		if (n_in > 2000000u)
		{
			ctx->err = "mulls_classify_nground: more than 2000000 points in one cloud";
			return MULLS_E_UNSUPPORTED;
		}
		// random_downsample_pcl(cloud_in, unground_down_fixed_num) (:2088-2089)
		std::vector<uint8_t> in_mask;
		uint32_t n = n_in;
		if (P->fixed_num_downsampling && (long)n_in > (long)P->unground_down_fixed_num)
		{
			in_mask.resize(n_in);
			n = thin_mask(in_mask.data(), n_in, P->unground_down_fixed_num, P->rng_seed, 30);
		}
		if (n_cloud_in_after)
			*n_cloud_in_after = n;
		if (n == 0)
			return MULLS_OK;
		Bytes packed;
		const unsigned char *src = static_cast<const unsigned char *>(pts);
		if (!pts_on_device && (stride != REC || !in_mask.empty()))
		{
			packed.resize((size_t)n * REC);
			size_t w = 0;
			for (uint32_t i = 0; i < n_in; i++)
				if (in_mask.empty() || in_mask[i])
					std::memcpy(packed.data() + (w++) * REC, src + (size_t)i * stride, REC);
			src = packed.data();
		}
		HIPCHK(ctx, hipSetDevice(ctx->device));
		hipStream_t st = ctx->stream;
		const uint32_t K = (uint32_t)P->neighbor_k;
		const size_t total = carve(nullptr, n, K, n_in).total;
		if (ctx->cl_cap < total)
		{
			if (ctx->cl_buf)
				(void)hipFree(ctx->cl_buf);
			ctx->cl_buf = nullptr;
			ctx->cl_cap = 0;
			HIPCHK(ctx, hipMalloc(&ctx->cl_buf, total + total / 4));
			ctx->cl_cap = total + total / 4;
		}
		Layout L = carve(static_cast<unsigned char *>(ctx->cl_buf), n, K, n_in);
		const ClArrays &A = L.A;
		if (!pts_on_device)
			HIPCHK(ctx, hipMemcpyAsync(A.recs, src, (size_t)n * REC, hipMemcpyHostToDevice, st));
		else if (in_mask.empty())
			HIPCHK(ctx, hipMemcpyAsync(A.recs, src, (size_t)n * REC, hipMemcpyDeviceToDevice, st));
		else
		{
			// the fixed-number thinning of a cloud that is already on the device: the selection mask goes up, a stable compaction applies it
			HIPCHK(ctx, hipMemcpyAsync(A.mask, in_mask.data(), n_in, hipMemcpyHostToDevice, st));
			MapCompactArgs ta;
			std::memset(&ta, 0, sizeof(ta));
			ta.cloud[0].in = reinterpret_cast<const float4 *>(src);
			ta.cloud[0].out = A.recs;
			ta.cloud[0].mask = A.mask;
			ta.cloud[0].n = n_in;
			ta.out_n = L.counts;
			ta.mode = 0;
			launch_map_compact(st, ta, L.seg);
			HIPCHK(ctx, hipStreamSynchronize(st)); // in_mask is a host vector
		}
		// features of points that are not queried (pca_down_rate > 1) are pca_feature_t's zeros
		// (closebits .. f_nd are carved back to back: one fill)
		HIPCHK(ctx, hipMemsetAsync(A.closebits, 0, (size_t)(reinterpret_cast<unsigned char *>(A.f_nd + n) - reinterpret_cast<unsigned char *>(A.closebits)), st));
		HIPCHK(ctx, hipMemsetAsync(A.round_cnt, 0, 64 * 4, st));

		ClParams Q;
		std::memset(&Q, 0, sizeof(Q));
		Q.n = n, Q.K = K;
		Q.down_rate = P->pca_down_rate, Q.k_min = P->neigh_k_min;
		Q.radius = P->neighbor_searching_radius;
		Q.adaptive = P->use_distance_adaptive_pca ? 1 : 0;
		Q.unit_distance = 30.0f; // :2098
		Q.edge_thre = P->edge_thre, Q.planar_thre = P->planar_thre, Q.edge_thre_down = P->edge_thre_down, Q.planar_thre_down = P->planar_thre_down;
		Q.lin_high = P->linear_vertical_sin_high_thre, Q.lin_low = P->linear_vertical_sin_low_thre;
		Q.pla_high = P->planar_vertical_sin_high_thre, Q.pla_low = P->planar_vertical_sin_low_thre;
		Q.beam_height_max = P->beam_height_max, Q.roof_height_min = P->roof_height_min;
		Q.nms = P->sharpen_with_nms ? 1 : 0;
		Q.vertex_method = (P->curvature_thre < 1e-8) ? 0 : P->extract_vertex_points_method; // :2161-2162
		Q.curvature_thre = P->curvature_thre;
		Q.vertex_ratio_thre = P->feature_pts_ratio_guess / P->pca_down_rate;									  // :2167
		Q.min_curvature = (float)(0.3 * P->curvature_thre);													  // :2210
		Q.min_neighbor_feature_pts = (int)(P->feature_pts_ratio_guess / P->pca_down_rate * P->neighbor_k) - 1; // :2206

		launch_cl_grid(st, A, Q);
		launch_cl_pca(st, A, Q);
		launch_cl_label(st, A, Q);
		if (Q.vertex_method == 2)
		{
			const int rcr = run_rounds(ctx, st, A.round_cnt, 4u, &ctx->cl_rounds_hint[0], [&](uint32_t slot) { launch_cl_promote_round(st, A, Q, slot); },
									   "mulls_classify_nground: the promotion loop did not settle");
			if (rcr != MULLS_OK)
				return rcr;
		}
		launch_cl_encode_and_masks(st, A, Q);
		// stable compactions: the class clouds (first-pass members, then promoted ones), the key points, the *_down clouds of sharpen_with_nms = 0
		MapCompactArgs ca;
		std::memset(&ca, 0, sizeof(ca));
		for (int k = 0; k < 6; k++)
		{
			ca.cloud[k].in = A.recs;
			ca.cloud[k].out = L.cls_a[k];
			ca.cloud[k].mask = A.mask + (size_t)k * n;
			ca.cloud[k].n = n;
		}
		ca.out_n = L.counts;
		ca.mode = 0;
		launch_map_compact(st, ca, L.seg);
		std::memset(&ca, 0, sizeof(ca));
		ca.cloud[0].in = A.vtx;
		ca.cloud[0].out = L.vertex;
		ca.cloud[0].mask = A.mask + (size_t)6 * n;
		ca.cloud[0].n = n;
		if (!Q.nms)
			for (int k = 0; k < 4; k++)
			{
				ca.cloud[1 + k].in = A.recs;
				ca.cloud[1 + k].out = L.down[k];
				ca.cloud[1 + k].mask = A.mask + (size_t)(7 + k) * n;
				ca.cloud[1 + k].n = n;
			}
		ca.out_n = L.counts + 6;
		ca.mode = 0;
		launch_map_compact(st, ca, L.seg);
		uint32_t cnt[12];
		ClGrid grid;
		HIPCHK(ctx, hipMemcpyAsync(cnt, L.counts, sizeof(cnt), hipMemcpyDeviceToHost, st));
		HIPCHK(ctx, hipMemcpyAsync(&grid, A.grid, sizeof(grid), hipMemcpyDeviceToHost, st));
		HIPCHK(ctx, hipStreamSynchronize(st));
		if (grid.nonfinite)
		{
			ctx->err = "mulls_classify_nground: the cloud has non-finite coordinates";
			return MULLS_E_INVALID;
		}
		// class clouds: pillar = [first pass | promoted], beam likewise (the promotion loop pushes after the first loop has finished)
		uint32_t ncls[4] = {cnt[0] + cnt[1], cnt[2] + cnt[3], cnt[4], cnt[5]};
		float4 *cls[4] = {L.cls_a[0], L.cls_a[2], L.cls_a[4], L.cls_a[5]};
		if (cnt[1])
			HIPCHK(ctx, hipMemcpyAsync(L.cls_a[0] + (size_t)cnt[0] * 3, L.cls_a[1], (size_t)cnt[1] * REC, hipMemcpyDeviceToDevice, st));
		if (cnt[3])
			HIPCHK(ctx, hipMemcpyAsync(L.cls_a[2] + (size_t)cnt[2] * 3, L.cls_a[3], (size_t)cnt[3] * REC, hipMemcpyDeviceToDevice, st));
		uint32_t ndown[4] = {cnt[7], cnt[8], cnt[9], cnt[10]};
		const uint32_t nvertex = cnt[6];
		if (Q.nms)
		{
			// non_max_suppress(cloud, cloud_down, 0.25 * radius) for the classes whose *_down_fixed_num is positive and that have 10 points
			const int fixed_num[4] = {P->pillar_down_fixed_num, P->beam_down_fixed_num, P->facade_down_fixed_num, P->roof_down_fixed_num};
			const float nms_radius = (float)(0.25 * P->neighbor_searching_radius);
			ClNmsArgs na;
			std::memset(&na, 0, sizeof(na));
			na.r2 = (float)((double)nms_radius * (double)nms_radius);
			bool any = false;
			for (int c = 0; c < 4; c++)
			{
				ndown[c] = 0;
				if (fixed_num[c] > 0 && ncls[c] >= 10)
				{
					launch_cl_keys(st, cls[c], ncls[c], L.keys + (size_t)c * n);
					any = true;
				}
			}
			if (any)
			{
				// keys down, visiting orders up: through the context's pinned scratch (a pageable vector is staged by the runtime, and 2 MB of it are zero-filled per frame)
				if (grow_pinned(ctx, &ctx->cl_pin, &ctx->cl_pin_cap, (size_t)n * 32, hipHostMallocDefault) != MULLS_OK)
					return MULLS_E_HIP;
				float *keys = reinterpret_cast<float *>(ctx->cl_pin);
				uint32_t *perm = reinterpret_cast<uint32_t *>(ctx->cl_pin + (size_t)n * 16);
				for (int c = 0; c < 4; c++)
					if (fixed_num[c] > 0 && ncls[c] >= 10)
						HIPCHK(ctx, hipMemcpyAsync(keys + (size_t)c * n, L.keys + (size_t)c * n, (size_t)ncls[c] * 4, hipMemcpyDeviceToHost, st));
				HIPCHK(ctx, hipStreamSynchronize(st));
				// the four classes' visiting orders, one host thread each (the sorts are the frame path's longest host step)
				// (the context's sleeping thread pool, not an OpenMP team whose threads go on spinning into the kernels that follow: HostPool, ctx.h)
				const std::function<void(long)> sort_class = [&](long c) {
					if (!(fixed_num[c] > 0 && ncls[c] >= 10))
						return;
					std::vector<KeyIdx> ki(ncls[c]);
					for (uint32_t i = 0; i < ncls[c]; i++)
						ki[i] = KeyIdx{keys[(size_t)c * n + i], i};
					// the comparator of cfilter.hpp:1255; the permutation std::sort leaves depends on keys and count only
					std::sort(ki.begin(), ki.end(), [](const KeyIdx &a, const KeyIdx &b) { return a.key > b.key; });
					for (uint32_t i = 0; i < ncls[c]; i++)
						perm[(size_t)c * n + i] = ki[i].idx;
				};
				shared_host_pool().parallel_for(0, 4, 1, sort_class);
				for (int c = 0; c < 4; c++)
				{
					if (!(fixed_num[c] > 0 && ncls[c] >= 10))
						continue;
					HIPCHK(ctx, hipMemcpyAsync(L.perm + (size_t)c * n, perm + (size_t)c * n, (size_t)ncls[c] * 4, hipMemcpyHostToDevice, st));
					launch_cl_gather(st, cls[c], L.perm + (size_t)c * n, L.cls_sorted[c], ncls[c]);
					cls[c] = L.cls_sorted[c]; // std::sort works on cloud_in itself: the class cloud stays in this order
					na.recs[c] = L.cls_sorted[c];
					na.n[c] = ncls[c];
					na.keep[c] = L.keep[c];
					na.list[c] = L.nms_list[c];
					na.cnt[c] = L.nms_cnt[c];
					na.off[c] = L.nms_off[c];
					na.wcur[c] = L.nms_wcur[c];
				}
				na.pool = L.nms_pool, na.pool_used = L.nms_pool_used, na.pool_cap = L.nms_pool_cap;
				launch_cl_nms_lists(st, na, A.round_cnt); // (zeroes the pool cursor and the round counters with its own arrays)
				const int rcn = run_rounds(ctx, st, A.round_cnt, 8u, &ctx->cl_rounds_hint[1], [&](uint32_t slot) { launch_cl_nms_round(st, na, A.round_cnt + slot); },
										   "mulls_classify_nground: the suppression rounds did not settle");
				if (rcn != MULLS_OK)
					return rcn;
				std::memset(&ca, 0, sizeof(ca));
				for (int c = 0; c < 4; c++)
				{
					ca.cloud[c].in = na.recs[c];
					ca.cloud[c].out = L.down[c];
					ca.cloud[c].mask = na.keep[c];
					ca.cloud[c].n = na.n[c];
				}
				ca.out_n = L.counts;
				ca.mode = 0;
				launch_map_compact(st, ca, L.seg);
				uint32_t dn[6];
				HIPCHK(ctx, hipMemcpyAsync(dn, L.counts, sizeof(dn), hipMemcpyDeviceToHost, st));
				HIPCHK(ctx, hipStreamSynchronize(st)); // perm / keys are host vectors
				for (int c = 0; c < 4; c++)
					ndown[c] = dn[c];
			}
		}
		// results to the host
		Bytes host[MULLS_CL_COUNT];
		const float4 *dev[MULLS_CL_COUNT] = {cls[0], cls[1], cls[2], cls[3], L.down[0], L.down[1], L.down[2], L.down[3], L.vertex};
		const uint32_t cntk[MULLS_CL_COUNT] = {ncls[0], ncls[1], ncls[2], ncls[3], ndown[0], ndown[1], ndown[2], ndown[3], nvertex};
		for (int k = 0; k < MULLS_CL_COUNT; k++)
		{
			const bool thinned_later = P->fixed_num_downsampling && k >= MULLS_CL_PILLAR_DOWN && k <= MULLS_CL_ROOF_DOWN;
			const uint32_t want = thinned_later ? cntk[k] : (dev_out ? 0u : std::min(cntk[k], cap[k]));
			n_out[k] = cntk[k];
			if (dev_out)
				dev_out->dev[k] = dev[k], dev_out->n[k] = cntk[k], dev_out->on_host[k] = false;
			if (!want)
				continue;
			if (thinned_later)
			{
				host[k].resize((size_t)want * REC);
				HIPCHK(ctx, hipMemcpyAsync(host[k].data(), dev[k], host[k].size(), hipMemcpyDeviceToHost, st));
			}
			else
				HIPCHK(ctx, hipMemcpyAsync(out[k], dev[k], (size_t)want * REC, hipMemcpyDeviceToHost, st));
		}
		if (cloud_in_after)
			HIPCHK(ctx, hipMemcpyAsync(cloud_in_after, A.recs, (size_t)n * REC, hipMemcpyDeviceToHost, st));
		if (dev_out)
			dev_out->cloud_in_after = A.recs, dev_out->n_after = n;
		HIPCHK(ctx, hipStreamSynchronize(st));
		if (P->fixed_num_downsampling) // :2247-2257
		{
			random_downsample(host[MULLS_CL_PILLAR_DOWN], P->pillar_down_fixed_num, P->rng_seed, 31);
			const int sector_num = 4;
			xy_normal_balanced_downsample(host[MULLS_CL_FACADE_DOWN], (int)(P->facade_down_fixed_num / sector_num), sector_num, P->rng_seed, 32);
			xy_normal_balanced_downsample(host[MULLS_CL_BEAM_DOWN], (int)(P->beam_down_fixed_num / sector_num), sector_num, P->rng_seed, 36);
			random_downsample(host[MULLS_CL_ROOF_DOWN], P->roof_down_fixed_num, P->rng_seed, 40);
			for (int k = MULLS_CL_PILLAR_DOWN; k <= MULLS_CL_ROOF_DOWN; k++)
			{
				n_out[k] = (uint32_t)(host[k].size() / REC);
				if (dev_out)
				{
					dev_out->n[k] = n_out[k], dev_out->on_host[k] = true;
					dev_out->host[k].swap(host[k]);
					continue;
				}
				const size_t m = std::min<size_t>(n_out[k], cap[k]);
				if (m)
					std::memcpy(out[k], host[k].data(), m * REC);
			}
		}
		return MULLS_OK;
	}

	int mulls_classify_nground(mulls_ctx *ctx, const void *pts, uint32_t n_in, uint32_t stride, const mulls_classify_params *P, void *const out[MULLS_CL_COUNT],
							   const uint32_t cap[MULLS_CL_COUNT], uint32_t n_out[MULLS_CL_COUNT], void *cloud_in_after, uint32_t *n_cloud_in_after)
	try
	{
		return mulls_classify_impl(ctx, pts, false, n_in, stride, P, out, cap, n_out, cloud_in_after, n_cloud_in_after, nullptr);
	}
	catch (...)
	{
		return mulls::abi_caught(ctx); // nothing is thrown across the ABI
	}
}
