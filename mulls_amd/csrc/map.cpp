// map.cpp — device-resident local map (include/mulls_hip.h, "mulls_map_*"): MapManager::update_local_map
// (src/map_manager.cpp:18-140) and map-based dynamic-object removal (:149-268) on class clouds that stay in HBM between
// frames.  Kernels: map_kernels.hip (+ k_transform_aos of k_reduce.hip).  As everywhere in this library there is no CPU
// fallback: the only host-side arithmetic is the pose algebra, the kept-point counts and the seeded selection masks.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <cstdlib>

#include "batch.h"
#include "ctx.h"
#include "device_types.h"
#include "hostmath.h"
#include "launch.h"
#include "map_launch.h"

using mulls::Mat4;

struct mulls_map
{
	float4 *rec[MULLS_NC] = {}; // class clouds, 48-B records
	float4 *alt[MULLS_NC] = {}; // compaction target (ping-pong)
	size_t cap[MULLS_NC] = {}, cap_alt[MULLS_NC] = {};
	uint32_t n[MULLS_NC] = {};
	float4 *frame[MULLS_NC] = {}, *frame_alt[MULLS_NC] = {};
	size_t cap_frame[MULLS_NC] = {}, cap_frame_alt[MULLS_NC] = {};
	uint32_t frame_n[MULLS_NC] = {};
	double pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; // pose_lo, column-major
	uint32_t *counts = nullptr; // [6] device
	uint32_t *keys = nullptr;	// [12] device
	double *T12 = nullptr;		// [12] device
	uint32_t *best = nullptr;
	uint8_t *keep = nullptr;
	size_t cap_best = 0, cap_keep = 0;
	uint8_t *dmask = nullptr; // the thinning masks of a frame's update
	size_t cap_dmask = 0;
	uint32_t *seg = nullptr; // per-segment counters of the stable compactions
	size_t cap_seg = 0;
};

bool mulls_is_map_memory(const mulls_ctx *ctx, const void *p, size_t bytes)
{
	const char *q = (const char *)p;
	for (const mulls_map *m : ctx->maps)
		for (int c = 0; c < MULLS_NC; c++)
		{
			const char *lo = (const char *)m->rec[c];
			if (lo && q >= lo && q + bytes <= lo + m->cap[c] * 3 * sizeof(float4))
				return true;
		}
	for (const mulls_block *b : ctx->blocks)
		if (b->buf && q >= (const char *)b->buf && q + bytes <= (const char *)b->buf + b->cap)
			return true;
	return false;
}

namespace
{
const size_t REC = MULLS_POINT_BYTES;

void rows12_of(const Mat4 &M, double out[12])
{
	for (int r = 0; r < 3; r++)
		for (int c = 0; c < 4; c++)
			out[r * 4 + c] = M.at(r, c);
}

// capacity in records; keeps the contents when `keep_n` > 0
int reserve(mulls_ctx *ctx, float4 **p, size_t *cap, size_t need, size_t keep_n)
{
	if (*p && *cap >= need)
		return MULLS_OK;
	const size_t want = std::max<size_t>(need + need / 2, 1024);
	float4 *q = nullptr;
	HIPCHK(ctx, hipMalloc((void **)&q, want * REC));
	if (*p && keep_n)
		HIPCHK(ctx, hipMemcpyAsync(q, *p, keep_n * REC, hipMemcpyDeviceToDevice, ctx->stream));
	if (*p)
	{
		HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
		(void)hipFree(*p);
	}
	*p = q;
	*cap = want;
	return MULLS_OK;
}

// the six clouds of a frame / a map at once: device-resident sources (a feature block's, another map's) move in one launch and nobody waits; host sources are
// copied one by one and waited for once
int upload_clouds(mulls_ctx *ctx, const mulls_cloud clouds[MULLS_NC], float4 **dst, size_t *cap)
{
	std::vector<unsigned char> pack[MULLS_NC];
	SegCopier dev(ctx);
	bool any_host = false;
	for (int c = 0; c < MULLS_NC; c++)
	{
		const mulls_cloud &cl = clouds[c];
		if (cl.n && (!cl.pts || cl.stride < REC))
		{
			ctx->err = "cloud with points but null pointer or stride < 48";
			return MULLS_E_INVALID;
		}
		const int rc = reserve(ctx, &dst[c], &cap[c], cl.n, 0);
		if (rc != MULLS_OK)
			return rc;
		if (!cl.n)
			continue;
		if (cl.stride == REC && mulls_is_map_memory(ctx, cl.pts, (size_t)cl.n * REC))
		{
			dev.add_dev(dst[c], cl.pts, (size_t)cl.n * REC);
			continue;
		}
		const void *src = cl.pts;
		if (cl.stride != REC)
		{
			pack[c].resize((size_t)cl.n * REC);
			for (uint32_t i = 0; i < cl.n; i++)
				std::memcpy(pack[c].data() + (size_t)i * REC, (const unsigned char *)cl.pts + (size_t)i * cl.stride, REC);
			src = pack[c].data();
		}
		HIPCHK(ctx, hipMemcpyAsync(dst[c], src, (size_t)cl.n * REC, hipMemcpyDefault, ctx->stream));
		any_host = true;
	}
	if (dev.flush(ctx->stream) != MULLS_OK)
		return MULLS_E_HIP;
	if (any_host)
		HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); // `pack` / the caller's buffers may go away
	return MULLS_OK;
}

int compact(mulls_ctx *ctx, mulls_map *m, const MapCompactArgs &a)
{
	const size_t need = (size_t)map_compact_segments(a) + 8;
	if (m->cap_seg < need)
	{
		if (m->seg)
			(void)hipFree(m->seg);
		m->seg = nullptr;
		if (dmalloc(ctx, &m->seg, need * 2) != MULLS_OK)
			return MULLS_E_HIP;
		m->cap_seg = need * 2;
	}
	launch_map_compact(ctx->stream, a, m->seg);
	return MULLS_OK;
}

double key_to_double(uint32_t k, bool is_min)
{
	const bool none = is_min ? k == 0xffffffffu : k == 0u;
	if (none)
		return is_min ? 1.7976931348623157e308 : -1.7976931348623157e308;
	const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
	float f;
	std::memcpy(&f, &u, sizeof(f));
	return (double)f;
}
} // namespace

extern "C"
{
	void mulls_map_default_params(mulls_map_params *p)
	{
		std::memset(p, 0, sizeof(*p));
		p->local_map_radius = 80;
		p->max_num_pts = 20000;
		p->kept_vertex_num = 800;
		p->last_frame_reliable_radius = 60;
		std::strcpy(p->used_feature_type, "111110");
		p->dynamic_removal_center_radius = 30.0f;
		p->dynamic_dist_thre_min = 0.3f;
		p->dynamic_dist_thre_max = 3.0f;
		p->near_dist_thre = 0.03f;
		std::strcpy(p->tree_used, "000000");
	}

	int mulls_map_create(mulls_ctx *ctx, mulls_map **out)
	try
	{
		if (!ctx || !out)
			return MULLS_E_INVALID;
		*out = nullptr;
		HIPCHK(ctx, hipSetDevice(ctx->device));
		mulls_map *m = new mulls_map();
		if (dmalloc(ctx, &m->counts, 6) != MULLS_OK || dmalloc(ctx, &m->keys, 12) != MULLS_OK || dmalloc(ctx, &m->T12, 12) != MULLS_OK)
		{
			mulls_map_destroy(ctx, m);
			return MULLS_E_HIP;
		}
		ctx->maps.push_back(m);
		*out = m;
		return MULLS_OK;
	}
	catch (...)
	{
		return mulls::abi_caught(const_cast<mulls_ctx *>(ctx)); // nothing is thrown across the ABI
	}

	void mulls_map_destroy(mulls_ctx *ctx, mulls_map *m)
	{
		if (!m)
			return;
		if (ctx)
		{
			(void)hipSetDevice(ctx->device);
			(void)hipStreamSynchronize(ctx->stream);
			ctx->maps.erase(std::remove(ctx->maps.begin(), ctx->maps.end(), m), ctx->maps.end());
		}
		for (int c = 0; c < MULLS_NC; c++)
		{
			void *p[] = {m->rec[c], m->alt[c], m->frame[c], m->frame_alt[c]};
			for (void *q : p)
				if (q)
					(void)hipFree(q);
		}
		void *p[] = {m->counts, m->keys, m->T12, m->best, m->keep, m->seg, m->dmask};
		for (void *q : p)
			if (q)
				(void)hipFree(q);
		delete m;
	}

	int mulls_map_set(mulls_ctx *ctx, mulls_map *m, const mulls_cloud clouds[6], const double pose_lo[16])
	try
	{
		if (!ctx || !m || !clouds || !pose_lo)
			return MULLS_E_INVALID;
		HIPCHK(ctx, hipSetDevice(ctx->device));
		const int rc_up = upload_clouds(ctx, clouds, m->rec, m->cap);
		if (rc_up != MULLS_OK)
			return rc_up;
		for (int c = 0; c < MULLS_NC; c++)
		{
			m->n[c] = clouds[c].n;
			m->frame_n[c] = 0;
		}
		std::memcpy(m->pose, pose_lo, sizeof(m->pose));
		return MULLS_OK;
	}
	catch (...)
	{
		return mulls::abi_caught(const_cast<mulls_ctx *>(ctx)); // nothing is thrown across the ABI
	}

	int mulls_map_update(mulls_ctx *ctx, mulls_map *m, const mulls_cloud frame_down[6], const double frame_pose_lo[16], const mulls_map_params *P,
						 mulls_map_report *rep)
	try
	{
		if (!ctx || !m || !frame_down || !frame_pose_lo || !P || !rep)
			return MULLS_E_INVALID;
		if (std::strlen(P->used_feature_type) < 6 || std::strlen(P->tree_used) < 6)
		{
			ctx->err = "used_feature_type and tree_used need 6 characters";
			return MULLS_E_INVALID;
		}
		HIPCHK(ctx, hipSetDevice(ctx->device));
		const auto wall0 = std::chrono::steady_clock::now();
		hipStream_t st = ctx->stream;
		std::memset(rep, 0, sizeof(*rep));

		// 1. the frame's clouds on the device
		const int rc_up = upload_clouds(ctx, frame_down, m->frame, m->cap_frame);
		if (rc_up != MULLS_OK)
			return rc_up;
		for (int c = 0; c < MULLS_NC; c++)
			m->frame_n[c] = frame_down[c].n;
		// 2. tran_target_map = last_target.pose_lo^-1 * local_map.pose_lo (:28); the five *_down clouds go to the map frame (:32)
		Mat4 map_T, frame_T;
		std::memcpy(map_T.v, m->pose, sizeof(map_T.v));
		std::memcpy(frame_T.v, frame_pose_lo, sizeof(frame_T.v));
		const Mat4 tran_target_map = mulls::invert4(frame_T) * map_T;
		const Mat4 inv = mulls::invert4(tran_target_map);
		double t12[12];
		rows12_of(inv, t12);
		launch_transform_clouds(st, m->frame, m->frame_n, 5, t12); // (the transform travels with the launch)

		// 3. map-based dynamic-object removal on the frame's pillar, beam and facade clouds (:37-47, :149-268)
		float dmax = P->dynamic_dist_thre_max;
		{
			const double lo = P->dynamic_dist_thre_min + 0.1;
			dmax = (float)(((double)dmax > lo) ? (double)dmax : lo);
		}
		const int fpn0 = (int)(m->n[MULLS_GROUND] + m->n[MULLS_FACADE] + m->n[MULLS_ROOF] + m->n[MULLS_PILLAR] + m->n[MULLS_BEAM]);
		if (P->map_based_dynamic_removal_on && fpn0 > P->max_num_pts / 5 && P->tree_mode != 0)
		{
			rep->dynamic_removal_ran = 1;
			// the three classes side by side (nearest tree point, verdict), then ONE compaction and one read of the counts
			static const int order[3] = {MULLS_PILLAR, MULLS_BEAM, MULLS_FACADE};
			uint32_t first[3] = {0, 0, 0}, total = 0;
			bool run[3];
			for (int k = 0; k < 3; k++)
			{
				const int c = order[k];
				run[k] = !(P->used_feature_type[c] != '1' || P->tree_used[c] != '1' || m->frame_n[c] <= 10 || m->n[c] == 0);
				first[k] = total;
				if (run[k])
					total += (m->frame_n[c] + 3u) & ~3u;
			}
			if (total)
			{
				if (m->cap_best < total)
				{
					HIPCHK(ctx, hipStreamSynchronize(st));
					if (m->best)
						(void)hipFree(m->best);
					if (m->keep)
						(void)hipFree(m->keep);
					m->best = nullptr, m->keep = nullptr;
					if (dmalloc(ctx, &m->best, (size_t)total * 2) != MULLS_OK || dmalloc(ctx, &m->keep, (size_t)total * 2) != MULLS_OK)
						return MULLS_E_HIP;
					m->cap_best = (size_t)total * 2;
				}
				HIPCHK(ctx, hipMemsetD32Async((hipDeviceptr_t)m->best, 0x7f800000, total, st));
				MapCompactArgs a;
				std::memset(&a, 0, sizeof(a));
				for (int k = 0; k < 3; k++)
				{
					if (!run[k])
						continue;
					const int c = order[k];
					const uint32_t nf = m->frame_n[c];
					launch_map_nn(st, m->frame[c], nf, m->rec[c], m->n[c], P->tree_mode == 2, P->tree_box, m->best + first[k]);
					launch_map_keep(st, m->frame[c], nf, m->best + first[k], P->dynamic_removal_center_radius, P->dynamic_dist_thre_min, dmax, P->near_dist_thre,
									m->keep + first[k]);
					const int rc = reserve(ctx, &m->frame_alt[c], &m->cap_frame_alt[c], nf, 0);
					if (rc != MULLS_OK)
						return rc;
					a.cloud[k].in = m->frame[c];
					a.cloud[k].out = m->frame_alt[c];
					a.cloud[k].mask = m->keep + first[k];
					a.cloud[k].n = nf;
				}
				a.out_n = m->counts;
				a.mode = 0;
				if (compact(ctx, m, a) != MULLS_OK) // the other slots are empty clouds
					return MULLS_E_HIP;
				uint32_t cnt[6];
				HIPCHK(ctx, hipMemcpyAsync(cnt, m->counts, sizeof(cnt), hipMemcpyDeviceToHost, st));
				HIPCHK(ctx, hipStreamSynchronize(st));
				for (int k = 0; k < 3; k++)
					if (run[k])
					{
						const int c = order[k];
						std::swap(m->frame[c], m->frame_alt[c]);
						std::swap(m->cap_frame[c], m->cap_frame_alt[c]);
						m->frame_n[c] = cnt[k];
					}
			}
		}
		for (int c = 0; c < MULLS_NC; c++)
			rep->frame_n[c] = m->frame_n[c];

		// 4. append_feature(last_target, true, used) (:54): the vertex cloud always
		{
			SegCopier app(ctx);
			for (int c = 0; c < MULLS_NC; c++)
				if ((c == MULLS_VERTEX || P->used_feature_type[c] == '1') && m->frame_n[c])
				{
					const int rc = reserve(ctx, &m->rec[c], &m->cap[c], (size_t)m->n[c] + m->frame_n[c], m->n[c]);
					if (rc != MULLS_OK)
						return rc;
					app.add_dev(m->rec[c] + (size_t)m->n[c] * 3, m->frame[c], (size_t)m->frame_n[c] * REC);
					m->n[c] += m->frame_n[c];
				}
			if (app.flush(st) != MULLS_OK)
				return MULLS_E_HIP;
		}
		// 5. the map moves to the frame's coordinates (:57-59)
		rows12_of(tran_target_map, t12);
		launch_transform_clouds(st, m->rec, m->n, MULLS_NC, t12);
		std::memcpy(m->pose, frame_pose_lo, sizeof(m->pose));
		// 6. dist_filter(cloud, local_map_radius) on all six (:62-67)
		MapCompactArgs a;
		std::memset(&a, 0, sizeof(a));
		for (int c = 0; c < MULLS_NC; c++)
		{
			const int rc = reserve(ctx, &m->alt[c], &m->cap_alt[c], m->n[c], 0);
			if (rc != MULLS_OK)
				return rc;
			a.cloud[c].in = m->rec[c];
			a.cloud[c].out = m->alt[c];
			a.cloud[c].n = m->n[c];
		}
		a.out_n = m->counts;
		a.mode = 1;
		a.radius = (double)P->local_map_radius;
		if (compact(ctx, m, a) != MULLS_OK)
			return MULLS_E_HIP;
		uint32_t cnt[6];
		HIPCHK(ctx, hipMemcpyAsync(cnt, m->counts, sizeof(cnt), hipMemcpyDeviceToHost, st));
		HIPCHK(ctx, hipStreamSynchronize(st));
		for (int c = 0; c < MULLS_NC; c++)
		{
			std::swap(m->rec[c], m->alt[c]);
			std::swap(m->cap[c], m->cap_alt[c]);
			m->n[c] = cnt[c];
		}
		// 7. random_downsample_pcl to the per-class share of max_num_pts (:70-86), seeded selection sampling
		const int cur = (int)(m->n[MULLS_GROUND] + m->n[MULLS_FACADE] + m->n[MULLS_ROOF] + m->n[MULLS_PILLAR] + m->n[MULLS_BEAM]);
		int kept[MULLS_NC];
		for (int c = 0; c < 5; c++)
			kept[c] = cur > 0 ? (int)(1.0 * P->max_num_pts / cur * m->n[c] + 1) : 1;
		kept[MULLS_VERTEX] = P->kept_vertex_num;
		bool any_thin = false;
		size_t mask_total = 0;
		for (int c = 0; c < MULLS_NC; c++)
			if ((long)m->n[c] > (long)kept[c])
			{
				any_thin = true;
				mask_total += m->n[c];
			}
		if (any_thin)
		{
			std::vector<uint8_t> mask(mask_total);
			if (m->cap_dmask < mask_total) // (grow-only: no allocator call per frame)
			{
				HIPCHK(ctx, hipStreamSynchronize(st));
				if (m->dmask)
					(void)hipFree(m->dmask);
				m->dmask = nullptr, m->cap_dmask = 0;
				if (dmalloc(ctx, &m->dmask, mask_total * 2) != MULLS_OK)
					return MULLS_E_HIP;
				m->cap_dmask = mask_total * 2;
			}
			uint8_t *dmask = m->dmask;
			std::memset(&a, 0, sizeof(a));
			size_t off = 0;
			uint32_t newn[MULLS_NC];
			for (int c = 0; c < MULLS_NC; c++)
			{
				newn[c] = m->n[c];
				if ((long)m->n[c] <= (long)kept[c])
					continue; // slot stays an empty cloud: nothing to do
				newn[c] = thin_mask(mask.data() + off, m->n[c], kept[c], P->rng_seed, 20 + c);
				a.cloud[c].in = m->rec[c];
				a.cloud[c].out = m->alt[c]; // capacity >= the pre-filter size
				a.cloud[c].mask = dmask + off;
				a.cloud[c].n = m->n[c];
				off += m->n[c];
			}
			a.out_n = m->counts;
			a.mode = 0;
			hipError_t e = hipMemcpyAsync(dmask, mask.data(), mask_total, hipMemcpyHostToDevice, st);
			if (e == hipSuccess)
			{
				if (compact(ctx, m, a) != MULLS_OK)
					e = hipErrorOutOfMemory;
				else
					e = hipStreamSynchronize(st);
			}
			if (e != hipSuccess)
			{
				ctx->err = std::string("map thinning: ") + hipGetErrorString(e);
				return MULLS_E_HIP;
			}
			for (int c = 0; c < MULLS_NC; c++)
				if ((long)m->n[c] > (long)kept[c])
				{
					std::swap(m->rec[c], m->alt[c]);
					std::swap(m->cap[c], m->cap_alt[c]);
					m->n[c] = newn[c];
				}
		}
		// 8. bounds of the merged cloud, local and posed (:88-94)
		uint32_t keys[12];
		for (int k = 0; k < 12; k++)
			keys[k] = (k % 6) < 3 ? 0xffffffffu : 0u;
		HIPCHK(ctx, hipMemcpyAsync(m->keys, keys, sizeof(keys), hipMemcpyHostToDevice, st));
		MapBoxArgs b;
		std::memset(&b, 0, sizeof(b));
		for (int c = 0; c < MULLS_NC; c++)
		{
			b.recs[c] = m->rec[c];
			b.n[c] = m->n[c];
		}
		rows12_of(frame_T, b.pose);
		b.keys = m->keys;
		launch_map_bbox(st, b);
		HIPCHK(ctx, hipMemcpyAsync(keys, m->keys, sizeof(keys), hipMemcpyDeviceToHost, st));
		HIPCHK(ctx, hipStreamSynchronize(st));
		for (int k = 0; k < 6; k++)
		{
			rep->local_bound[k] = key_to_double(keys[k], k < 3);
			rep->bound[k] = key_to_double(keys[6 + k], k < 3);
		}
		// 9. recalculate_feature_on (:98-118): principal directions of the pillar and beam clouds from their own neighbourhoods
		//    (pca_radius 1.8, pca_max_k 20, pca_min_k 6, min_linearity 0.65), pillars kept above sin 0.80, beams below sin 0.25
		if (P->recalculate_feature_on)
		{
			static const int cls[2] = {MULLS_PILLAR, MULLS_BEAM};
			static const float sin_low[2] = {0.0f, 0.25f}, sin_high[2] = {0.80f, 1.0f};
			for (int k = 0; k < 2; k++)
			{
				const int c = cls[k];
				const uint32_t nc = m->n[c];
				if (P->used_feature_type[c] != '1' || nc == 0)
					continue;
				if (m->cap_best < nc)
				{
					if (m->best)
						(void)hipFree(m->best);
					if (m->keep)
						(void)hipFree(m->keep);
					m->best = nullptr, m->keep = nullptr;
					m->cap_best = 0;
					if (dmalloc(ctx, &m->best, (size_t)nc * 2) != MULLS_OK || dmalloc(ctx, &m->keep, (size_t)nc * 2) != MULLS_OK)
						return MULLS_E_HIP;
					m->cap_best = (size_t)nc * 2;
				}
				MapPcaArgs pa;
				pa.recs = m->rec[c];
				pa.n = nc;
				pa.radius = 1.8f;
				pa.max_k = 20;
				pa.min_k = 6;
				pa.sin_low = sin_low[k];
				pa.sin_high = sin_high[k];
				pa.min_linearity = 0.65f;
				pa.keep = m->keep;
				launch_map_pca(st, pa);
				const int rc = reserve(ctx, &m->alt[c], &m->cap_alt[c], nc, 0);
				if (rc != MULLS_OK)
					return rc;
				std::memset(&a, 0, sizeof(a));
				a.cloud[0].in = m->rec[c];
				a.cloud[0].out = m->alt[c];
				a.cloud[0].mask = m->keep;
				a.cloud[0].n = nc;
				a.out_n = m->counts;
				a.mode = 0;
				if (compact(ctx, m, a) != MULLS_OK)
					return MULLS_E_HIP;
				HIPCHK(ctx, hipMemcpyAsync(cnt, m->counts, sizeof(cnt), hipMemcpyDeviceToHost, st));
				HIPCHK(ctx, hipStreamSynchronize(st));
				std::swap(m->rec[c], m->alt[c]);
				std::swap(m->cap[c], m->cap_alt[c]);
				m->n[c] = cnt[0];
			}
		}
		for (int c = 0; c < MULLS_NC; c++)
			rep->n[c] = m->n[c];
		rep->feature_point_num = (int)(m->n[MULLS_GROUND] + m->n[MULLS_FACADE] + m->n[MULLS_ROOF] + m->n[MULLS_PILLAR] + m->n[MULLS_BEAM]);
		rep->ms_total = (float)(std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count() * 1e3);
		return MULLS_OK;
	}
	catch (...)
	{
		return mulls::abi_caught(const_cast<mulls_ctx *>(ctx)); // nothing is thrown across the ABI
	}

	int mulls_map_cloud(mulls_ctx *ctx, const mulls_map *m, int cls, mulls_cloud *out)
	try
	{
		if (!ctx || !m || !out || cls < 0 || cls >= MULLS_NC)
			return MULLS_E_INVALID;
		out->pts = m->rec[cls];
		out->n = m->n[cls];
		out->stride = (uint32_t)REC;
		return MULLS_OK;
	}
	catch (...)
	{
		return mulls::abi_caught(const_cast<mulls_ctx *>(ctx)); // nothing is thrown across the ABI
	}

	int mulls_map_pose(mulls_ctx *ctx, const mulls_map *m, double pose_lo[16])
	try
	{
		if (!ctx || !m || !pose_lo)
			return MULLS_E_INVALID;
		std::memcpy(pose_lo, m->pose, sizeof(m->pose));
		return MULLS_OK;
	}
	catch (...)
	{
		return mulls::abi_caught(const_cast<mulls_ctx *>(ctx)); // nothing is thrown across the ABI
	}

	static int download(mulls_ctx *ctx, const float4 *src, uint32_t have, void *pts, uint32_t cap, uint32_t *n)
	{
		if (n)
			*n = have;
		const uint32_t k = std::min(have, cap);
		if (k && !pts)
			return MULLS_E_INVALID;
		if (k)
		{
			HIPCHK(ctx, hipSetDevice(ctx->device));
			HIPCHK(ctx, hipMemcpyAsync(pts, src, (size_t)k * REC, hipMemcpyDeviceToHost, ctx->stream));
			HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
		}
		return MULLS_OK;
	}
	int mulls_map_download(mulls_ctx *ctx, const mulls_map *m, int cls, void *pts, uint32_t cap, uint32_t *n)
	try
	{
		if (!ctx || !m || cls < 0 || cls >= MULLS_NC)
			return MULLS_E_INVALID;
		return download(ctx, m->rec[cls], m->n[cls], pts, cap, n);
	}
	catch (...)
	{
		return mulls::abi_caught(const_cast<mulls_ctx *>(ctx)); // nothing is thrown across the ABI
	}
	int mulls_map_frame_download(mulls_ctx *ctx, const mulls_map *m, int cls, void *pts, uint32_t cap, uint32_t *n)
	try
	{
		if (!ctx || !m || cls < 0 || cls >= MULLS_NC)
			return MULLS_E_INVALID;
		return download(ctx, m->frame[cls], m->frame_n[cls], pts, cap, n);
	}
	catch (...)
	{
		return mulls::abi_caught(const_cast<mulls_ctx *>(ctx)); // nothing is thrown across the ABI
	}
}
