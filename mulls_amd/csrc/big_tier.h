// big_tier.h — device code of the global-memory tier of the correspondence search with CERTIFIED correspondences: target class clouds of any
// size (occupancy-bitmap grid in global memory: k_bm_*, k_grid.hip), source class clouds of any size.  The counterpart of lds_tier.h for
// the clouds that do not fit the LDS tier — a dense scan pair, a scan against a large local map.
//
// One kernel, k_cert_big (k_search.hip), two job shapes:
//   class-level   one 512-lane workgroup holds a whole source class cloud of at most 3 x 512 points (a down-sampled scan against a large map):
//                 rigid step + certificates, search of the leftovers, duplicate rule and rejection chain in ONE pass, the per-point state in
//                 registers — what cert_class_flat does in the LDS tier.  The duplicate table is the batch's global `winner` table (the workgroup sees
//                 every query of the cloud, so a workgroup barrier orders it).  No k_filter launch.
//   chunk-level   512 consecutive source points of a large class cloud per job (optionally shared by `split` workgroups): rigid step + certificates +
//                 search; the duplicate rule needs every chunk of the cloud, so it and the rejection chain stay in k_filter (one launch later).
// The certificate is lds_tier.h's (triangle inequality on the bound the last search left behind); what differs is the index width (32-bit
// hints: no cost class), the grid (bitmap rows probed by 16-lane sub-groups, device_util.h) and the sweep policy: a hinted query sweeps
// the cube of (distance to the hinted target + slack) right away — with cells of 0.25-0.7 m that is a handful of rows, while the LDS
// tier's 1.3 m cells make "at most three cells per axis" the cheaper first probe.  Same outputs as every other tier.
#pragma once
#include "lds_tier.h"

namespace
{
// (best key, second-best distance) over the 16 lanes of a sub-group (row16_min2 for MULLS_GRID_GROUP = 16 lanes: one DPP row)
__device__ __forceinline__ void grp16_min2(nnkey &bk, float &sec)
{
	nnkey g = bk;
	dpp_min_step<0xB1>(g);	// quad_perm [1,0,3,2]
	dpp_min_step<0x4E>(g);	// quad_perm [2,3,0,1]
	dpp_min_step<0x141>(g); // row_half_mirror
	dpp_min_step<0x140>(g); // row_mirror
	float c = (bk == g) ? sec : key_dist(bk);
	dpp_fmin_step<0xB1>(c);
	dpp_fmin_step<0x4E>(c);
	dpp_fmin_step<0x141>(c);
	dpp_fmin_step<0x140>(c);
	sec = c;
	bk = g;
}
// one candidate of a lane (take_pair's update, lds_tier.h)
__device__ __forceinline__ void take_one(const float4 c, float px, float py, float pz, nnkey &bk, float &sec)
{
	const float dx = px - c.x, dy = py - c.y, dz = pz - c.z;
	const float dist = (dx * dx + dy * dy) + dz * dz; // L2_Simple<float>, no FMA
	sec = __builtin_amdgcn_fmed3f(sec, dist, key_dist(bk));
	const nnkey k = nn_key(dist, __float_as_uint(c.w));
	bk = k < bk ? k : bk;
}
// grid_scan_box (device_util.h) with the (best key, second-smallest distance) state of the certificates: every target in the cells that meet the
// cube [p - R, p + R] is evaluated; 16 rows per two memory latencies, four candidate loads in flight per lane
__device__ __forceinline__ void bm_scan_box2(const GridDesc &g, const BmGrid &B, const float4 *__restrict__ ts, float px, float py, float pz, float R,
											  uint32_t sub, nnkey &bk, float &sec)
{
	const float Rm = R * 1.0001f + 1e-4f;
	const int x0 = grid_cell(px - Rm, g.ox, g.inv_h, g.nx), x1 = grid_cell(px + Rm, g.ox, g.inv_h, g.nx);
	const int y0 = grid_cell(py - Rm, g.oy, g.inv_h, g.ny), y1 = grid_cell(py + Rm, g.oy, g.inv_h, g.ny);
	const int z0 = grid_cell(pz - Rm, g.oz, g.inv_h, g.nz), z1 = grid_cell(pz + Rm, g.oz, g.inv_h, g.nz);
	const int nyc = y1 - y0 + 1, nrows = nyc * (z1 - z0 + 1);
	const uint32_t gshift = (threadIdx.x & 63u) & ~(MULLS_GRID_GROUP - 1u); // first lane of this sub-group inside its wave
	for (int base = 0; base < nrows; base += (int)MULLS_GRID_GROUP)
	{
		const int j = base + (int)sub;
		uint32_t lo = 0, hi = 0;
		if (j < nrows)
		{
			const uint32_t rowword = ((uint32_t)(z0 + j / nyc) * g.ny + (uint32_t)(y0 + j % nyc)) * g.wpr;
			bm_row_range(B, rowword, (uint32_t)x0, (uint32_t)x1, lo, hi);
		}
		const uint32_t nonempty = (uint32_t)(__ballot(hi > lo) >> gshift) & 0xffffu; // per sub-group: which of its 16 rows hold points
		if (!nonempty)
			continue;
#pragma unroll
		for (int quarter = 0; quarter < 4; quarter++) // four rows at a time: eight candidate records in flight per lane cost 13 more registers than the kernel has at
		{											  // four waves per SIMD (-Rpass-analysis: 128 + 15 spilled against 115)
			if (!((nonempty >> (4 * quarter)) & 0xfu))
				continue;
			float4 q[4];
			bool ok[4];
#pragma unroll
			for (int jj = 0; jj < 4; jj++)
			{
				const uint32_t t = __shfl(lo, 4 * quarter + jj, MULLS_GRID_GROUP) + sub;
				ok[jj] = t < __shfl(hi, 4 * quarter + jj, MULLS_GRID_GROUP);
				if (ok[jj])
					q[jj] = ts[t];
			}
#pragma unroll
			for (int jj = 0; jj < 4; jj++)
				if (ok[jj])
					take_one(q[jj], px, py, pz, bk, sec);
		}
		// rows holding more than 16 candidates: four loads in flight per lane and trip
		for (uint32_t rem = nonempty; rem; rem &= rem - 1u)
		{
			const int jj = __ffs((int)rem) - 1;
			const uint32_t lo_j = __shfl(lo, jj, MULLS_GRID_GROUP), hi_j = __shfl(hi, jj, MULLS_GRID_GROUP);
			for (uint32_t t = lo_j + sub + MULLS_GRID_GROUP; t < hi_j; t += 4 * MULLS_GRID_GROUP)
			{
				float4 c[4];
				bool v[4];
#pragma unroll
				for (int w = 0; w < 4; w++)
				{
					v[w] = t + w * MULLS_GRID_GROUP < hi_j;
					if (v[w])
						c[w] = ts[t + w * MULLS_GRID_GROUP];
				}
#pragma unroll
				for (int w = 0; w < 4; w++)
					if (v[w])
						take_one(c[w], px, py, pz, bk, sec);
			}
		}
	}
}

// The exact nearest target of one query (q.xyz; q.w = radius of the hinted sweep, +inf = no hint) by the 16 lanes of a sub-group, against the
// bitmap grid.  Out as search_query (lds_tier.h): bk = (squared distance, original index) of the nearest target or NNKEY_NONE; every target
// other than bk's is at least min(sqrt(sec), Rfin) away (sec = 0: nothing is claimed).
//   hinted    the cube of q.w (>= the distance to the hinted target: it contains that target, so the sweep is complete);
//   unhinted  the own cell, the cube of min(first-probe radius, best so far), then — while nothing lies inside the probed radius — one last cube
//             of the distance found, or cubes of twice the radius up to the rejection radius r (k_nn_grid's policy, rounds 1-3).
//             probe_own = false skips the own-cell probe: on a grid with one or two points per occupied cell it rarely bounds the search inside the cell and
//             costs three dependent round trips of the query's six (a 236 k-point pair's first iteration: 2.2 ms of a 15 ms step, profiles/r04_large_steps.txt)
// co: the other lanes' nearest targets of the sweep that produced (bk, sec), for the k-candidate certificates (lane_bests, lds_tier.h).
__device__ __forceinline__ void search_query_bm(const GridDesc &g, const BmGrid &B, const float4 *__restrict__ ts, const float4 q, float r, float m, uint32_t sub,
												 bool probe_own, nnkey &bk, float &sec, float &Rfin, CandOut &co)
{
	bk = NNKEY_NONE;
	sec = __builtin_inff();
	Rfin = 0.0f;
	nnkey lane_k = NNKEY_NONE; // this lane's own share of the standing sweep
	float lane_s = __builtin_inff();
	// one sweep of the cube of radius R: its cells are a superset of every earlier sweep's cells unless the radius was clipped to r (see search_query)
	auto sweep = [&](float R) {
		nnkey lk = NNKEY_NONE;
		float ls = __builtin_inff();
		bm_scan_box2(g, B, ts, q.x, q.y, q.z, R, sub, lk, ls);
		const nnkey own_k = lk;
		const float own_s = ls;
		grp16_min2(lk, ls);
		// (selects: two branches ending in stores to different variables are merged into one store through a selected address, which puts both on the stack)
		const bool adopt = !(bk < lk);
		bk = adopt ? lk : bk;
		sec = adopt ? ls : 0.0f;
		lane_k = adopt ? own_k : lane_k;
		lane_s = adopt ? own_s : lane_s;
		Rfin = R;
	};
	float Rc;
	if (q.w < __builtin_inff())
	{
		Rc = fminf(r, q.w);
		sweep(Rc);
	}
	else if (!probe_own)
	{
		Rc = m;
		sweep(Rc);
	}
	else
	{
		// probe 0: the query's own cell
		const int ocx = grid_cell(q.x, g.ox, g.inv_h, g.nx), ocy = grid_cell(q.y, g.oy, g.inv_h, g.ny), ocz = grid_cell(q.z, g.oz, g.inv_h, g.nz);
		{
			uint32_t lo, hi;
			bm_row_range(B, ((uint32_t)ocz * g.ny + (uint32_t)ocy) * g.wpr, (uint32_t)ocx, (uint32_t)ocx, lo, hi);
			for (uint32_t t = lo + sub; t < hi; t += 4 * MULLS_GRID_GROUP)
			{
				float4 c[4];
				bool v[4];
#pragma unroll
				for (int w = 0; w < 4; w++)
				{
					v[w] = t + w * MULLS_GRID_GROUP < hi;
					if (v[w])
						c[w] = ts[t + w * MULLS_GRID_GROUP];
				}
#pragma unroll
				for (int w = 0; w < 4; w++)
					if (v[w])
						take_one(c[w], q.x, q.y, q.z, bk, sec);
			}
		}
		lane_k = bk, lane_s = sec;
		grp16_min2(bk, sec);
		// probe 1: the cells within min(first-probe radius, current best distance); when that cube is the own cell the probe above was complete
		const float R1 = key_found(bk) ? fminf(m, sqrtf(key_dist(bk))) : m;
		const float Rm = R1 * 1.0001f + 1e-4f;
		const bool own_only = grid_cell(q.x - Rm, g.ox, g.inv_h, g.nx) == ocx && grid_cell(q.x + Rm, g.ox, g.inv_h, g.nx) == ocx &&
							  grid_cell(q.y - Rm, g.oy, g.inv_h, g.ny) == ocy && grid_cell(q.y + Rm, g.oy, g.inv_h, g.ny) == ocy &&
							  grid_cell(q.z - Rm, g.oz, g.inv_h, g.nz) == ocz && grid_cell(q.z + Rm, g.oz, g.inv_h, g.nz) == ocz;
		if (own_only)
			Rfin = R1;
		else
			sweep(R1);
		Rc = m;
	}
	// nothing inside the probed radius yet: one last probe at the distance found, else double the radius (up to r)
	while (!(key_found(bk) && key_dist(bk) <= Rc * Rc) && Rc < r)
	{
		const bool last = key_found(bk);
		Rc = last ? fminf(r, sqrtf(key_dist(bk))) : fminf(r, 2.0f * Rc);
		sweep(Rc);
		if (last)
			break;
	}
	co = lane_bests<(int)MULLS_GRID_GROUP, false>(lane_k, lane_s);
	if (sec == 0.0f)
		co.b2 = 0.0f; // the standing result lies outside the last (clipped) sweep: nothing is claimed about the other targets
}

#define MULLS_US16_KCERT 0x8000u // BigLds::us: a hinted point — it gets the k-candidate certificate's look (slots stay below 1536)
// LDS of k_cert_big: the leftover queries of the workgroup's points (every point can be one), reduction scratch
#define MULLS_BIG_HASH 4096u // class-level jobs: slots of the on-chip duplicate table (at most 1536 targets are claimed: load <= 0.375)
template <int SLOTS>
struct BigLds
{
	float4 uq[SLOTS];
	float um[SLOTS];	// how far this iteration's step moved the listed point (the k-candidate certificate's look needs it)
	uint16_t us[SLOTS]; // the query's slot among the workgroup's points (| MULLS_US16_KCERT: a hinted point, gets the look)
	uint32_t ucount, red[3 * 16];
	// class-level jobs (the workgroup holds every query of its class cloud): what a search found for the point of slot k — correspondence and squared distance, read by
	// the owner lane's tail from here instead of from nn_idx / nn_d2 in memory — and the duplicate table as an open-addressing hash of the claimed targets (the
	// batch's winner table took a device-scope fence and a gather past the L2 per point: the chain of a workgroup whose points all certify was mostly that)
	uint2 res[SLOTS];
	uint32_t hkey[MULLS_BIG_HASH], hval[MULLS_BIG_HASH];
};
// lowest source slot that claims target t (duplicate rule: the first source in the serial walk keeps a target, cregistration.hpp:1762-1789)
template <int SLOTS>
__device__ __forceinline__ void big_hash_min(BigLds<SLOTS> &CL, uint32_t t, uint32_t s)
{
	uint32_t h = (t * 2654435761u) >> 20;
	for (;;)
	{
		const uint32_t k = atomicCAS(&CL.hkey[h], 0xffffffffu, t);
		if (k == 0xffffffffu || k == t)
		{
			atomicMin(&CL.hval[h], s);
			return;
		}
		h = (h + 1u) & (MULLS_BIG_HASH - 1u);
	}
}
// ... of a target this workgroup has claimed (after the barrier that completes the table)
template <int SLOTS>
__device__ __forceinline__ uint32_t big_hash_get(const BigLds<SLOTS> &CL, uint32_t t)
{
	uint32_t h = (t * 2654435761u) >> 20;
	for (uint32_t n = 0; n < MULLS_BIG_HASH; n++)
	{
		if (CL.hkey[h] == t)
			return CL.hval[h];
		h = (h + 1u) & (MULLS_BIG_HASH - 1u);
	}
	return 0xffffffffu;
}

// One job of the big tier: the source points [q0, q1) of class cloud d by one workgroup of BLK lanes, at most TRIPS * BLK of them.
// class_level: [q0, q1) is the whole cloud — the workgroup finishes the cloud's iteration itself (duplicate rule, rejection chain, counters);
// otherwise it leaves nn_idx / nn_d2 / the winner entries for k_filter and adds its matches to the class counter.  Every lane of the workgroup calls it.
template <int BLK, int TRIPS>
__device__ __forceinline__ void cert_big(BigLds<BLK * TRIPS> &CL, const RunParams &rp, const PairState &ps, uint32_t cls, uint32_t q0, uint32_t q1, bool class_level,
										  CloudDesc &d, const GridDesc &g, const BmGrid &B, const float4 *__restrict__ ts, float4 *__restrict__ spos,
										  float4 *__restrict__ snrm, uint8_t *flag, int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2,
										  unsigned long long *__restrict__ winner, const float4 *__restrict__ tnrm, int32_t *__restrict__ match, float *__restrict__ wd,
										  const float4 *__restrict__ tpos, int2 *__restrict__ hint2, float4 *__restrict__ mq)
{
	static_assert(TRIPS == 3, "two trips' loads in flight, three trips at most");
	const uint32_t tgt_n = d.tgt_n;
	const bool called = class_called(rp, d, (int)cls);
	const ClassCtx C = class_ctx(rp, ps, g, (int)cls, d.alive_cur, called);
	const bool have_prev = ps.iter > 0; // hint records of this run exist from its second iteration on
	const bool normal_check = cls != 5u; // vertex correspondences skip the direction check (cregistration.hpp:1292)
	const bool kc = rp.kcert != 0u && have_prev && called && C.cand != nullptr; // (uniform)
	const uint32_t nq = q1 > q0 ? q1 - q0 : 0u;
	const uint32_t ntrips = (nq + BLK - 1u) / BLK; // uniform
	static_assert((1u << 12) == MULLS_BIG_HASH && BLK * TRIPS <= 1536, "big_hash_min's 12-bit hash; at most 1536 claims");
	const bool hashed = class_level && C.gate; // (uniform) the duplicate rule is in force and every query of the cloud is this workgroup's: the table stays on chip
	auto claim = [&](uint32_t t, uint32_t src) {
		if (hashed)
			big_hash_min(CL, t, src);
		else
			atomicMin(&winner[d.tgt_off + t], C.key_hi | (unsigned long long)src);
	};

	struct Rec
	{
		uint32_t fl;
		int32_t pm;
		float3 p, n, q, t; // position, direction, standing target position and direction (x y z: the fourth words stay where they are)
		int2 h;
	};
	const uint32_t last = d.src_off + (q1 ? q1 - 1u : 0u);
	auto load = [&](int k) {
		Rec r;
		const uint32_t gi = min(d.src_off + q0 + threadIdx.x + (uint32_t)k * BLK, last);
		r.fl = flag[gi];
		r.p = *reinterpret_cast<const float3 *>(spos + gi), r.n = *reinterpret_cast<const float3 *>(snrm + gi);
		r.h = hint2[gi];
		r.pm = match[gi];
		r.q = *reinterpret_cast<const float3 *>(mq + 2u * gi);
		r.t = make_float3(0.0f, 0.0f, 0.0f);
		if (class_level) // (uniform) the direction check of the rejection chain runs here only for class-level jobs
			r.t = *reinterpret_cast<const float3 *>(mq + 2u * gi + 1u);
		return r;
	};
	uint32_t ST[TRIPS];
	int32_t M[TRIPS];
	float D0[TRIPS];
	uint32_t matched_cnt = 0;
	// rigid step + certificate of one trip's point (cert_class_flat's arithmetic)
	auto cert = [&](int k, const Rec &r) {
		const uint32_t s = q0 + threadIdx.x + (uint32_t)k * BLK, gi = d.src_off + s;
		ST[k] = 0u, M[k] = -1, D0[k] = 0.0f;
		if (s >= q1 || !(r.fl & MULLS_F_ALIVE))
			return;
		ST[k] = r.fl & (MULLS_FS_ALIVE | MULLS_FS_VALID);
		const float3 p = r.p, n = r.n;
		uint32_t hj = 0xffffffffu;
		float lb = 0.0f;
		int32_t pm = -1;
		if (have_prev)
		{
			lb = __int_as_float(r.h.y);
			if (called)
			{
				hj = (uint32_t)r.h.x;
				pm = r.pm;
			}
		}
		float4 tj = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		if (hj < tgt_n)
			tj = (int32_t)hj == pm ? make_float4(r.q.x, r.q.y, r.q.z, 0.0f) : tpos[d.tgt_off + hj];
		// fused rigid step (cregistration.hpp:1690-1695): double math, float store, in place
		const double *T = ps.T;
		const double x = p.x, y = p.y, z = p.z, nx = n.x, ny = n.y, nz = n.z;
		float4 out;
		out.x = (float)(T[0] * x + T[1] * y + T[2] * z + T[3]);
		out.y = (float)(T[4] * x + T[5] * y + T[6] * z + T[7]);
		out.z = (float)(T[8] * x + T[9] * y + T[10] * z + T[11]);
		const float onx = (float)(T[0] * nx + T[1] * ny + T[2] * nz);
		const float ony = (float)(T[4] * nx + T[5] * ny + T[6] * nz);
		const float onz = (float)(T[8] * nx + T[9] * ny + T[10] * nz);
		*reinterpret_cast<float3 *>(spos + gi) = make_float3(out.x, out.y, out.z);
		*reinterpret_cast<float3 *>(snrm + gi) = make_float3(onx, ony, onz);
		if (class_level)
		{
			// the direction check of the rejection chain (:1818-1826) against the standing correspondence's target direction
			const double dot = (double)onx * (double)r.t.x + (double)ony * (double)r.t.y + (double)onz * (double)r.t.z;
			const float c = (float)fabs(dot);
			if (!((double)c < rp.cos_bearing))
				ST[k] |= MULLS_FS_DIR_OK;
		}
		const float mx = out.x - p.x, my = out.y - p.y, mz = out.z - p.z;
		const float moved = sqrtf((mx * mx + my * my) + mz * mz);
		const float lb_next = lb - moved * 1.00001f;
		if (!called)
		{
			// the points still move: keep the bounds of a class that sits this iteration out valid (none exist at iteration 0)
			hint2[gi].y = __float_as_int(have_prev ? lb_next : 0.0f);
			return;
		}
		out.w = __builtin_inff(); // sweep radius of a search: +inf = no hint
		bool certified = false;
		if (hj < tgt_n)
		{
			const float dx = out.x - tj.x, dy = out.y - tj.y, dz = out.z - tj.z;
			const float d0 = (dx * dx + dy * dy) + dz * dz; // the very expression a search evaluates for this candidate
			if (d0 >= 0.0f)
			{
				const float dh = sqrtf(d0);
				certified = rp.cert != 0u && (dh * 1.00001f + moved * 1.00001f < lb * 0.99999f); // NaN anywhere fails the test
				out.w = dh + fminf(fmaxf(rp.cert_slack_rate * moved, rp.cert_slack_min), rp.cert_slack_max);
				if (certified)
				{
					const bool matched = !((double)d0 > C.max_dist_sqr);
					M[k] = matched ? (int32_t)hj : -1;
					D0[k] = d0;
					hint2[gi] = make_int2((int32_t)hj, __float_as_int(lb_next));
					if (matched)
					{
						matched_cnt++;
						if ((int32_t)hj == r.pm)
							ST[k] |= MULLS_FS_STANDING;
						if (C.gate)
							claim(hj, s);
					}
				}
			}
		}
		if (!certified)
		{
			M[k] = MULLS_NEEDS_SEARCH;
			D0[k] = __int_as_float(r.pm); // (the sweep radius travels in the list entry; the tail wants the standing match)
			const uint32_t u = atomicAdd(&CL.ucount, 1u);
			const bool second = kc && hj < tgt_n; // a hinted point: the k-candidate certificate gets a look before the search (below)
			CL.uq[u] = out;
			CL.um[u] = moved;
			CL.us[u] = (uint16_t)((threadIdx.x + (uint32_t)k * BLK) | (second ? MULLS_US16_KCERT : 0u));
		}
	};

	Rec r0 = load(0), r1 = r0;
	if (ntrips > 1u)
		r1 = load(1);
	if (threadIdx.x == 0)
		CL.ucount = 0u;
	if (hashed)
		for (uint32_t i = threadIdx.x; i < MULLS_BIG_HASH; i += BLK)
			CL.hkey[i] = 0xffffffffu, CL.hval[i] = 0xffffffffu;
	__syncthreads();
	cert(0, r0);
	if (ntrips > 2u)
		r0 = load(2);
	if (ntrips > 1u)
		cert(1, r1);
	else
		ST[1] = 0u, M[1] = -1, D0[1] = 0.0f;
	if (ntrips > 2u)
		cert(2, r0);
	else
		ST[2] = 0u, M[2] = -1, D0[2] = 0.0f;
	if (!called)
		return; // correspondences of the previous iteration stay in force
	__syncthreads();
	uint32_t U = CL.ucount;
	// Second chance of the leftovers (the k-candidate certificates, kcert_list of the LDS tier for this tier's list): one lane per listed point gathers the
	// candidates its last search left behind and either certifies it — results where the search below would put them — or restores the sweep radius.  The
	// list is compacted in place, BLK entries per round (a round writes below what the later rounds still have to read).
	if (kc && U)
	{
		const uint32_t U0 = U;
		uint32_t kept_total = 0u, n_tried = 0u, n_pass = 0u;
		for (uint32_t base = 0; base < U0; base += BLK)
		{
			const uint32_t i = base + threadIdx.x;
			float4 e = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			uint32_t es = 0u;
			bool keep = false;
			if (i < U0)
			{
				e = CL.uq[i];
				es = CL.us[i];
				keep = true;
				if (es & MULLS_US16_KCERT)
				{
					es &= ~MULLS_US16_KCERT;
					n_tried++;
					const uint32_t s = q0 + es, gi = d.src_off + s;
					const uint4 cr = C.cand[gi];
					const int2 h = hint2[gi];
					nnkey bk;
					float lb_next;
					uint4 cr_next;
					const float moved = CL.um[i];
					const float4 *__restrict__ tp0 = tpos + d.tgt_off; // (read once: through `d` inside the accessor it is re-loaded before every gather)
					auto pre = [&](uint32_t t) -> uint32_t { return t; };
					auto pos = [&](uint32_t t) -> float4 { return tp0[t]; };
					if (kcert_point<false>(rp, cr, h, (uint32_t)ps.iter, e.x, e.y, e.z, moved, tgt_n, pre, pos, bk, lb_next, cr_next))
					{
						const float best = key_dist(bk);
						const uint32_t bi = (uint32_t)bk;
						const bool matched = !((double)best > C.max_dist_sqr);
						if (class_level)
							CL.res[es] = make_uint2(matched ? bi : 0xffffffffu, __float_as_uint(best));
						else
						{
							nn_idx[gi] = matched ? (int32_t)bi : -1;
							nn_d2[gi] = best;
						}
						hint2[gi] = make_int2((int32_t)bi, __float_as_int(lb_next));
						C.cand[gi] = cr_next;
						if (matched)
						{
							matched_cnt++;
							if (C.gate)
								claim(bi, s);
						}
						keep = false;
						n_pass++;
					}
				}
			}
			const unsigned long long bal = __ballot(keep);
			const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
			if (lane == 0u)
				CL.red[wave] = (uint32_t)__popcll(bal);
			__syncthreads(); // every entry of this round has been read
			uint32_t pre = 0u, total = 0u;
			for (uint32_t w = 0; w < (uint32_t)(BLK / 64); w++)
			{
				const uint32_t c = CL.red[w];
				pre += w < wave ? c : 0u;
				total += c;
			}
			if (keep)
			{
				const uint32_t k = kept_total + pre + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
				CL.uq[k] = e;
				CL.us[k] = (uint16_t)es;
			}
			kept_total += total;
			__syncthreads();
		}
		U = kept_total;
		if (rp.dbg_ticks && rp.debug_stop == 21u) // diagnostics: points that got the second chance, points it certified
		{
			for (int off = 32; off > 0; off >>= 1)
				n_tried += __shfl_down(n_tried, off), n_pass += __shfl_down(n_pass, off);
			if ((threadIdx.x & 63u) == 0u && n_tried)
			{
				atomicAdd(&rp.dbg_ticks[14], (unsigned long long)n_tried);
				atomicAdd(&rp.dbg_ticks[15], (unsigned long long)n_pass);
			}
		}
	}
	// the leftovers against the bitmap grid: 16-lane sub-groups, one query at a time each
	if (U)
	{
		const bool dbg = rp.dbg_ticks && rp.debug_stop == 21u; // diagnostics: leftover queries and the workgroup time they take
		const unsigned long long t_dbg = dbg ? wall_clock64() : 0ull;
		const uint32_t sub = threadIdx.x & (MULLS_GRID_GROUP - 1u), grp = threadIdx.x / MULLS_GRID_GROUP;
		const bool probe_own = tgt_n >= 4u * g.nocc; // dense cells (a map of many scans): the own cell first; a scan's own density: straight to the cube
		for (uint32_t i = grp; i < U; i += BLK / MULLS_GRID_GROUP)
		{
			nnkey bk;
			float sec, Rfin;
			CandOut co = {0xffffffffu, 0xffffffffu, 0.0f};
			search_query_bm(g, B, ts, CL.uq[i], C.r, C.m, sub, probe_own, bk, sec, Rfin, co);
			if (sub == 0)
			{
				const uint32_t slot = CL.us[i] & ~MULLS_US16_KCERT, s = q0 + slot, gi = d.src_off + s;
				const float best = key_dist(bk);
				const int bi = (int)(uint32_t)bk; // -1: nothing found
				const bool matched = bi >= 0 && !((double)best > C.max_dist_sqr);
				if (class_level)
					CL.res[slot] = make_uint2(matched ? (uint32_t)bi : 0xffffffffu, __float_as_uint(best));
				else
				{
					nn_idx[gi] = matched ? bi : -1;
					nn_d2[gi] = best;
				}
				const float lb_new = fminf(sqrtf(sec), Rfin);
				hint2[gi] = make_int2(bi, __float_as_int(lb_new)); // every target but the one found is at least that far away
				if (C.cand) // the other lanes' nearest targets, and how much farther everything outside {result, candidates} lies (co.b2 >= sec)
					C.cand[gi] = make_uint4(co.cx, co.cy, __float_as_uint(fminf(sqrtf(co.b2), Rfin) - lb_new), C.epoch);
				if (matched)
				{
					matched_cnt++;
					if (C.gate)
						claim((uint32_t)bi, s);
				}
			}
		}
		if (dbg)
		{
			__syncthreads();
			if (threadIdx.x == 0)
			{
				atomicAdd(&rp.dbg_ticks[13], (unsigned long long)U);
				atomicAdd(&rp.dbg_ticks[7], wall_clock64() - t_dbg);
			}
		}
	}
	for (int off = 32; off > 0; off >>= 1)
		matched_cnt += __shfl_down(matched_cnt, off);
	if (!class_level)
	{
		// chunk-level job: k_filter reads every live point's result from memory
#pragma unroll
		for (int k = 0; k < TRIPS; k++)
			if ((ST[k] & MULLS_FS_ALIVE) && M[k] != MULLS_NEEDS_SEARCH)
			{
				const uint32_t gi = d.src_off + q0 + threadIdx.x + (uint32_t)k * BLK;
				nn_idx[gi] = M[k];
				nn_d2[gi] = D0[k];
			}
		// the class's |Corr| > 0 flag (k_filter's any_match: the count itself is never read) — one plain store per workgroup: an atomicAdd per wave was
		// 1700 atomics on one word for a 108 k-point class cloud, 20 us of the launch's 35 (profiles/r04_large_steps.txt)
		if ((threadIdx.x & 63) == 0)
			CL.red[threadIdx.x >> 6] = matched_cnt;
		__syncthreads();
		if (threadIdx.x == 0)
		{
			uint32_t any = 0;
			for (int w = 0; w < BLK / 64; w++)
				any |= CL.red[w];
			if (any)
				d.n_matched = 1u;
		}
		return;
	}
	if ((threadIdx.x & 63) == 0)
		CL.red[threadIdx.x >> 6] = matched_cnt;
	__threadfence_block(); // the searched points' results and the duplicate table: LDS (class-level jobs only reach this)
	__syncthreads();
	uint32_t total_matched = 0;
	for (int w = 0; w < BLK / 64; w++)
		total_matched += CL.red[w];

	// duplicate rule + rejection chain (filter_point's decisions) on the registers
	const float thr = ps.thr[cls], max_sqr = thr * thr; // CorrespondenceRejectorDistance::setMaximumDistance (float)
	const bool any_match = total_matched > 0u, strict = rp.rej_strict != 0;
	uint32_t n_alive = 0, n_valid = 0;
#pragma unroll
	for (int k = 0; k < TRIPS; k++)
	{
		const uint32_t st = ST[k];
		if (!(st & MULLS_FS_ALIVE))
			continue;
		const uint32_t s = q0 + threadIdx.x + (uint32_t)k * BLK, gi = d.src_off + s;
		int32_t m = M[k];
		float dist = D0[k];
		bool standing = (st & MULLS_FS_STANDING) != 0;
		if (m == MULLS_NEEDS_SEARCH)
		{
			const uint2 found = CL.res[threadIdx.x + (uint32_t)k * BLK];
			m = (int32_t)found.x;
			standing = m >= 0 && __float_as_int(dist) == m; // (D0 of a waiting point: its standing match)
			dist = __uint_as_float(found.y);
		}
		// first source (lowest index) matched to a target keeps it (cregistration.hpp:1762-1789); the others become unmatched
		if (hashed && m >= 0 && big_hash_get(CL, (uint32_t)m) != s)
			m = -1;
		bool alive = true, valid, dir_ok = (st & MULLS_FS_DIR_OK) != 0;
		if (any_match)
		{
			valid = m >= 0;
			if (C.gate && m < 0)
			{
				alive = false; // unmatched or duplicate-losing source points vanish for good (:1762-1789)
				valid = false;
			}
			if (valid)
			{
				valid = strict ? dist < max_sqr : !(dist > max_sqr); // CorrespondenceRejectorDistance (see mulls_params.rejector_strict)
				if (valid)
				{
					wd[gi] = dist; // pcl::Correspondence::distance (shares storage with ::weight)
					if (!standing)
					{
						// a new correspondence: its target record travels with the source point from here on (filter_point), and the direction check
						// runs against the new target direction
						match[gi] = m;
						const float4 q2 = tpos[d.tgt_off + (uint32_t)m], n2 = tnrm[d.tgt_off + (uint32_t)m];
						mq[2u * gi] = q2;
						mq[2u * gi + 1u] = n2;
						if (normal_check)
						{
							const float3 n1 = *reinterpret_cast<const float3 *>(snrm + gi); // this lane's own store of phase 1
							const double dot = (double)n1.x * (double)n2.x + (double)n1.y * (double)n2.y + (double)n1.z * (double)n2.z;
							const float c = (float)fabs(dot);
							dir_ok = !((double)c < rp.cos_bearing);
						}
					}
				}
			}
		}
		else if (C.gate)
		{
			alive = false; // the whole cloud was swapped for an empty one; reference behaviour undefined, see oracle
			valid = false;
		}
		else
			valid = (st & MULLS_FS_VALID) != 0; // the previous Corr_f is still in place (SURVEY B-4) and goes through the direction check again
		if (valid && normal_check && !dir_ok)
			valid = false;
		const uint32_t nf = (alive ? MULLS_F_ALIVE : 0u) | (valid ? MULLS_F_VALID : 0u);
		if (nf != (st & (MULLS_FS_ALIVE | MULLS_FS_VALID)))
			flag[gi] = (uint8_t)nf;
		n_alive += alive ? 1u : 0u;
		n_valid += valid ? 1u : 0u;
	}
	for (int off = 32; off > 0; off >>= 1)
	{
		n_alive += __shfl_down(n_alive, off);
		n_valid += __shfl_down(n_valid, off);
	}
	if ((threadIdx.x & 63) == 0)
	{
		CL.red[BLK / 64 + (threadIdx.x >> 6)] = n_alive;
		CL.red[2 * (BLK / 64) + (threadIdx.x >> 6)] = n_valid;
	}
	__syncthreads();
	if (threadIdx.x == 0)
	{
		uint32_t ta = 0, tv = 0;
		for (int w = 0; w < BLK / 64; w++)
		{
			ta += CL.red[BLK / 64 + w];
			tv += CL.red[2 * (BLK / 64) + w];
		}
		d.n_matched = total_matched; // k_finish reset it to 0 after the previous iteration
		d.alive_next = ta;
		d.valid_next = tv;
		d.n_search = U;
	}
}
} // namespace
