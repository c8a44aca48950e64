// lds_tier.h — device code of the LDS grid tier of the correspondence search, shared by the lock-step kernels (k_cert / k_nn_lds /
// k_filter in k_search.hip) and the device-resident registration loop (k_icp.hip): the rejection chain of one source point, the
// exact grid search of one query by a sub-group of lanes, and the two per-class-cloud passes of an ICP iteration (cert_class:
// rigid step + certificates + leftovers; lds_search_class: target cloud staged in LDS + search of the uncertified points).
#pragma once
#include "device_util.h"
// ---------------------------------------------------------------------------------------------------------------
// Rejection chain of determine_corres after the search (cregistration.hpp:1755-1830), one source point.
// `dedup_done`: the duplicate rule has been applied already (losers carry nn_idx = -1, k_nn_lds with rp.lds_dedup).
struct FilterCtx
{
	bool gate, any_match, normal_check, dedup_done, strict;
	float max_sqr;	 // CorrespondenceRejectorDistance::setMaximumDistance (float)
	double cos_thre; // cos(angle_thre_degree / 180.0 * M_PI), evaluated on the host (:1818)
	unsigned long long key_hi;
	const float4 *tgt_stage; // RunParams::tgt_stage / tgt_map (tgt_record)
	const uint16_t *tgt_map;
};
__device__ __forceinline__ void filter_point(const FilterCtx &F, const CloudDesc &d, uint32_t s, const float4 *__restrict__ snrm,
											  const float4 *__restrict__ tnrm, uint8_t *__restrict__ flag, const int32_t *__restrict__ nn_idx,
											  const float *__restrict__ nn_d2, int32_t *__restrict__ match, float *__restrict__ wd,
											  const unsigned long long *__restrict__ winner, const float4 *__restrict__ tpos,
											  float4 *__restrict__ mq, uint32_t &n_alive, uint32_t &n_valid)
{
	const uint32_t g = d.src_off + s;
	const uint32_t f = flag[g];
	if (!(f & MULLS_F_ALIVE))
		return;
	bool alive = true, valid, fresh = false;
	int m;
	float4 n2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	if (F.any_match)
	{
		m = nn_idx[g];
		valid = m >= 0;
		if (F.gate && (m < 0 || (!F.dedup_done && winner[d.tgt_off + m] != (F.key_hi | (unsigned long long)s))))
		{
			alive = false; // unmatched or duplicate-losing source points vanish for good (:1762-1789)
			valid = false;
		}
		if (valid)
		{
			const float dist = nn_d2[g];
			valid = F.strict ? dist < F.max_sqr : !(dist > F.max_sqr); // CorrespondenceRejectorDistance (see mulls_params.rejector_strict)
			if (valid)
			{
				wd[g] = dist; // pcl::Correspondence::distance (shares storage with ::weight)
				// the matched target travels with the source point from here on: k_accum streams (position, direction) records
				// instead of gathering two cache lines per correspondence (its launches were bound by exactly that traffic).
				// From the second iteration on most points keep their target: the record is already there (it is only ever
				// written together with match[]), so neither gather nor store is needed — one coalesced 16-B read instead.
				if (match[g] == m)
					n2 = mq[2u * g + 1u];
				else
				{
					match[g] = m;
					float4 q2;
					tgt_record(F.tgt_stage, F.tgt_map, d, (uint32_t)m, tpos, tnrm, q2, n2);
					mq[2u * g] = q2;
					mq[2u * g + 1u] = n2;
				}
				fresh = true;
			}
		}
	}
	else if (F.gate)
	{
		alive = false; // the whole cloud was swapped for an empty one; reference behaviour undefined, see oracle
		valid = false;
		m = -1;
	}
	else
	{
		// CorrespondenceRejectorDistance::getCorrespondences returned early on the empty input: the previous
		// Corr_f is still in place (SURVEY B-4) and goes through the direction check again.
		valid = (f & MULLS_F_VALID) != 0;
		m = match[g];
	}
	if (valid && F.normal_check)
	{
		const float4 n1 = snrm[g];
		if (!fresh)
			n2 = mq[2u * g + 1u]; // the standing correspondence's target direction
		const double dot = (double)n1.x * (double)n2.x + (double)n1.y * (double)n2.y + (double)n1.z * (double)n2.z;
		const float c = (float)fabs(dot);
		if ((double)c < F.cos_thre)
			valid = false;
	}
	flag[g] = (uint8_t)((alive ? MULLS_F_ALIVE : 0u) | (valid ? MULLS_F_VALID : 0u));
	n_alive += alive ? 1u : 0u;
	n_valid += valid ? 1u : 0u;
}
// ---------------------------------------------------------------------------------------------------------------
// Correspondence search, LDS grid tier — the default whenever every searched target class cloud holds at most
// MULLS_LDS_MAXPTS points (the reference-default KITTI sizes).  Two kernels per ICP iteration:
//
//   k_cert     light pass, 512 lanes per (pair, class), several workgroups per CU.  One lane per source point: this
//              iteration's rigid step (cregistration.hpp:1690-1695) and the CERTIFICATE — a point whose previous nearest
//              target is provably still its nearest one (triangle inequality on a bound its last search left behind) gets its
//              correspondence without a search.  As the registration converges the steps shrink and from the third or fourth
//              iteration on 95-100 % of the points certify.  A class cloud with at most MULLS_CERT_SMALL points left over
//              searches them right here against the grid in global memory (8-lane sub-groups) and runs the duplicate rule and
//              the rejection chain; otherwise it queues itself for
//   k_nn_lds   heavy pass, one persistent 1024-lane workgroup per CU taking class clouds from that queue: brings the whole
//              cell-sorted target class cloud on chip once with coalesced 16-B loads (12-B position records, a uint16 original
//              index per point, a uint16 cell table — as many cells as the rest of the 160 KiB allows) and searches the
//              uncertified points against it in equal chunks of at most MULLS_LDS_QCHUNK queries:
//     cost order   counting sort of the chunk's queries by the candidate trips they took last time (kept next to the hint):
//                  the eight sub-groups of a wave run in lock step, so neighbours should cost the same;
//     search       8-lane sub-groups, one query each: the cube around the query whose radius is the distance to last
//                  iteration's nearest target (an exact upper bound: the hint) plus a slack, two rows of cells per step, their
//                  candidate ranges laid end to end, two candidates per lane in flight, state = one 64-bit (distance bits,
//                  index) key and the second-smallest distance, DPP minima.  Unhinted queries probe their own cell first;
//                  the 2.5*thr ball is swept only if nothing lies within one cell edge;
//     tail         duplicate rule in an LDS table, then the rejection chain (filter_point) on the results while they are
//                  still in cache, and the matched target's record for k_accum.
// Why the certificate is exact: a search examines every target whose cell meets the cube [p - R, p + R], i.e. every target
// within distance R of p, and reports the nearest one j.  It leaves lb = min(second-smallest distance examined, R): every
// target other than j is at least lb away from p.  When the point moves to p' (|p' - p| = moved, computed from the two float
// positions), those targets are at least lb - moved away from p'.  If dist(p', j) < lb - moved — tested with 1e-5 relative
// slack on both sides, two orders of magnitude above the rounding of the float expressions — j is the unique nearest
// neighbour of p', and its squared distance is evaluated with the very expression a search uses: same index, same bits.
// Ties and near-ties (equal distances, duplicate target points) fail the test and are searched, where the lowest index wins.
// The bound then travels on as lb - moved.  Same exactness argument for the search itself, same outputs as k_nn / k_nn_grid.
namespace
{
// Search state of a query: (distance bits << 32) | target index.  Distances are sums of squares (>= +0, or NaN), so the
// unsigned order of the key is the lexicographic (distance, index) order the tie rule asks for, NaN keys sort after
// NNKEY_NONE and are never taken, and one 64-bit compare + two selects update the running minimum.
typedef unsigned long long nnkey;
#define NNKEY_NONE 0x7f800000ffffffffull
__device__ __forceinline__ nnkey nn_key(float dist, uint32_t idx) { return ((nnkey)__float_as_uint(dist) << 32) | idx; }
__device__ __forceinline__ float key_dist(nnkey k) { return __uint_as_float((uint32_t)(k >> 32)); }
__device__ __forceinline__ bool key_found(nnkey k) { return (uint32_t)k != 0xffffffffu; }

// the target class cloud as the sub-groups see it: staged in LDS (k_nn_lds) ...
struct LdsGrid
{
	const float *P;	// staged target positions, 12-B records (x, y, z): one address, three immediate offsets, conflict-free stride
	const uint16_t *IDX, *CS;
	__device__ __forceinline__ uint32_t cs(uint32_t i) const { return CS[i]; }
	__device__ __forceinline__ void cand(uint32_t t, float &x, float &y, float &z, uint32_t &i) const
	{
		const float *p = P + 3u * t;
		x = p[0], y = p[1], z = p[2];
		i = IDX[t];
	}
};
// ... or where k_grid_build_sort left it in global memory (k_cert's handful of leftover queries): cell-sorted float4 records with
// the original index in .w, and the same uint16 cell table
struct GlobGrid
{
	const float4 *ts;
	const uint16_t *CS;
	__device__ __forceinline__ uint32_t cs(uint32_t i) const { return CS[i]; }
	__device__ __forceinline__ void cand(uint32_t t, float &x, float &y, float &z, uint32_t &i) const
	{
		const float4 v = ts[t];
		x = v.x, y = v.y, z = v.z;
		i = __float_as_uint(v.w);
	}
};

// minimum over the lanes of a sub-group, VALU only (DPP; no LDS-crossbar shuffles): quad xor-1, quad xor-2, half-row
// mirror (8 lanes), row mirror (16 lanes).  Result in every lane.
template <int CTRL>
__device__ __forceinline__ void dpp_min_step(nnkey &bk)
{
	const uint32_t oh = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(bk >> 32), CTRL, 0xf, 0xf, false);
	const uint32_t ol = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)bk, CTRL, 0xf, 0xf, false);
	const nnkey o = ((nnkey)oh << 32) | ol;
	bk = o < bk ? o : bk;
}
// GSH: log2 of the lanes that share a query — a template argument (MULLS_LDS_GSH) or, < 0, the run-time value `gsh` (uniform over the workgroup).  The run-time
// form served one experiment — the light pass choosing 4 or 2 lanes per query when its leftover list is longer than one round of 8-lane sub-groups: slower,
// profiles/r06_experiments.txt item 6 — and no kernel uses it
#define MULLS_LDS_GSH (MULLS_LDS_GROUP == 16u ? 4 : (MULLS_LDS_GROUP == 8u ? 3 : 2))
template <int GSH>
__device__ __forceinline__ uint32_t sub_lanes(uint32_t gsh) { return GSH >= 0 ? (1u << (GSH >= 0 ? GSH : 0)) : (1u << gsh); }
template <int GSH = MULLS_LDS_GSH>
__device__ __forceinline__ void row16_min(nnkey &bk, uint32_t gsh = 0u)
{
	const uint32_t GL = sub_lanes<GSH>(gsh);
	dpp_min_step<0xB1>(bk); // quad_perm [1,0,3,2]
	if (GL >= 4u)
		dpp_min_step<0x4E>(bk); // quad_perm [2,3,0,1]
	if (GL >= 8u)
		dpp_min_step<0x141>(bk); // row_half_mirror: lanes i <-> 7 - i of each 8-lane half
	if (GL >= 16u)
		dpp_min_step<0x140>(bk); // row_mirror: lanes i <-> 15 - i
}
// (best key, second-best distance) over the lanes of a sub-group.  Every lane enters with the best key and the second-smallest
// distance among ITS candidates; a lane whose best lost the sub-group minimum contributes that best's distance instead.
template <int CTRL>
__device__ __forceinline__ void dpp_fmin_step(float &v)
{
	const float o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
	v = fminf(o, v);
}
template <int GSH = MULLS_LDS_GSH>
__device__ __forceinline__ void row16_min2(nnkey &bk, float &sec, uint32_t gsh = 0u)
{
	const uint32_t GL = sub_lanes<GSH>(gsh);
	nnkey g = bk;
	row16_min<GSH>(g, gsh);
	float c = (bk == g) ? sec : key_dist(bk);
	dpp_fmin_step<0xB1>(c);
	if (GL >= 4u)
		dpp_fmin_step<0x4E>(c);
	if (GL >= 8u)
		dpp_fmin_step<0x141>(c);
	if (GL >= 16u)
		dpp_fmin_step<0x140>(c);
	sec = c;
	bk = g;
}

// one candidate pair of a lane: `sec` follows the second-smallest distance this lane has seen (one v_med3_f32 per candidate: the
// median of (second, candidate, best) is the new second whichever of the three orders holds); with !ok2 the second candidate is
// the first one again — no effect on the best, and kept away from the second-best
__device__ __forceinline__ void take_pair(float da, uint32_t ia, float db, uint32_t ib, bool ok2, nnkey &bk, float &sec)
{
	const nnkey ka = nn_key(da, ia), kb = nn_key(db, ib);
	sec = __builtin_amdgcn_fmed3f(sec, da, key_dist(bk));
	bk = ka < bk ? ka : bk;
	sec = __builtin_amdgcn_fmed3f(sec, ok2 ? db : __builtin_inff(), key_dist(bk));
	bk = kb < bk ? kb : bk;
}

// ---- k-candidate certificates (round 5): what a search leaves behind besides its result -------------------------------------------------------
// The lanes of a sub-group evaluate disjoint shares of the candidates of a sweep.  Lane l ends with its own nearest key lk_l and the
// second-smallest distance ls_l among ITS candidates, so every target the sweep evaluated is either some lane's best or at least
// min_l ls_l away.  The bests of the other lanes are therefore a ready-made candidate set — no work per candidate, a few DPP moves per query:
// the NO nearest of them (by key) go into the record, and b2 = min(min_l ls_l, the nearest lane best that did not fit) bounds every evaluated
// target outside {the result, the record}.  (On 8 lanes that set is about as good as the true 3 nearest: profiles/r05_kcert_study.txt.)
// The k-candidate certificates on the LDS tier are compiled OUT by default.  Measured (profiles/r05_experiments.txt item 6, 4096 pairs, search ms per step):
// records kept + look 11.92, records kept without the look 12.03, neither 11.46 — keeping the records (lane_bests: ~70 VALU instructions per search and lane,
// 24 M searches a step, half of them in the VALU-bound staged pass) costs 0.57 ms and the look returns 0.11.  The global-memory tier keeps them (+4 ... 8 % there).
// -DMULLS_LDS_KCERT=1 builds the LDS tier with them (tools/build_variant.sh; same bits either way: tests/test_gpu_icp.py runs against MULLS_HIP_LIB).
#ifndef MULLS_LDS_KCERT
#define MULLS_LDS_KCERT 0
#endif
struct CandOut
{
	uint32_t cx, cy; // candidates besides the result: LDS tier 4 x uint16 (0xffff = none), global-memory tier 2 x uint32 (0xffffffff = none)
	float b2;		 // squared distance every evaluated target outside {result, candidates} keeps at least (0: nothing claimed)
};
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
// the value of lane (i ^ J) of an 8-lane sub-group / of lane (i + J) mod 16 of a 16-lane sub-group (quad permutes, row_half_mirror, row_ror)
template <int G, int J>
__device__ __forceinline__ uint32_t sub_peer(uint32_t v)
{
	static_assert(G == 8 || G == 16, "sub-groups of 8 or 16 lanes");
	if constexpr (G == 16)
		return dpp_mov<0x120 + J>(v);
	else if constexpr (J == 1)
		return dpp_mov<0xB1>(v);
	else if constexpr (J == 2)
		return dpp_mov<0x4E>(v);
	else if constexpr (J == 3)
		return dpp_mov<0x1B>(v);
	else if constexpr (J == 7)
		return dpp_mov<0x141>(v);
	else
		return dpp_mov<0x141>(sub_peer<8, 7 - J>(v)); // i ^ J = (i ^ 7) ^ (7 - J) for J = 4, 5, 6
}
template <int G, int J>
struct RankLoop
{
	static __device__ __forceinline__ void run(uint32_t kh, uint32_t kl, nnkey lk, uint32_t &rank)
	{
		const nnkey o = ((nnkey)sub_peer<G, J>(kh) << 32) | sub_peer<G, J>(kl);
		rank += o < lk ? 1u : 0u;
		RankLoop<G, J + 1>::run(kh, kl, lk, rank);
	}
};
template <int G>
struct RankLoop<G, G>
{
	static __device__ __forceinline__ void run(uint32_t, uint32_t, nnkey, uint32_t &) {}
};
template <int CTRL>
__device__ __forceinline__ void dpp_and_step(uint32_t &v)
{
	v &= dpp_mov<CTRL>(v);
}
// lk / ls: this lane's own best key and second-smallest distance of the sweep whose merged result stands.  Every lane of the sub-group calls it;
// the result is valid in every lane.  Keys of different lanes differ unless both are NNKEY_NONE (a target is evaluated by one lane).
template <int G, bool W16>
__device__ __forceinline__ CandOut lane_bests(nnkey lk, float ls)
{
	constexpr uint32_t NO = W16 ? 4u : 2u;
	uint32_t rank = 0;
	RankLoop<G, 1>::run((uint32_t)(lk >> 32), (uint32_t)lk, lk, rank); // lanes of the sub-group with a smaller key
	float c = fminf(ls, rank > NO ? key_dist(lk) : __builtin_inff());
	dpp_fmin_step<0xB1>(c);
	dpp_fmin_step<0x4E>(c);
	dpp_fmin_step<0x141>(c);
	if constexpr (G == 16)
		dpp_fmin_step<0x140>(c);
	// (selects, not conditional stores: the compiler turns `if (..) vx = w; else vy = w;` into a dynamically indexed stack array)
	const bool fits = key_found(lk) && rank >= 1u && rank <= NO;
	const uint32_t idx = (uint32_t)lk;
	uint32_t vx, vy;
	if (W16)
	{
		const uint32_t sh = ((rank - 1u) & 1u) << 4, w = ~(0xffffu << sh) | ((idx & 0xffffu) << sh);
		vx = fits && rank <= 2u ? w : 0xffffffffu;
		vy = fits && rank > 2u ? w : 0xffffffffu;
	}
	else
	{
		vx = fits && rank == 1u ? idx : 0xffffffffu;
		vy = fits && rank == 2u ? idx : 0xffffffffu;
	}
	dpp_and_step<0xB1>(vx), dpp_and_step<0xB1>(vy);
	dpp_and_step<0x4E>(vx), dpp_and_step<0x4E>(vy);
	dpp_and_step<0x141>(vx), dpp_and_step<0x141>(vy);
	if constexpr (G == 16)
		dpp_and_step<0x140>(vx), dpp_and_step<0x140>(vy);
	CandOut co = {0xffffffffu, 0xffffffffu, 0.0f};
	co.cx = vx, co.cy = vy, co.b2 = c;
	return co;
}

// Evaluate every target in the cells intersecting the cube [p - R, p + R] (same exactness argument as grid_scan_box).  The
// rows (x-runs of cells, contiguous in the sorted cloud) are taken two at a time: every lane of the sub-group reads their
// bounds (same addresses: LDS broadcast), the candidate ranges are laid end to end and the sub-group strides over the
// concatenation, two candidates per lane in flight — the trip count is that of the total, not the sum of the per-row
// round-ups, and the only per-row work is two table reads and a running sum.  Returns false when the cube lies inside the
// query's own cell and `own_done` says that cell has been swept already.
// CHUNK = rows per step.  2 against the cloud staged in LDS (k_nn_lds is bound by its instruction count: every further row costs a select per candidate).  4 against
// the grid in global memory (the light pass's leftovers, bound by the latency of their chains: a hinted cube spans at most two cells per axis, i.e. at most four
// rows — with four rows per step a round of sub-groups is one table trip and one or two candidate trips where two rows per step took two of each as soon as
// ONE of its 64 queries straddled a cell boundary in y and in z).
#define MULLS_LDS_CHUNK 2
#ifndef MULLS_GLOB_CHUNK
#define MULLS_GLOB_CHUNK 2
#endif
template <int CHUNK, int GSH, class G>
__device__ __forceinline__ bool lds_scan_box(const GridDesc &g, const G &L, float px, float py, float pz, float R, uint32_t sub, nnkey &bk,
											  float &sec, uint32_t &trips, bool own_done = false, uint32_t gsh = 0u)
{
	const uint32_t GL = sub_lanes<GSH>(gsh), GS = GSH >= 0 ? (uint32_t)(GSH >= 0 ? GSH : 0) : gsh;
	static_assert(CHUNK == 1 || CHUNK == 2 || CHUNK == 4, "rows per step");
	const float Rm = R * 1.0001f + 1e-4f;
	const uint32_t x0 = (uint32_t)grid_cell(px - Rm, g.ox, g.inv_h, g.nx), x1 = (uint32_t)grid_cell(px + Rm, g.ox, g.inv_h, g.nx) + 1u;
	const int y0 = grid_cell(py - Rm, g.oy, g.inv_h, g.ny), y1 = grid_cell(py + Rm, g.oy, g.inv_h, g.ny);
	const int z0 = grid_cell(pz - Rm, g.oz, g.inv_h, g.nz), z1 = grid_cell(pz + Rm, g.oz, g.inv_h, g.nz);
	if (own_done && x1 - x0 == 1u && y0 == y1 && z0 == z1)
		return false; // the cube stays inside the query's own cell, which has been swept already
	int cy = y0, cz = z0;
	while (cz <= z1)
	{
		uint32_t lo[CHUNK], pre[CHUNK], acc = 0;
		uint32_t a[CHUNK], e[CHUNK];
		bool valid[CHUNK];
#pragma unroll
		for (int jj = 0; jj < CHUNK; jj++) // every row's bounds requested before the first is used
		{
			valid[jj] = cz <= z1;
			const uint32_t row = ((uint32_t)(valid[jj] ? cz : z0) * g.ny + (uint32_t)cy) * g.nx;
			a[jj] = L.cs(row + x0), e[jj] = L.cs(row + x1);
			if (++cy > y1)
			{
				cy = y0;
				cz++;
			}
		}
#pragma unroll
		for (int jj = 0; jj < CHUNK; jj++)
		{
			lo[jj] = a[jj] - acc; // candidate f of the concatenation lives at lo[jj] + f while f < pre[jj]
			acc += valid[jj] ? e[jj] - a[jj] : 0u;
			pre[jj] = acc;
		}
		trips += (acc + GL - 1u) >> GS;
		for (uint32_t f = sub; f < acc; f += 2u * GL)
		{
			const uint32_t f2 = f + GL;
			const bool ok2 = f2 < acc;
			const uint32_t ff = ok2 ? f2 : f;
			uint32_t ta, tb;
			if constexpr (CHUNK == 1)
				ta = f + lo[0], tb = ff + lo[0];
			else if constexpr (CHUNK == 2)
			{
				ta = f + (f < pre[0] ? lo[0] : lo[1]);
				tb = ff + (ff < pre[0] ? lo[0] : lo[1]);
			}
			else
			{
				ta = f + (f < pre[1] ? (f < pre[0] ? lo[0] : lo[1]) : (f < pre[2] ? lo[2] : lo[3]));
				tb = ff + (ff < pre[1] ? (ff < pre[0] ? lo[0] : lo[1]) : (ff < pre[2] ? lo[2] : lo[3]));
			}
			float ax, ay, az, bx, by, bz;
			uint32_t ia, ib;
			L.cand(ta, ax, ay, az, ia);
			L.cand(tb, bx, by, bz, ib);
			float dx = px - ax, dy = py - ay, dz = pz - az;
			const float da = (dx * dx + dy * dy) + dz * dz; // L2_Simple<float>, no FMA
			dx = px - bx, dy = py - by, dz = pz - bz;
			const float db = (dx * dx + dy * dy) + dz * dz;
			take_pair(da, ia, db, ib, ok2, bk, sec);
		}
	}
	return true;
}

// The exact nearest target of one query (q.xyz; q.w = radius of the hinted sweep, +inf = no hint) by the MULLS_LDS_GROUP lanes of
// a sub-group.  Out: bk = (squared distance, original index) of the nearest target or NNKEY_NONE; sec = second-smallest squared
// distance the last sweep saw (0 = unknown), Rfin = that sweep's radius: every target other than bk's is at least
// min(sqrt(sec), Rfin) away; trips = candidate trips taken (cost class of the next iteration).
// co: the other lanes' nearest targets of the sweep that produced (bk, sec), for the k-candidate certificates (lane_bests; b2 = 0 with sec = 0).
template <int CHUNK = MULLS_LDS_CHUNK, int GSH = MULLS_LDS_GSH, class G>
__device__ __forceinline__ void search_query(const GridDesc &g, const G &L, const float4 q, float r, float m, uint32_t sub, nnkey &bk, float &sec,
											  float &Rfin, uint32_t &trips, CandOut &co, uint32_t gsh = 0u)
{
	const uint32_t GL = sub_lanes<GSH>(gsh), GS = GSH >= 0 ? (uint32_t)(GSH >= 0 ? GSH : 0) : gsh;
	bk = NNKEY_NONE;
	sec = __builtin_inff();
	Rfin = 0.0f;
	trips = 0u;
	nnkey lane_k = NNKEY_NONE; // this lane's own share of the standing sweep
	float lane_s = __builtin_inff();
	// One sweep of the cube of radius R.  Its cells are a superset of every earlier sweep's cells, so its own (best, second)
	// pair replaces the standing one; only when the radius was clipped to the rejection radius can the standing best lie
	// outside — it stays the answer then, and nothing is claimed about the other targets.
	auto sweep = [&](float R, bool own_done) {
		nnkey lk = NNKEY_NONE;
		float ls = __builtin_inff();
		if (lds_scan_box<CHUNK, GSH>(g, L, q.x, q.y, q.z, R, sub, lk, ls, trips, own_done, gsh))
		{
			const nnkey own_k = lk;
			const float own_s = ls;
			row16_min2<GSH>(lk, ls, gsh);
			// (selects: two branches ending in stores to different variables are merged into one store through a selected address, which puts both on the stack)
			const bool adopt = !(bk < lk);
			bk = adopt ? lk : bk;
			sec = adopt ? ls : 0.0f;
			lane_k = adopt ? own_k : lane_k;
			lane_s = adopt ? own_s : lane_s;
		}
		Rfin = R;
	};
	if (q.w < __builtin_inff())
		sweep(fminf(m, q.w), false); // bounded by last iteration's correspondence: the cube contains that target
	else
	{
		// probe 0: the query's own cell.  In dense regions (tens of targets per cell) this already yields a tight bound.
		const int cx = grid_cell(q.x, g.ox, g.inv_h, g.nx), cy = grid_cell(q.y, g.oy, g.inv_h, g.ny), cz = grid_cell(q.z, g.oz, g.inv_h, g.nz);
		const uint32_t cell = ((uint32_t)cz * g.ny + (uint32_t)cy) * g.nx + (uint32_t)cx;
		const uint32_t lo = L.cs(cell), hi = L.cs(cell + 1u);
		trips += (hi - lo + GL - 1u) >> GS;
		for (uint32_t t = lo + sub; t < hi; t += 2u * GL) // two candidates in flight per lane and trip
		{
			const uint32_t t2 = t + GL;
			const bool ok2 = t2 < hi;
			float ax, ay, az, bx, by, bz;
			uint32_t ia, ib;
			L.cand(t, ax, ay, az, ia);
			L.cand(ok2 ? t2 : t, bx, by, bz, ib);
			float dx = q.x - ax, dy = q.y - ay, dz = q.z - az;
			const float da = (dx * dx + dy * dy) + dz * dz;
			dx = q.x - bx, dy = q.y - by, dz = q.z - bz;
			const float db = (dx * dx + dy * dy) + dz * dz;
			take_pair(da, ia, db, ib, ok2, bk, sec);
		}
		lane_k = bk, lane_s = sec;
		row16_min2<GSH>(bk, sec, gsh);
		// probe 1: every cell within min(first-probe radius, current best distance) of the query
		sweep(key_found(bk) ? fminf(m, sqrtf(key_dist(bk))) : m, true);
	}
	if (!(key_found(bk) && key_dist(bk) <= m * m))
		sweep(key_found(bk) ? fminf(r, sqrtf(key_dist(bk))) : r, false); // nothing within the first-probe radius: widen to the best distance, or to the rejection radius
#if MULLS_LDS_KCERT
	static_assert(GSH >= 0, "the k-candidate records are ranked by a compile-time number of lanes");
	co = lane_bests<(int)MULLS_LDS_GROUP, true>(lane_k, lane_s);
	if (sec == 0.0f)
		co.b2 = 0.0f; // the standing result lies outside the last (clipped) sweep: nothing is claimed about the other targets
#else
	(void)lane_k, (void)lane_s;
	co.cx = co.cy = 0xffffffffu, co.b2 = 0.0f;
#endif
}

// what the search of one class cloud needs to know about its iteration
struct ClassCtx
{
	float r, m;			 // rejection radius 2.5 * thr (filter_dis_times * dis_thre, cregistration.hpp:1745) and first-probe radius
	double max_dist_sqr; // (double)r squared: CorrespondenceEstimation's max_distance test
	bool gate, dedup;	 // >= 500 live source points: duplicate rule in force; ... and resolved in this workgroup's LDS table
	unsigned long long key_hi;
	uint4 *cand;	// candidate records of the k-candidate certificates (RunParams::cand; null: none are kept)
	uint32_t epoch; // ... and the epoch a record written in this iteration carries
};
__device__ __forceinline__ ClassCtx class_ctx(const RunParams &rp, const PairState &ps, const GridDesc &g, int cls, uint32_t alive_cur, bool called)
{
	ClassCtx C;
	// (wave-uniform values a VALU computed sit in VGPRs — two for the double — for the whole kernel; k_cert<512> has 64 and spilled this one: read back into SGPRs)
	C.r = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(2.5f * ps.thr[cls])));
	const double maxd = (double)C.r, md2 = maxd * maxd;
	C.max_dist_sqr = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(md2)), __builtin_amdgcn_readfirstlane(__double2loint(md2)));
	C.m = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(fminf(C.r, 0.999f * g.h - 2e-4f)))); // first-probe radius: its margin-inflated cube spans at most 3 cells per axis
	C.gate = alive_cur >= 500u;
	C.dedup = rp.lds_dedup != 0u && called && C.gate;
	C.key_hi = (unsigned long long)(0xffffffffu - (rp.tick_base + (uint32_t)ps.iter)) << 32;
	C.cand = rp.cand;
	C.epoch = rp.tick_base + (uint32_t)ps.iter;
	return C;
}

// The duplicate table of a class cloud in LDS: per target, the lowest source index matched to it this iteration (0xffff.. = none).  32-bit entries,
// or — W16, k_cert: source clouds of at most 65534 points, half the LDS, one more workgroup per CU — 16-bit entries packed in pairs and lowered by a
// compare-and-swap on the word that holds them (a target is rarely claimed twice: one read and one swap).
template <bool W16>
__device__ __forceinline__ void dedup_init(uint32_t *W, uint32_t tgt_n, uint32_t blk)
{
	const uint32_t words = W16 ? (tgt_n + 1u) >> 1 : tgt_n;
	for (uint32_t t = threadIdx.x; t < words; t += blk)
		W[t] = 0xffffffffu;
}
template <bool W16>
__device__ __forceinline__ void dedup_min(uint32_t *W, uint32_t t, uint32_t s)
{
	if (!W16)
	{
		atomicMin(&W[t], s);
		return;
	}
	uint32_t *w = W + (t >> 1);
	const uint32_t sh = (t & 1u) << 4;
	uint32_t old = *w;
	while (((old >> sh) & 0xffffu) > s)
	{
		const uint32_t prev = atomicCAS(w, old, (old & ~(0xffffu << sh)) | (s << sh));
		if (prev == old)
			break;
		old = prev;
	}
}
// whether source s holds target t
template <bool W16>
__device__ __forceinline__ bool dedup_holds(const uint32_t *W, uint32_t t, uint32_t s)
{
	return W16 ? ((W[t >> 1] >> ((t & 1u) << 4)) & 0xffffu) == s : W[t] == s;
}

// lane `sub == 0` of a sub-group commits the result of a searched query; returns whether it is a match
template <bool W16 = false>
__device__ __forceinline__ bool commit_search(const ClassCtx &C, const CloudDesc &d, uint32_t s, nnkey bk, float sec, float Rfin, uint32_t trips, const CandOut &co,
											   int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2, int2 *__restrict__ hint2, uint32_t *W,
											   unsigned long long *__restrict__ winner)
{
	const float best = key_dist(bk);
	const int bi = (int)(uint32_t)bk; // -1: nothing found
	const bool matched = bi >= 0 && !((double)best > C.max_dist_sqr);
	nn_idx[d.src_off + s] = matched ? bi : -1;
	nn_d2[d.src_off + s] = best;
	// hint and cost class of the next iteration; every target but the one found is at least min(second, radius swept) away
	const float lb_new = fminf(sqrtf(sec), Rfin);
	hint2[d.src_off + s] = make_int2((int32_t)(((uint32_t)bi & 0xffffu) | (min(trips, 31u) << 16)), __float_as_int(lb_new));
#if MULLS_LDS_KCERT
	if (C.cand) // the other lanes' nearest targets, and how much farther than lb_new everything outside {result, candidates} lies (co.b2 >= sec)
		C.cand[d.src_off + s] = make_uint4(co.cx, co.cy, __float_as_uint(fminf(sqrtf(co.b2), Rfin) - lb_new), C.epoch);
#else
	(void)co;
#endif
	if (matched)
	{
		if (C.dedup)
			dedup_min<W16>(W, (uint32_t)bi, s); // this workgroup sees every query of the class cloud: the duplicate table stays on chip
		else if (C.gate)
			atomicMin(&winner[d.tgt_off + bi], C.key_hi | (unsigned long long)s);
	}
	return matched;
}

// state bits of a point in the one-pass walk (cert_class_flat)
#define MULLS_FS_ALIVE 1u	 // = MULLS_F_ALIVE
#define MULLS_FS_VALID 2u	 // = MULLS_F_VALID (member of Corr_f before this iteration)
#define MULLS_FS_DIR_OK 4u	 // the direction check passes against the standing correspondence's target direction
#define MULLS_FS_STANDING 8u // the certified correspondence is the standing one (its record is in place)
#define MULLS_FS_SEARCH 16u	 // PARK: the point waits for a search; the correspondence's word holds its STANDING match meanwhile, the distance's word the sweep radius
// ... of a query the one-pass walk (cert_class_flat, PARK) searches itself: the result goes into the owner lane's two parked words in LDS — park[s] = state bits << 24 |
// (correspondence + 2) with the wait mark cleared and "it is the standing match" set from the match the word held, park[D0 + s] = the squared distance — instead of
// nn_idx / nn_d2 in memory; the hint record and the duplicate table as commit_search.  (rp.lds_dedup: a duplicate rule in force is this workgroup's table.)
template <bool W16, int D0>
__device__ __forceinline__ bool commit_search_park(const ClassCtx &C, const CloudDesc &d, uint32_t s, nnkey bk, float sec, float Rfin, uint32_t trips, const CandOut &co,
													int2 *__restrict__ hint2, uint32_t *W, uint32_t *park)
{
	const float best = key_dist(bk);
	const int bi = (int)(uint32_t)bk; // -1: nothing found
	const bool matched = bi >= 0 && !((double)best > C.max_dist_sqr);
	const float lb_new = fminf(sqrtf(sec), Rfin);
	hint2[d.src_off + s] = make_int2((int32_t)(((uint32_t)bi & 0xffffu) | (min(trips, 31u) << 16)), __float_as_int(lb_new));
#if MULLS_LDS_KCERT
	if (C.cand)
		C.cand[d.src_off + s] = make_uint4(co.cx, co.cy, __float_as_uint(fminf(sqrtf(co.b2), Rfin) - lb_new), C.epoch);
#else
	(void)co;
#endif
	const uint32_t w0 = park[s], st = (w0 >> 24) & ~MULLS_FS_SEARCH;
	const int32_t pm = (int32_t)(w0 & 0xffffffu) - 2, m = matched ? bi : -1;
	park[s] = ((st | ((matched && bi == pm) ? MULLS_FS_STANDING : 0u)) << 24) | (uint32_t)(m + 2);
	park[(uint32_t)D0 + s] = __float_as_uint(best);
	if (matched && C.dedup)
		dedup_min<W16>(W, (uint32_t)bi, s);
	return matched;
}

// End of a class cloud's iteration, by the workgroup that holds all of its correspondences (k_cert when it searched the few
// leftovers itself, else k_nn_lds): duplicate rule, then the rejection chain (k_filter's work) on the results while they are
// still in cache, and the class's counters.  `red`: 3 * (BLK / 64) words of LDS.  Chunk-level jobs (no rp.lds_dedup) only
// add their matches to the class counter; k_filter does the rest.
template <int BLK, bool W16 = false>
__device__ __forceinline__ void class_tail(const RunParams &rp, const PairState &ps, const ClassCtx &C, CloudDesc &d, const Job &job, uint32_t q_end,
											uint32_t matched_cnt, uint32_t searched, const uint32_t *W, uint32_t *red, const float4 *__restrict__ snrm,
											const float4 *__restrict__ tnrm, uint8_t *flag, int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2,
											int32_t *__restrict__ match, float *__restrict__ wd, const unsigned long long *__restrict__ winner,
											const float4 *__restrict__ tpos, float4 *__restrict__ mq)
{
	if (C.dedup)
	{
		// first source (lowest index) matched to a target keeps it (cregistration.hpp:1762-1789); the others become unmatched,
		// which is what k_filter does with them anyway (it skips its winner-table check when rp.lds_dedup is set)
		__threadfence_block();
		__syncthreads();
		for (uint32_t s = job.start + threadIdx.x; s < q_end; s += BLK)
			if (flag[d.src_off + s] & MULLS_F_ALIVE)
			{
				const int m = nn_idx[d.src_off + s];
				if (m >= 0 && !dedup_holds<W16>(W, (uint32_t)m, s))
					nn_idx[d.src_off + s] = -1;
			}
	}
	for (int off = 32; off > 0; off >>= 1)
		matched_cnt += __shfl_down(matched_cnt, off);
	if (!rp.lds_dedup)
	{
		if ((threadIdx.x & 63) == 0 && matched_cnt)
			atomicAdd(&d.n_matched, matched_cnt);
		return;
	}
	__threadfence_block(); // this workgroup's nn_idx / nn_d2 / snrm stores, read back below by other lanes
	__syncthreads();
	if ((threadIdx.x & 63) == 0)
		red[threadIdx.x >> 6] = matched_cnt;
	__syncthreads();
	uint32_t total_matched = 0;
	for (int w = 0; w < BLK / 64; w++)
		total_matched += red[w];
	const float thr = ps.thr[job.cls];
	// vertex correspondences skip the direction check (cregistration.hpp:1292)
	const FilterCtx F = {C.gate, total_matched > 0u, job.cls != 5, true, rp.rej_strict != 0, thr * thr, rp.cos_bearing, C.key_hi, rp.tgt_stage, rp.tgt_map};
	uint32_t n_alive = 0, n_valid = 0;
	for (uint32_t s = job.start + threadIdx.x; s < q_end; s += BLK)
		filter_point(F, d, s, snrm, tnrm, flag, nn_idx, nn_d2, match, wd, winner, tpos, mq, n_alive, n_valid);
	for (int off = 32; off > 0; off >>= 1)
	{
		n_alive += __shfl_down(n_alive, off);
		n_valid += __shfl_down(n_valid, off);
	}
	if ((threadIdx.x & 63) == 0)
	{
		red[BLK / 64 + (threadIdx.x >> 6)] = n_alive;
		red[2 * (BLK / 64) + (threadIdx.x >> 6)] = n_valid;
	}
	__syncthreads();
	if (threadIdx.x == 0)
	{
		uint32_t ta = 0, tv = 0;
		for (int w = 0; w < BLK / 64; w++)
		{
			ta += red[BLK / 64 + w];
			tv += red[2 * (BLK / 64) + w];
		}
		d.n_matched = total_matched; // k_finish reset it to 0 after the previous iteration
		d.alive_next = ta;
		d.valid_next = tv;
		d.n_search = searched;
	}
}
// class_tail for a WHOLE class cloud of at most TRIPS * BLK source slots (class-level job, rp.lds_dedup) by the workgroup that searched it: every
// record the duplicate rule and the rejection chain read — flag, correspondence, distance, standing match, the point's direction, the standing
// target direction — is requested at once, before the barrier that completes the duplicate table, instead of one after the other behind the
// tests that need them (k_nn_lds runs one workgroup per CU: each dependent round trip is exposed; the tail took as long as the search,
// profiles/r03_cert_phases_4096.txt).  Same decisions and outputs as class_tail (losers of the duplicate rule keep their nn_idx in memory: nothing
// reads it after this).
template <int BLK, int TRIPS, bool W16>
__device__ __forceinline__ void class_tail_flat(const RunParams &rp, const PairState &ps, const ClassCtx &C, CloudDesc &d, const Job &job, uint32_t q_end,
												 uint32_t matched_cnt, uint32_t searched, const uint32_t *W, uint32_t *red, const float4 *__restrict__ snrm,
												 const float4 *__restrict__ tnrm, uint8_t *flag, const int32_t *__restrict__ nn_idx, const float *__restrict__ nn_d2,
												 int32_t *__restrict__ match, float *__restrict__ wd, const float4 *__restrict__ tpos, float4 *__restrict__ mq)
{
	__threadfence_block(); // the searched points' nn_idx / nn_d2 (commit_search, other lanes)
	__syncthreads();	   // ... and the duplicate table is complete
	uint32_t FL[TRIPS];
	int32_t M[TRIPS], PM[TRIPS];
	float D[TRIPS];
	float3 N1[TRIPS], T2[TRIPS];
	const uint32_t last = d.src_off + (q_end ? q_end - 1u : 0u);
#pragma unroll
	for (int k = 0; k < TRIPS; k++)
	{
		const uint32_t gi = min(d.src_off + threadIdx.x + (uint32_t)k * BLK, last);
		FL[k] = flag[gi];
		M[k] = nn_idx[gi];
		D[k] = nn_d2[gi];
		PM[k] = match[gi];
		N1[k] = *reinterpret_cast<const float3 *>(snrm + gi);
		T2[k] = *reinterpret_cast<const float3 *>(mq + 2u * gi + 1u);
	}
	for (int off = 32; off > 0; off >>= 1)
		matched_cnt += __shfl_down(matched_cnt, off);
	if ((threadIdx.x & 63) == 0)
		red[threadIdx.x >> 6] = matched_cnt;
	__syncthreads();
	uint32_t total_matched = 0;
	for (int w = 0; w < BLK / 64; w++)
		total_matched += red[w];
	const float thr = ps.thr[job.cls], max_sqr = thr * thr; // CorrespondenceRejectorDistance::setMaximumDistance (float)
	const bool any_match = total_matched > 0u, normal_check = job.cls != 5, strict = rp.rej_strict != 0; // vertex correspondences skip the direction check (:1292)
	uint32_t n_alive = 0, n_valid = 0;
#pragma unroll
	for (int k = 0; k < TRIPS; k++)
	{
		const uint32_t s = threadIdx.x + (uint32_t)k * BLK, gi = d.src_off + s;
		if (s >= q_end || !(FL[k] & MULLS_F_ALIVE))
			continue;
		int32_t m = M[k];
		// first source (lowest index) matched to a target keeps it (cregistration.hpp:1762-1789); the others become unmatched
		if (C.dedup && m >= 0 && !dedup_holds<W16>(W, (uint32_t)m, s))
			m = -1;
		bool alive = true, valid;
		float3 n2 = T2[k]; // the standing correspondence's target direction
		if (any_match)
		{
			valid = m >= 0;
			if (C.gate && m < 0) // (rp.lds_dedup: the duplicate rule was resolved above)
			{
				alive = false; // unmatched or duplicate-losing source points vanish for good (:1762-1789)
				valid = false;
			}
			if (valid)
			{
				const float dist = D[k];
				valid = strict ? dist < max_sqr : !(dist > max_sqr); // CorrespondenceRejectorDistance (see mulls_params.rejector_strict)
				if (valid)
				{
					wd[gi] = dist; // pcl::Correspondence::distance (shares storage with ::weight)
					if (PM[k] != m)
					{
						match[gi] = m;
						float4 q2, n4;
						tgt_record(rp.tgt_stage, rp.tgt_map, d, (uint32_t)m, tpos, tnrm, q2, n4);
						mq[2u * gi] = q2;
						mq[2u * gi + 1u] = n4;
						n2 = make_float3(n4.x, n4.y, n4.z);
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
			valid = (FL[k] & MULLS_F_VALID) != 0; // the previous Corr_f is still in place (SURVEY B-4) and goes through the direction check again
		if (valid && normal_check)
		{
			const double dot = (double)N1[k].x * (double)n2.x + (double)N1[k].y * (double)n2.y + (double)N1[k].z * (double)n2.z;
			const float c = (float)fabs(dot);
			if ((double)c < rp.cos_bearing)
				valid = false;
		}
		const uint32_t nf = (alive ? MULLS_F_ALIVE : 0u) | (valid ? MULLS_F_VALID : 0u);
		if (nf != FL[k])
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
		red[BLK / 64 + (threadIdx.x >> 6)] = n_alive;
		red[2 * (BLK / 64) + (threadIdx.x >> 6)] = n_valid;
	}
	__syncthreads();
	if (threadIdx.x == 0)
	{
		uint32_t ta = 0, tv = 0;
		for (int w = 0; w < BLK / 64; w++)
		{
			ta += red[BLK / 64 + w];
			tv += red[2 * (BLK / 64) + w];
		}
		d.n_matched = total_matched; // k_finish reset it to 0 after the previous iteration
		d.alive_next = ta;
		d.valid_next = tv;
		d.n_search = searched;
	}
}

// LDS of the light pass next to the duplicate table: the leftover queries a workgroup searches itself, reduction scratch
template <int SMALL>
struct CertLds
{
	float4 uq[SMALL];
	uint32_t us[SMALL];
	static constexpr bool LOOK = MULLS_LDS_KCERT != 0 && SMALL >= 128; // lists of a few dozen entries (the resident loop, the small batches' one-launch search) never get the look: no room kept
	float um[LOOK ? SMALL : 1];				   // how far this iteration's step moved the listed point (entries flagged MULLS_US_KCERT: the k-candidate certificate's look needs it)
	uint32_t ucount, red[3 * 16];
};

// nn_idx value of a live point that cert_class could not certify and left to lds_search_class (its sweep radius waits in nn_d2)
#define MULLS_NEEDS_SEARCH (-2)

// The k-candidate certificate of one point (px, py, pz: its position after this iteration's rigid step, `moved`: how far the step took it) whose
// hinted target alone did not certify.  h = its (hint, bound) record, cr = its candidate record, tp(i) = position of target i.  The hinted target and
// the candidates are evaluated with the search's own distance expression and compared as keys (lowest index on ties): the nearest of them IS the
// nearest target if it beats, by the plain certificate's margins, the bound on everything outside the set — (bound on the non-hinted targets) + cr.z
// at the previous position, hence `moved` less here.  Exactness: DESIGN.md section 14.1 (the triangle inequality of section 4 on a larger exception set).
// Returns whether it certifies; then bk = (squared distance, index) of the nearest target, lb_next = the bound on every OTHER target from here (what a
// search would leave as min(sqrt(sec), Rfin)), cr_next = the record with the old hint in the new one's slot.  pre(i) / pos(token): the two dependent
// loads of a target's position (the crop map's entry, then the staged record), so that all candidates' loads of a stage are in flight together.
template <bool W16, class TPre, class TPos>
__device__ __forceinline__ bool kcert_point(const RunParams &rp, const uint4 cr, const int2 h, uint32_t iter, float px, float py, float pz, float moved, uint32_t tgt_n,
											 const TPre &pre, const TPos &pos, nnkey &bk, float &lb_next, uint4 &cr_next)
{
	constexpr int NO = W16 ? 4 : 2;
	const uint32_t hj = W16 ? ((uint32_t)h.x & 0xffffu) : (uint32_t)h.x; // (hj < tgt_n: the caller's condition)
	if (!(rp.kcert != 0u && cr.w - rp.tick_base <= iter)) // not written by a search of THIS run (unsigned: an older or a zeroed record wraps far above iter)
		return false;
	// every target outside the set kept at least this distance before the step.  The bound travels as the sum of two floats (h.y is lowered by every step's move and
	// may have gone negative while cr.z stayed large): the sum's rounding error is absolute — up to an ulp of the larger term — which the relative 1e-5 margins
	// below do not cover when the sum is small; four such ulps are taken off (advisor, round 5)
	const float hy = __int_as_float(h.y), crz = __uint_as_float(cr.z);
	const float Bp = (hy + crz) - 4.0f * 1.1920929e-7f * fmaxf(fabsf(hy), fabsf(crz));
	if (!(moved * 1.00001f < Bp * 0.99999f))					   // hopeless whatever the candidates' distances are: no gathers
		return false;
	// every load of a stage is issued before the first is used, at indices that are valid whatever the record holds (an absent candidate re-reads the hint)
	uint32_t ci[NO + 1], tok[NO + 1];
	bool ok[NO + 1];
	ci[0] = hj, ok[0] = true;
#pragma unroll
	for (int j = 0; j < NO; j++)
	{
		const uint32_t w = j < NO / 2 ? cr.x : cr.y, c = W16 ? (w >> ((j & 1) << 4)) & 0xffffu : w;
		ok[j + 1] = c < tgt_n;
		ci[j + 1] = ok[j + 1] ? c : hj;
	}
#pragma unroll
	for (int j = 0; j <= NO; j++)
		tok[j] = pre(ci[j]);
	float4 cp[NO + 1];
#pragma unroll
	for (int j = 0; j <= NO; j++)
		cp[j] = pos(tok[j]);
	bk = NNKEY_NONE;
	float sec = __builtin_inff();
	int slot = -1;
#pragma unroll
	for (int j = 0; j <= NO; j++)
	{
		const float dx = px - cp[j].x, dy = py - cp[j].y, dz = pz - cp[j].z;
		const float dj = ok[j] ? (dx * dx + dy * dy) + dz * dz : __builtin_inff(); // L2_Simple<float>, no FMA: the expression of lds_scan_box / take_one
		sec = __builtin_amdgcn_fmed3f(sec, dj, key_dist(bk));
		const nnkey kj = ok[j] ? nn_key(dj, ci[j]) : NNKEY_NONE;
		const bool take = kj < bk;
		bk = take ? kj : bk;
		slot = take ? j - 1 : slot;
	}
	const float best = key_dist(bk);
	if (!(best >= 0.0f) || !(sqrtf(best) * 1.00001f + moved * 1.00001f < Bp * 0.99999f)) // NaN anywhere fails the test
		return false;
	const float Bn = Bp - moved * 1.00001f;
	lb_next = fminf(sqrtf(sec), Bn);
	// the old hint takes the new one's place among the candidates
	if (W16)
	{
		const uint32_t sh = ((uint32_t)slot & 1u) << 4, keepm = ~(0xffffu << sh), put = hj << sh;
		cr_next.x = slot >= 0 && slot < NO / 2 ? (cr.x & keepm) | put : cr.x;
		cr_next.y = slot >= NO / 2 ? (cr.y & keepm) | put : cr.y;
	}
	else
	{
		cr_next.x = slot == 0 ? hj : cr.x;
		cr_next.y = slot == 1 ? hj : cr.y;
	}
	cr_next.z = __float_as_uint(Bn - lb_next);
	cr_next.w = cr.w;
	return true;
}

// Second chance of a light pass's leftover list (the k-candidate certificates): one lane per listed point whose slot word carries MULLS_US_KCERT (a hinted
// point) loads the point's records, gathers the candidates and, if they certify it, puts the results where a search would put them — nn_idx, nn_d2, the hint
// and candidate records, the duplicate table.  The list (U <= SMALL entries) is compacted in place; returns the number of points left to search.
// Every lane of the workgroup calls it; barriers inside.
#define MULLS_US_KCERT 0x80000000u
template <int BLK, bool W16, int SMALL>
__device__ __forceinline__ uint32_t kcert_list(CertLds<SMALL> &CL, const RunParams &rp, const PairState &ps, const ClassCtx &C, const CloudDesc &d, uint32_t U,
												int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2, int2 *__restrict__ hint2, uint32_t *W,
												unsigned long long *__restrict__ winner, const float4 *__restrict__ tpos, uint32_t &matched_cnt)
{
	static_assert(!CertLds<SMALL>::LOOK || SMALL <= BLK, "one lane per listed point (the look is compiled out of the lists that are longer)");
	const uint32_t UL = U; // (<= SMALL: the caller's condition)
	float4 e = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	uint32_t es = 0u;
	bool keep = false;
	if (threadIdx.x < UL)
	{
		e = CL.uq[threadIdx.x];
		es = CL.us[threadIdx.x];
		keep = true;
		if (es & MULLS_US_KCERT)
		{
			es &= ~MULLS_US_KCERT;
			const uint32_t gi = d.src_off + es;
			const uint4 cr = C.cand[gi];
			const int2 h = hint2[gi];
			nnkey bk;
			float lb_next;
			uint4 cr_next;
			const float moved = CL.um[threadIdx.x];
			// (the descriptor's fields in registers: read through `d` inside the accessors they are re-loaded, with a wait, before every gather)
			const uint32_t tgt_off = d.tgt_off, tgt_n = d.tgt_n, st_off = d.tgt_stage, st_mul = ((d.stage_fmt >> 2) & 3u) == MULLS_STAGE_AOS48 ? 3u : 1u;
			const uint16_t *__restrict__ tmap = rp.tgt_map;
			const float4 *__restrict__ stage = rp.tgt_stage;
			bool pass;
			if (tmap) // (uniform) no cropped copy of the target: the crop map's entry, then the staged record (tgt_point)
			{
				auto pre = [&](uint32_t i) -> uint32_t { return (uint32_t)tmap[tgt_off + i]; };
				auto pos = [&](uint32_t t) -> float4 { return stage[(size_t)st_off + (size_t)t * st_mul]; };
				pass = kcert_point<true>(rp, cr, h, (uint32_t)ps.iter, e.x, e.y, e.z, moved, tgt_n, pre, pos, bk, lb_next, cr_next);
			}
			else
			{
				auto pre = [&](uint32_t i) -> uint32_t { return i; };
				auto pos = [&](uint32_t t) -> float4 { return tpos[tgt_off + t]; };
				pass = kcert_point<true>(rp, cr, h, (uint32_t)ps.iter, e.x, e.y, e.z, moved, tgt_n, pre, pos, bk, lb_next, cr_next);
			}
			if (pass)
			{
				const float best = key_dist(bk);
				const uint32_t bi = (uint32_t)bk;
				const bool matched = !((double)best > C.max_dist_sqr);
				nn_idx[gi] = matched ? (int32_t)bi : -1;
				nn_d2[gi] = best;
				hint2[gi] = make_int2((int32_t)bi, __float_as_int(lb_next)); // cost class 0
				C.cand[gi] = cr_next;
				if (matched)
				{
					matched_cnt++;
					if (C.dedup)
						dedup_min<W16>(W, bi, es);
					else if (C.gate)
						atomicMin(&winner[d.tgt_off + bi], C.key_hi | (unsigned long long)es);
				}
				keep = false;
			}
		}
	}
	const unsigned long long bal = __ballot(keep);
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	if (rp.dbg_ticks) // diagnostics (MULLS_OPT_DEBUG_STOP = 20): points that got the second chance, points it certified
	{
		const unsigned long long tried = __ballot(threadIdx.x < UL && (CL.us[min(threadIdx.x, (uint32_t)SMALL - 1u)] & MULLS_US_KCERT) != 0u);
		const unsigned long long passed = __ballot(threadIdx.x < UL && !keep);
		if (lane == 0u && tried)
		{
			atomicAdd(&rp.dbg_ticks[14], (unsigned long long)__popcll(tried));
			atomicAdd(&rp.dbg_ticks[15], (unsigned long long)__popcll(passed));
		}
	}
	if (lane == 0u)
		CL.red[wave] = (uint32_t)__popcll(bal);
	__syncthreads(); // every listed entry has been read
	uint32_t base = 0u, total = 0u;
	for (uint32_t w = 0; w < (uint32_t)(BLK / 64); w++)
	{
		const uint32_t c = CL.red[w];
		base += w < wave ? c : 0u;
		total += c;
	}
	if (keep)
	{
		const uint32_t k = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
		CL.uq[k] = e;
		CL.us[k] = es;
	}
	__syncthreads();
	return total;
}

// Light pass of one class cloud's iteration, BLK lanes, one source point per lane and trip (see the tier's description above):
// rigid step, certificates, and — when at most MULLS_CERT_SMALL points are left over — their search against the grid in global
// memory, the duplicate rule and the rejection chain.  Returns true when the class cloud is done for this iteration, false when
// the caller has to stage the target cloud (lds_search_class).  W: LDS, tgt_n words (only touched with rp.lds_dedup).  Every lane
// of the workgroup must call it (barriers inside).
template <int BLK, bool W16 = false, int SMALL = MULLS_CERT_SMALL>
__device__ __forceinline__ bool cert_class(CertLds<SMALL> &CL, const RunParams &rp, const PairState &ps, const Job &job, CloudDesc &d, const GridDesc &g, uint32_t *W,
											float4 *__restrict__ spos, float4 *__restrict__ snrm, const uint32_t *__restrict__ cell_start,
											const float4 *__restrict__ tsorted, uint8_t *flag, int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2,
											unsigned long long *__restrict__ winner, const float4 *__restrict__ tnrm, int32_t *__restrict__ match,
											float *__restrict__ wd, const float4 *__restrict__ tpos, int32_t *__restrict__ nn_hint, float4 *__restrict__ mq)
{
	float4 *uq = CL.uq; // the few queries this workgroup searches itself
	uint32_t *us = CL.us, *red = CL.red;
	uint32_t &ucount = CL.ucount;
	int2 *__restrict__ hint2 = reinterpret_cast<int2 *>(nn_hint); // per source point: (hint word, bound on every OTHER target's distance)

	const uint32_t src_n = d.src_n, tgt_n = d.tgt_n;
	const bool called = class_called(rp, d, job.cls);
	const ClassCtx C = class_ctx(rp, ps, g, job.cls, d.alive_cur, called);
	const bool have_prev = ps.iter > 0; // hint records of this run exist from its second iteration on
	const bool use_hint = called && have_prev;
	const bool kc = CertLds<SMALL>::LOOK && rp.kcert != 0u && use_hint && C.cand != nullptr; // (uniform)
	const uint32_t q_end = min(src_n, job.start + (job.count ? job.count : (uint32_t)MULLS_SRC_PER_BLOCK));
	if (C.dedup)
		dedup_init<W16>(W, tgt_n, BLK);
	if (threadIdx.x == 0)
		ucount = 0u;
	__syncthreads();

	uint32_t matched_cnt = 0;
	for (uint32_t s = job.start + threadIdx.x; s < q_end; s += BLK)
	{
		const uint32_t gi = d.src_off + s;
		// every record of the point is requested at once, before the flag is looked at (position, direction, hint, standing match and its target
		// position): one memory round trip instead of a chain of four
		const uint32_t fl = flag[gi];
		const float4 p = spos[gi], n = snrm[gi];
		const int2 h = hint2[gi];
		const int32_t pm0 = match[gi];
		const float4 q0 = mq[2u * gi];
		if (!(fl & MULLS_F_ALIVE))
			continue;
		uint32_t hv = 0xffffu;
		float lb = 0.0f;
		int32_t pm = -1;
		if (have_prev)
		{
			lb = __int_as_float(h.y);
			if (use_hint)
			{
				hv = (uint32_t)h.x;
				pm = pm0;
			}
		}
		// the hinted target's position: for a point whose hint is its standing correspondence it sits in the point's own
		// record (coalesced), otherwise it is gathered
		const uint32_t hj = hv & 0xffffu;
		float4 tj = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		if (hj < tgt_n)
			tj = (int32_t)hj == pm ? q0 : tgt_point(rp.tgt_stage, rp.tgt_map, d, hj, tpos);
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
		spos[gi] = make_float4(out.x, out.y, out.z, p.w);
		snrm[gi] = make_float4(onx, ony, onz, n.w);
		// How far this step moved the point (float positions before and after: both exact, the arithmetic below carries a few
		// ulp).  Every bound on "the distance to any target other than the hinted one" shrinks by exactly that much.
		const float mx = out.x - p.x, my = out.y - p.y, mz = out.z - p.z;
		const float moved = sqrtf((mx * mx + my * my) + mz * mz);
		const float lb_next = lb - moved * 1.00001f;
		if (!called)
		{
			// the points still move: keep the bounds of a class that sits this iteration out valid (none exist at iteration 0)
			nn_hint[2u * gi + 1u] = __float_as_int(have_prev ? lb_next : 0.0f);
			continue;
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
				// a searched query sweeps a little farther than the hinted target: what lies beyond the sweep is what bounds the
				// next iterations' certificates, and the steps shrink as the registration converges
				out.w = dh + fminf(fmaxf(rp.cert_slack_rate * moved, rp.cert_slack_min), rp.cert_slack_max);
				if (certified)
				{
					const bool matched = !((double)d0 > C.max_dist_sqr);
					nn_idx[gi] = matched ? (int32_t)hj : -1;
					nn_d2[gi] = d0;
					hint2[gi] = make_int2((int32_t)hj, __float_as_int(lb_next)); // cost class 0
					if (matched)
					{
						matched_cnt++;
						if (C.dedup)
							dedup_min<W16>(W, hj, s);
						else if (C.gate)
							atomicMin(&winner[d.tgt_off + hj], C.key_hi | (unsigned long long)s);
					}
				}
			}
		}
		if (!certified)
		{
			nn_idx[gi] = MULLS_NEEDS_SEARCH;
			nn_d2[gi] = out.w;
			const uint32_t k = atomicAdd(&ucount, 1u);
			if (k < (uint32_t)SMALL)
			{
				const bool second = kc && hj < tgt_n; // a hinted point: the k-candidate certificate gets a look before the search (kcert_list)
				uq[k] = out;
				us[k] = second ? s | MULLS_US_KCERT : s;
				if (CertLds<SMALL>::LOOK)
					CL.um[k] = moved;
			}
		}
	}
	if (!called)
		return true;
	__syncthreads();
	uint32_t U = ucount;
	if (kc && U >= rp.kcert_min && U <= (uint32_t)SMALL) // (a list that overflowed goes to the heavy pass, whose searches cost a tenth of a look)
		U = kcert_list<BLK, W16, SMALL>(CL, rp, ps, C, d, U, nn_idx, nn_d2, hint2, W, winner, tpos, matched_cnt);
	if (U > (uint32_t)SMALL) // (a compile-time bound: the same test against a run-time option cost configs[0] 9 % of its search time, profiles/r05_experiments.txt item 5)
		return false; // too many for the global-memory walk: the caller has the target cloud staged (lds_search_class), which counts the
					  // matches certified here again from nn_idx; chunk-level jobs add theirs to the class counter there too
	// the few leftovers against the grid where k_grid_build_sort left it (L2-resident): same sweeps, same keys
	const GlobGrid L = {tsorted + d.tgt_off, reinterpret_cast<const uint16_t *>(cell_start) + g.cell_off};
	const uint32_t sub = threadIdx.x & (MULLS_LDS_GROUP - 1u), grp = threadIdx.x / MULLS_LDS_GROUP;
	for (uint32_t i = grp; i < U; i += BLK / MULLS_LDS_GROUP)
	{
		nnkey bk;
		float sec, Rfin;
		uint32_t trips;
		CandOut co = {0xffffffffu, 0xffffffffu, 0.0f};
		search_query<MULLS_GLOB_CHUNK>(g, L, uq[i], C.r, C.m, sub, bk, sec, Rfin, trips, co);
		if (sub == 0 && commit_search<W16>(C, d, us[i] & ~MULLS_US_KCERT, bk, sec, Rfin, trips, co, nn_idx, nn_d2, hint2, W, winner))
			matched_cnt++;
	}
	class_tail<BLK, W16>(rp, ps, C, d, job, q_end, matched_cnt, U, W, red, snrm, tnrm, flag, nn_idx, nn_d2, match, wd, winner, tpos, mq);
	return true;
}

// cert_class for a WHOLE class cloud of at most TRIPS * BLK source slots held by one workgroup (class-level job, rp.lds_dedup): one pass.
// cert_class + class_tail walk the cloud three times (certificates; duplicate rule; rejection chain) and hand everything from one walk to the
// next through memory — nn_idx, nn_d2, the flags, the standing match, the transformed direction — which the second and third walk read back
// from beyond the L2 (a launch has 20 MB of class clouds in flight per XCD): 126 B fetched per source slot where 77 are owed (rocprofv3
// FETCH_SIZE, profiles/r02_zzz_pmc_traffic.txt).  Here a lane keeps three words per point across the workgroup barriers — state bits, the
// correspondence, its squared distance — and stores only what a later kernel reads: positions, directions, hints, flags,
// pcl::Correspondence::distance, the (match, record) of a changed correspondence.  The direction check against the STANDING correspondence's
// target direction is evaluated right after the rigid step, while both directions are in registers (one bit kept); a changed correspondence
// fetches its new record and the point's direction again.  nn_idx / nn_d2 are written for the points a search has to see — the leftovers, or
// every live point when the cloud goes to k_nn_lds (return false).  The loads of two trips are in flight at a time.  The kernel is bound by the
// latency of its memory round trips at the occupancy its registers and LDS allow (profiles/r03_k_cert_occupancy.txt): both are kept small.
// The same arithmetic, decisions and outputs as cert_class + class_tail; `called` class clouds only (class_called: the caller checks).
template <int BLK, int TRIPS, bool W16, int SMALL, bool PARK = false>
__device__ __forceinline__ bool cert_class_flat(CertLds<SMALL> &CL, const RunParams &rp, const PairState &ps, const Job &job, CloudDesc &d, const GridDesc &g, uint32_t *W,
												 float4 *__restrict__ spos, float4 *__restrict__ snrm, const uint32_t *__restrict__ cell_start,
												 const float4 *__restrict__ tsorted, uint8_t *flag, int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2,
												 unsigned long long *__restrict__ winner, const float4 *__restrict__ tnrm, int32_t *__restrict__ match,
												 float *__restrict__ wd, const float4 *__restrict__ tpos, int32_t *__restrict__ nn_hint, float4 *__restrict__ mq, uint32_t *park = nullptr)
{
	float4 *uq = CL.uq; // the few queries this workgroup searches itself
	uint32_t *us = CL.us, *red = CL.red;
	uint32_t &ucount = CL.ucount;
	int2 *__restrict__ hint2 = reinterpret_cast<int2 *>(nn_hint);
	const uint32_t src_n = d.src_n, tgt_n = d.tgt_n;
	const ClassCtx C = class_ctx(rp, ps, g, job.cls, d.alive_cur, true);
	const bool have_prev = ps.iter > 0; // hint records of this run exist from its second iteration on
	const bool normal_check = job.cls != 5; // vertex correspondences skip the direction check (:1292)
	const bool kc = CertLds<SMALL>::LOOK && rp.kcert != 0u && have_prev && C.cand != nullptr; // (uniform)
	const uint32_t q_end = src_n;		// class-level job: job.start == 0, job.count >= src_n
	unsigned long long t_prev = rp.dbg_ticks ? wall_clock64() : 0ull;
	// (t_prev is updated by every lane: a value only lane 0 writes is divergent, lives in two VGPRs for the whole kernel and was spilled to scratch in k_cert<512>)
#define FLAT_TICK(k)                                                  \
	if (rp.dbg_ticks)                                                 \
	{                                                                 \
		const unsigned long long now_ = wall_clock64();               \
		if (threadIdx.x == 0)                                         \
			atomicAdd(&rp.dbg_ticks[k], now_ - t_prev);               \
		t_prev = now_;                                                \
	}

	// the records of one trip's point, requested together (slots beyond the cloud re-read its last point: no branch between the loads)
	struct Rec
	{
		uint32_t fl;
		int32_t pm;
		float3 p, n, q, t; // x y z of position, direction, standing target position and direction: the fourth words (intensity, curvature) stay where they are
		int2 h;
	};
	const uint32_t last = d.src_off + (q_end ? q_end - 1u : 0u);
	auto load = [&](int k) {
		Rec r;
		const uint32_t gi = min(d.src_off + threadIdx.x + (uint32_t)k * BLK, last);
		r.fl = flag[gi];
		r.p = *reinterpret_cast<const float3 *>(spos + gi), r.n = *reinterpret_cast<const float3 *>(snrm + gi);
		r.h = hint2[gi];
		r.pm = match[gi];
		r.q = *reinterpret_cast<const float3 *>(mq + 2u * gi), r.t = *reinterpret_cast<const float3 *>(mq + 2u * gi + 1u);
		return r;
	};
	// per point and trip across the workgroup barriers: state bits, the correspondence, its squared distance.  PARK: the last two wait in LDS (`park`,
	// 2 * TRIPS * BLK words; the state bits ride in the correspondence's word) instead of registers — k_cert<512> holds 64 registers per lane (four workgroups per CU) and spilled twelve of them to scratch
	// PARK: word 0 = state bits << 24 | (correspondence + 2), word 1 = the distance's bits
	uint32_t STr[PARK ? 1 : TRIPS];
	int32_t Mr[PARK ? 1 : TRIPS];
	float D0r[PARK ? 1 : TRIPS];
	auto setSM = [&](int k, uint32_t st, int32_t m) {
		if (PARK)
			park[(uint32_t)k * BLK + threadIdx.x] = (st << 24) | (uint32_t)(m + 2);
		else
			STr[PARK ? 0 : k] = st, Mr[PARK ? 0 : k] = m;
	};
	auto getST = [&](int k) -> uint32_t { return PARK ? park[(uint32_t)k * BLK + threadIdx.x] >> 24 : STr[PARK ? 0 : k]; };
	auto getM = [&](int k) -> int32_t { return PARK ? (int32_t)(park[(uint32_t)k * BLK + threadIdx.x] & 0xffffffu) - 2 : Mr[PARK ? 0 : k]; };
	auto setD0 = [&](int k, float v) {
		if (PARK)
			park[(uint32_t)(TRIPS + k) * BLK + threadIdx.x] = __float_as_uint(v);
		else
			D0r[PARK ? 0 : k] = v;
	};
	auto getD0 = [&](int k) -> float { return PARK ? __uint_as_float(park[(uint32_t)(TRIPS + k) * BLK + threadIdx.x]) : D0r[PARK ? 0 : k]; };
	// PARK: a point that waits for a search is marked by a state bit, and the sub-group that searches it here puts the result into the point's two words
	// (commit_search_park: the owner's tail reads LDS instead of waiting for nn_idx / nn_d2 / match to come back from memory — one round trip less in the chain
	// search -> tail, which is what a workgroup with leftovers spends its time on); nn_idx / nn_d2 are written only when the class cloud goes to k_nn_lds
	auto waits = [&](int k) -> bool { return PARK ? (getST(k) & MULLS_FS_SEARCH) != 0u : getM(k) == MULLS_NEEDS_SEARCH; };
	uint32_t matched_cnt = 0;
	// rigid step + certificate of one trip's point (cert_class's arithmetic)
	auto cert = [&](int k, const Rec &r) {
		const uint32_t s = threadIdx.x + (uint32_t)k * BLK, gi = d.src_off + s;
		uint32_t st_k = 0u;
		int32_t m_k = -1;
		setSM(k, 0u, -1), setD0(k, 0.0f);
		if (s >= q_end || !(r.fl & MULLS_F_ALIVE))
			return;
		st_k = r.fl & (MULLS_FS_ALIVE | MULLS_FS_VALID);
		const float3 p = r.p, n = r.n;
		uint32_t hv = 0xffffu;
		float lb = 0.0f;
		int32_t pm = -1;
		if (have_prev)
		{
			lb = __int_as_float(r.h.y);
			hv = (uint32_t)r.h.x;
			pm = r.pm;
		}
		const uint32_t hj = hv & 0xffffu;
		float4 tj = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		if (hj < tgt_n)
			tj = (int32_t)hj == pm ? make_float4(r.q.x, r.q.y, r.q.z, 0.0f) : tgt_point(rp.tgt_stage, rp.tgt_map, d, hj, tpos);
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
		{
			// the direction check of the rejection chain (:1818-1826) against the standing correspondence's target direction
			const double dot = (double)onx * (double)r.t.x + (double)ony * (double)r.t.y + (double)onz * (double)r.t.z;
			const float c = (float)fabs(dot);
			if (!((double)c < rp.cos_bearing))
				st_k |= MULLS_FS_DIR_OK;
		}
		const float mx = out.x - p.x, my = out.y - p.y, mz = out.z - p.z;
		const float moved = sqrtf((mx * mx + my * my) + mz * mz);
		const float lb_next = lb - moved * 1.00001f;
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
					m_k = matched ? (int32_t)hj : -1;
					setD0(k, d0);
					hint2[gi] = make_int2((int32_t)hj, __float_as_int(lb_next)); // cost class 0
					if (matched)
					{
						matched_cnt++;
						if ((int32_t)hj == r.pm)
							st_k |= MULLS_FS_STANDING;
						if (C.dedup) // (rp.lds_dedup: the duplicate rule in force means the table is this workgroup's)
							dedup_min<W16>(W, hj, s);
					}
				}
			}
		}
		if (!certified)
		{
			setD0(k, out.w);
			if (PARK)
			{
				st_k |= MULLS_FS_SEARCH;
				m_k = r.pm; // (>= -1: the setup's -1 or a target index)
			}
			else
			{
				m_k = MULLS_NEEDS_SEARCH;
				nn_idx[gi] = MULLS_NEEDS_SEARCH;
				nn_d2[gi] = out.w;
			}
			const uint32_t u = atomicAdd(&ucount, 1u);
			if (u < (uint32_t)SMALL)
			{
				const bool second = kc && hj < tgt_n; // a hinted point: the k-candidate certificate gets a look before the search (kcert_list)
				uq[u] = out;
				us[u] = second ? s | MULLS_US_KCERT : s;
				if (CertLds<SMALL>::LOOK)
					CL.um[u] = moved;
			}
		}
		setSM(k, st_k, m_k);
	};

	static_assert(TRIPS == 2 || TRIPS == 3, "two trips' loads in flight");
	Rec r0 = load(0), r1 = load(1);
	if (C.dedup)
		dedup_init<W16>(W, tgt_n, BLK);
	if (threadIdx.x == 0)
		ucount = 0u;
	__syncthreads(); // the duplicate table is armed
	FLAT_TICK(0)
	{
		// (a cloud of at most two trips — the ground and pillar clouds of a down-sampled scan — does not wait for a third trip's loads; uniform)
		const bool third = TRIPS == 3 && q_end > 2u * (uint32_t)BLK;
		cert(0, r0);
		if (third)
		{
			r0 = load(2);
			cert(1, r1);
			cert(2, r0);
		}
		else
		{
			cert(1, r1);
			if (TRIPS == 3)
				setSM(2, 0u, -1), setD0(2, 0.0f);
		}
	}
	__syncthreads();
	FLAT_TICK(2)
	uint32_t U = ucount;
	const uint32_t U0 = U;
	if (kc && U >= rp.kcert_min && U <= (uint32_t)SMALL) // (a list that overflowed goes to the heavy pass, whose searches cost a tenth of a look)
		U = kcert_list<BLK, W16, SMALL>(CL, rp, ps, C, d, U, nn_idx, nn_d2, hint2, W, winner, tpos, matched_cnt);
	FLAT_TICK(1)
	if (rp.dbg_ticks && threadIdx.x == 0)
	{
		atomicAdd(&rp.dbg_ticks[13], (unsigned long long)U0); // points the plain certificate left over ...
		atomicAdd(&rp.dbg_ticks[7], (unsigned long long)U);	  // ... and what is searched (here or, beyond SMALL, by the heavy pass)
	}
	if (U > (uint32_t)SMALL) // (a compile-time bound: the same test against a run-time option cost configs[0] 9 % of its search time, profiles/r05_experiments.txt item 5)
	{
		// too many for the global-memory walk: k_nn_lds stages the target cloud (lds_search_class) and reads every live point's result from memory
#pragma unroll
		for (int k = 0; k < TRIPS; k++)
			if ((getST(k) & MULLS_FS_ALIVE) && (PARK || !waits(k)))
			{
				const uint32_t gi = d.src_off + threadIdx.x + (uint32_t)k * BLK;
				nn_idx[gi] = waits(k) ? MULLS_NEEDS_SEARCH : getM(k);
				nn_d2[gi] = getD0(k); // (a waiting point: its sweep radius)
			}
		return false;
	}
	// the few leftovers against the grid where the setup left it (L2-resident): same sweeps, same keys
	if (U)
	{
		const GlobGrid L = {tsorted + d.tgt_off, reinterpret_cast<const uint16_t *>(cell_start) + g.cell_off};
		const uint32_t sub = threadIdx.x & (MULLS_LDS_GROUP - 1u), grp = threadIdx.x / MULLS_LDS_GROUP;
		for (uint32_t i = grp; i < U; i += BLK / MULLS_LDS_GROUP)
		{
			nnkey bk;
			float sec, Rfin;
			uint32_t trips;
			CandOut co = {0xffffffffu, 0xffffffffu, 0.0f};
			search_query<MULLS_GLOB_CHUNK>(g, L, uq[i], C.r, C.m, sub, bk, sec, Rfin, trips, co);
			if (sub == 0 && (PARK ? commit_search_park<W16, BLK * TRIPS>(C, d, us[i] & ~MULLS_US_KCERT, bk, sec, Rfin, trips, co, hint2, W, park)
								  : commit_search<W16>(C, d, us[i] & ~MULLS_US_KCERT, bk, sec, Rfin, trips, co, nn_idx, nn_d2, hint2, W, winner)))
				matched_cnt++;
		}
	}
	for (int off = 32; off > 0; off >>= 1)
		matched_cnt += __shfl_down(matched_cnt, off);
	if ((threadIdx.x & 63) == 0)
		red[threadIdx.x >> 6] = matched_cnt;
	__threadfence_block(); // the searched (or k-candidate certified) points' results, read back below by the lanes that own them
	__syncthreads();
	uint32_t total_matched = 0;
	for (int w = 0; w < BLK / 64; w++)
		total_matched += red[w];
	FLAT_TICK(3)

	// duplicate rule + rejection chain (filter_point's decisions) on the registers
	const float thr = ps.thr[job.cls], max_sqr = thr * thr; // CorrespondenceRejectorDistance::setMaximumDistance (float)
	const bool any_match = total_matched > 0u, strict = rp.rej_strict != 0;
	uint32_t n_alive = 0, n_valid = 0;
#pragma unroll
	for (int k = 0; k < TRIPS; k++)
	{
		const uint32_t st = getST(k);
		if (!(st & MULLS_FS_ALIVE))
			continue;
		const uint32_t s = threadIdx.x + (uint32_t)k * BLK, gi = d.src_off + s;
		int32_t m = getM(k);
		float dist = getD0(k);
		bool standing = (st & MULLS_FS_STANDING) != 0;
		if (PARK ? (st & MULLS_FS_SEARCH) != 0u : m == MULLS_NEEDS_SEARCH) // (PARK: only what kcert_list certified is still marked — a searched point's words hold its result)
		{
			m = nn_idx[gi];
			dist = nn_d2[gi];
			standing = m >= 0 && match[gi] == m;
		}
		// first source (lowest index) matched to a target keeps it (cregistration.hpp:1762-1789); the others become unmatched
		if (C.dedup && m >= 0 && !dedup_holds<W16>(W, (uint32_t)m, s))
			m = -1;
		bool alive = true, valid, dir_ok = (st & MULLS_FS_DIR_OK) != 0;
		if (any_match)
		{
			valid = m >= 0;
			if (C.gate && m < 0) // (rp.lds_dedup: the duplicate rule was resolved above)
			{
				alive = false; // unmatched or duplicate-losing source points vanish for good (:1762-1789)
				valid = false;
			}
			if (valid)
			{
				if (!standing)
				{
					// a new nearest target: its record travels with the source point from here on (filter_point), and the direction check runs against the new
					// target direction.  Written whether or not the distance rejector below keeps the correspondence (only valid points' records are ever used, and
					// a valid point's record is its correspondence's: every change of the nearest target passes here): the next walk then finds the hinted target's
					// position in the point's own record instead of gathering it through the crop map — two dependent round trips in the middle of its certificates
					// for every wave that holds one such point
					match[gi] = m;
					float4 q2, n2;
					tgt_record(rp.tgt_stage, rp.tgt_map, d, (uint32_t)m, tpos, tnrm, q2, n2);
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
				valid = strict ? dist < max_sqr : !(dist > max_sqr); // CorrespondenceRejectorDistance (see mulls_params.rejector_strict)
				if (valid)
					wd[gi] = dist; // pcl::Correspondence::distance (shares storage with ::weight)
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
		red[BLK / 64 + (threadIdx.x >> 6)] = n_alive;
		red[2 * (BLK / 64) + (threadIdx.x >> 6)] = n_valid;
	}
	__syncthreads();
	if (threadIdx.x == 0)
	{
		uint32_t ta = 0, tv = 0;
		for (int w = 0; w < BLK / 64; w++)
		{
			ta += red[BLK / 64 + w];
			tv += red[2 * (BLK / 64) + w];
		}
		d.n_matched = total_matched; // k_finish reset it to 0 after the previous iteration
		d.alive_next = ta;
		d.valid_next = tv;
		d.n_search = U;
	}
	FLAT_TICK(4)
	if (rp.dbg_ticks && threadIdx.x == 0)
		atomicAdd(&rp.dbg_ticks[6], 1ull);
#undef FLAT_TICK
	return true;
}

// The LDS layout of the heavy pass: query block, cost-sort tables, the staged target class cloud (12-B position records, uint16
// original index, uint16 cell table) and the duplicate table.  `cap`: points the staged cloud may hold (driver: lds_cap).
struct LdsLayout
{
	float4 *qpos;	 // [MULLS_LDS_QCHUNK] queries of the chunk, w = sweep radius of a hinted query / +inf
	uint32_t *HIST;	 // [32] cost histogram, [32] bucket bases, [64] live queries of the chunk, [65] queue ticket
	uint16_t *ORDER; // [MULLS_LDS_QCHUNK] query slots, most expensive first
	float *P;		 // [3 * cap] x, y, z records
	uint16_t *IDX;	 // [cap]
	uint16_t *CS;	 // [grid_maxcells + 1]
	uint32_t *W;	 // [cap] lowest source index matched to each target (lds_dedup)
};
__device__ __forceinline__ LdsLayout lds_layout(unsigned char *lds_raw, uint32_t cap, uint32_t grid_maxcells)
{
	LdsLayout L;
	L.qpos = reinterpret_cast<float4 *>(lds_raw);
	L.HIST = reinterpret_cast<uint32_t *>(L.qpos + MULLS_LDS_QCHUNK);
	L.ORDER = reinterpret_cast<uint16_t *>(L.HIST + 80);
	L.P = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(L.HIST) + MULLS_LDS_AUX);
	L.IDX = reinterpret_cast<uint16_t *>(L.P + 3u * cap);
	L.CS = L.IDX + cap;
	L.W = reinterpret_cast<uint32_t *>(L.CS + ((grid_maxcells + 8u) & ~1u));
	return L;
}

// Heavy pass of one class cloud's iteration, MULLS_LDS_BLOCK lanes: stage the cell-sorted target class cloud, search the points
// cert_class left uncertified (nn_idx == MULLS_NEEDS_SEARCH) in equal chunks, then the class's tail (duplicate rule, rejection
// chain, counters).  Every lane of the workgroup must call it; the caller puts a barrier between two calls.
__device__ __forceinline__ void lds_search_class(const RunParams &rp, const PairState &ps, const Job &job, CloudDesc &d, const GridDesc &g,
												  const LdsLayout &Y, unsigned char *lds_raw, const float4 *__restrict__ spos,
												  const float4 *__restrict__ snrm, const uint32_t *__restrict__ cell_start,
												  const float4 *__restrict__ tsorted, uint8_t *flag, int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2,
												  unsigned long long *__restrict__ winner, const float4 *__restrict__ tnrm, int32_t *__restrict__ match,
												  float *__restrict__ wd, const float4 *__restrict__ tpos, int32_t *__restrict__ nn_hint, float4 *__restrict__ mq,
												  bool first = false)
{
	// first (uniform): iteration 0 without a light pass in front — the setup has applied the iteration's rigid step (identity_step), no point has a hint or a
	// correspondence, so every live point is a query of the whole rejection ball and nothing of nn_idx / nn_d2 / the hint records is read
	// diagnostics (MULLS_OPT_DEBUG_STOP = 20): phase clocks of the heavy pass, summed over its class clouds (rp.dbg_ticks[8..12])
	unsigned long long t_prev = rp.dbg_ticks ? wall_clock64() : 0ull;
#define HEAVY_TICK(k)                                                 \
	if (rp.dbg_ticks)                                                 \
	{                                                                 \
		const unsigned long long now_ = wall_clock64();               \
		if (threadIdx.x == 0)                                         \
			atomicAdd(&rp.dbg_ticks[k], now_ - t_prev);               \
		t_prev = now_;                                                \
	}
	float4 *qpos = Y.qpos;
	uint32_t *HIST = Y.HIST;
	uint16_t *ORDER = Y.ORDER;
	float *P = Y.P;
	uint16_t *IDX = Y.IDX, *CS = Y.CS;
	uint32_t *W = Y.W;
	int2 *__restrict__ hint2 = reinterpret_cast<int2 *>(nn_hint);
	const uint32_t src_n = d.src_n, tgt_n = d.tgt_n;
	const ClassCtx C = class_ctx(rp, ps, g, job.cls, d.alive_cur, true);
	const uint32_t q_end = min(src_n, job.start + (job.count ? job.count : (uint32_t)MULLS_SRC_PER_BLOCK));
	// chunks of equal size (1200 points: 2 x 600, not 1024 + 176: the last chunk would leave most sub-groups idle)
	const uint32_t q_cnt = q_end > job.start ? q_end - job.start : 0u, n_chunks = (q_cnt + MULLS_LDS_QCHUNK - 1u) / MULLS_LDS_QCHUNK;
	const uint32_t q_step = n_chunks ? (q_cnt + n_chunks - 1u) / n_chunks : 1u;

	// what k_cert left for the lanes of a chunk (one point per lane), loaded one chunk ahead — the first chunk's while the
	// target cloud is staged
	float4 pf_p = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	float pf_w = 0.0f;
	uint32_t pf_hv = 0u, pf_f = 0u;
	int32_t pf_i = -1;
	auto prefetch = [&](uint32_t chunk) {
		const uint32_t s = chunk + threadIdx.x, gi = d.src_off + s;
		pf_f = 0u;
		pf_i = -1;
		if (chunk < q_end && s < min(q_end, chunk + q_step))
		{
			pf_f = flag[gi];
			pf_p = spos[gi];
			if (first)
				pf_i = MULLS_NEEDS_SEARCH, pf_w = __builtin_inff(), pf_hv = 0u;
			else
			{
				pf_i = nn_idx[gi];
				pf_w = nn_d2[gi];
				pf_hv = (uint32_t)hint2[gi].x;
			}
		}
	};
	prefetch(job.start);

	// stage the cell-sorted target cloud and its cell table (coalesced reads).  Loads are issued in batches of 8 / 2
	// per lane before the first LDS write: one memory latency per batch instead of one per element.
	{
		const float4 *__restrict__ ts = tsorted + d.tgt_off;
		for (uint32_t k0 = threadIdx.x; k0 < tgt_n; k0 += 8 * MULLS_LDS_BLOCK)
		{
			float4 t[8];
#pragma unroll
			for (int u = 0; u < 8; u++)
			{
				const uint32_t k = k0 + u * MULLS_LDS_BLOCK;
				if (k < tgt_n)
					t[u] = ts[k];
			}
#pragma unroll
			for (int u = 0; u < 8; u++)
			{
				const uint32_t k = k0 + u * MULLS_LDS_BLOCK;
				if (k < tgt_n)
				{
					P[3u * k] = t[u].x;
					P[3u * k + 1u] = t[u].y;
					P[3u * k + 2u] = t[u].z;
					IDX[k] = (uint16_t)__float_as_int(t[u].w);
				}
			}
		}
		// cell table: (ncell + 1) uint16 entries written by k_grid_build_sort, moved as uint4 words of 8 (the table slot of a
		// cloud is uint4-aligned and padded)
		const uint4 *__restrict__ cs4 = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(cell_start) + g.cell_off);
		const uint32_t nw = (g.ncell + 1u + 7u) >> 3;
		for (uint32_t w0 = threadIdx.x; w0 < nw; w0 += 2 * MULLS_LDS_BLOCK)
		{
			// unconditional loads at clamped indices: a predicated `if (w < nw) v = ...` makes the compiler wait for every load on
			// its own (serialised round trips, seen in the ISA listing)
			const uint32_t w1 = w0 + MULLS_LDS_BLOCK;
			const uint4 va = cs4[w0], vb = cs4[min(w1, nw - 1u)];
			reinterpret_cast<uint4 *>(CS)[w0] = va;
			if (w1 < nw)
				reinterpret_cast<uint4 *>(CS)[w1] = vb;
		}
	}
	if (C.dedup)
		for (uint32_t t = threadIdx.x; t < tgt_n; t += MULLS_LDS_BLOCK)
			W[t] = 0xffffffffu;
	const LdsGrid L = {P, IDX, CS};
	const uint32_t sub = threadIdx.x & (MULLS_LDS_GROUP - 1u);
	uint32_t matched_cnt = 0, searched = 0;

	if (threadIdx.x < 32u)
		HIST[threadIdx.x] = 0u;
	for (uint32_t chunk = job.start; chunk < q_end; chunk += q_step)
	{
		__syncthreads(); // the previous chunk's queries have been consumed (and, first trip, the staging stores are visible below)
		HEAVY_TICK(chunk == job.start ? 8 : 10)
		uint32_t bucket = 0, rank = 0xffffffffu; // cost class of this lane's query (0 = most expensive) and its rank inside the class
		// phase 1: one source point per lane — uncertified points become the chunk's queries, certified matches enter the duplicate table
		if (threadIdx.x < MULLS_LDS_QCHUNK && (pf_f & MULLS_F_ALIVE))
		{
			if (pf_i == MULLS_NEEDS_SEARCH)
			{
				qpos[threadIdx.x] = make_float4(pf_p.x, pf_p.y, pf_p.z, pf_w);
				// cost class: the candidate trips the query took last time; an unhinted query (first iteration, stale hint) whose own cell is empty will
				// sweep the whole 2.5 * thr ball — those go first and together, the ones with neighbours in their cell after them
				bucket = 31u - ((pf_hv >> 16) & 31u);
				if (!(pf_w < __builtin_inff()))
				{
					const uint32_t c = grid_cell_id(g, pf_p.x, pf_p.y, pf_p.z);
					bucket = CS[c + 1u] > CS[c] ? 8u : 0u;
				}
				rank = atomicAdd(&HIST[bucket], 1u);
			}
			else if (pf_i >= 0)
			{
				matched_cnt++;
				if (C.dedup)
					atomicMin(&W[pf_i], chunk + threadIdx.x);
			}
		}
		prefetch(chunk + q_step); // the next chunk's loads, consumed after this chunk's search
		__syncthreads();
		// queries of the chunk in order of the work they took in the previous iteration (candidate trips, kept next to the hint):
		// the eight sub-groups of a wave run in lock step, so a wave is as slow as its most expensive query — neighbours in
		// this order cost about the same.  Counting sort over 32 classes; dead and certified points are not in it.
		if (threadIdx.x < 32u)
		{
			const uint32_t v = HIST[threadIdx.x];
			uint32_t incl = v;
			for (int off = 1; off < 32; off <<= 1)
			{
				const uint32_t o = __shfl_up(incl, off);
				if ((int)threadIdx.x >= off)
					incl += o;
			}
			HIST[32u + threadIdx.x] = incl - v;
			HIST[threadIdx.x] = 0u; // ready for the next chunk
			if (threadIdx.x == 31u)
			{
				HIST[64] = incl;
				HIST[66] = 0u; // the chunk's query counter (phase 2)
			}
		}
		__syncthreads();
		if (rank != 0xffffffffu)
			ORDER[HIST[32u + bucket] + rank] = (uint16_t)threadIdx.x;
		__syncthreads();
		const uint32_t n_live = HIST[64];
		searched += n_live;
		HEAVY_TICK(9)

		// phase 2: sub-groups of MULLS_LDS_GROUP lanes, one query at a time each.  A wave takes the next eight queries of the cost order from a
		// counter when its eight sub-groups are done (longest first, whoever is free: the waves finish together instead of by their luck with a
		// fixed share)
		for (;;)
		{
			uint32_t base = 0;
			if ((threadIdx.x & 63u) == 0u)
				base = atomicAdd(&HIST[66], 64u / MULLS_LDS_GROUP);
			base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
			if (base >= n_live)
				break;
			const uint32_t i = base + (threadIdx.x & 63u) / MULLS_LDS_GROUP;
			if (i < n_live)
			{
				const uint32_t k = ORDER[i];
				nnkey bk;
				float sec, Rfin;
				uint32_t trips;
				CandOut co = {0xffffffffu, 0xffffffffu, 0.0f};
				search_query(g, L, qpos[k], C.r, C.m, sub, bk, sec, Rfin, trips, co);
				if (sub == 0 && commit_search(C, d, chunk + k, bk, sec, Rfin, trips, co, nn_idx, nn_d2, hint2, W, winner))
					matched_cnt++;
			}
		}
	}
	HEAVY_TICK(10)
	uint32_t *red = reinterpret_cast<uint32_t *>(lds_raw); // the query block is free once the tail's first barrier has passed
	if (rp.lds_dedup && rp.debug_stop != 9u && job.start == 0u && q_end <= 2u * MULLS_LDS_BLOCK)
		class_tail_flat<MULLS_LDS_BLOCK, 2, false>(rp, ps, C, d, job, q_end, matched_cnt, searched, W, red, snrm, tnrm, flag, nn_idx, nn_d2, match, wd, tpos, mq);
	else
		class_tail<MULLS_LDS_BLOCK>(rp, ps, C, d, job, q_end, matched_cnt, searched, W, red, snrm, tnrm, flag, nn_idx, nn_d2, match, wd, winner, tpos, mq);
	HEAVY_TICK(11)
	if (rp.dbg_ticks && threadIdx.x == 0)
		atomicAdd(&rp.dbg_ticks[12], 1ull);
#undef HEAVY_TICK
}
} // namespace
