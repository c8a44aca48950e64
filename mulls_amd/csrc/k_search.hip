// k_search.hip — correspondence search (three exact tiers + normal shooting) and the rejection chain
// (gfx950 / CDNA4, wave64; numerics policy and launch geometry: device_util.h)
#include "device_util.h"
#include "lds_tier.h"
#include "big_tier.h"

// ---------------------------------------------------------------------------------------------------------------
// Correspondence search.  A workgroup is 2 wave64 (MULLS_NN_BLOCK = 128 lanes); each lane owns MULLS_NN_PTS = 4
// source points of the job's 512-point slice: it applies this iteration's rigid step (double math, float store, in
// place — the reference accumulates float rounding the same way), then scans the whole target class cloud.  Targets
// are streamed from HBM with coalesced 16-B loads and staged in LDS as three planar arrays X[], Y[], Z[], so that one
// ds_read_b128 (wave-uniform address -> broadcast) delivers one coordinate of FOUR targets: 6 LDS reads per 8
// targets against ~290 VALU instructions, which keeps the LDS pipe ~10 % busy and the kernel VALU-bound.
// Distances are FLANN's L2_Simple<float>: ((dx*dx)+(dy*dy))+(dz*dz), no FMA.  Per source point the scan keeps the
// running minimum and the first index of the 8-target group that produced it (one v_min3 chain + one compare per
// group instead of a compare/select pair per target); the exact target index — lowest index among bit-equal
// distances — is recovered afterwards by re-evaluating that one group from L2.
template <int NPTS>
__device__ __forceinline__ void nn_scan_group(const float *__restrict__ X, const float *__restrict__ Y, const float *__restrict__ Z, uint32_t j,
											   const float (&px)[NPTS], const float (&py)[NPTS], const float (&pz)[NPTS], float (&best)[NPTS],
											   uint32_t (&grp)[NPTS], uint32_t gidx)
{
	const float4 x0 = *reinterpret_cast<const float4 *>(X + j), x1 = *reinterpret_cast<const float4 *>(X + j + 4);
	const float4 y0 = *reinterpret_cast<const float4 *>(Y + j), y1 = *reinterpret_cast<const float4 *>(Y + j + 4);
	const float4 z0 = *reinterpret_cast<const float4 *>(Z + j), z1 = *reinterpret_cast<const float4 *>(Z + j + 4);
	const float tx[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
	const float ty[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
	const float tz[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
#pragma unroll
	for (int u = 0; u < NPTS; u++)
	{
		float dd[8];
#pragma unroll
		for (int v = 0; v < 8; v++)
		{
			const float dx = px[u] - tx[v], dy = py[u] - ty[v], dz = pz[u] - tz[v];
			dd[v] = (dx * dx + dy * dy) + dz * dz;
		}
		const float m = fminf(fminf(fminf(dd[0], dd[1]), fminf(dd[2], dd[3])), fminf(fminf(dd[4], dd[5]), fminf(dd[6], dd[7])));
		const bool upd = m < best[u]; // strict: an equal later distance never replaces an earlier one
		best[u] = upd ? m : best[u];
		grp[u] = upd ? gidx : grp[u];
	}
}
__global__ __launch_bounds__(MULLS_NN_BLOCK) void k_nn(const Job *__restrict__ jobs, CloudDesc *__restrict__ descs,
														const PairState *__restrict__ states, RunParams rp, float4 *__restrict__ spos,
														float4 *__restrict__ snrm, const float4 *__restrict__ tpos,
														const uint8_t *__restrict__ flag, int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2,
														unsigned long long *__restrict__ winner)
{
	__shared__ __attribute__((aligned(16))) float tileX[MULLS_TILE];
	__shared__ __attribute__((aligned(16))) float tileY[MULLS_TILE];
	__shared__ __attribute__((aligned(16))) float tileZ[MULLS_TILE];
	const Job job = jobs[blockIdx.x];
	const PairState &ps = states[job.pair];
	if (!ps.active || (rp.normal_shooting && (job.cls == 0 || job.cls == 2 || job.cls == 4)))
		return; // planar classes are served by k_nn_shoot while normal shooting is on
	CloudDesc &d = descs[job.pair * MULLS_NC + job.cls];
	const uint32_t src_n = d.src_n, tgt_n = d.tgt_n, alive_cur = d.alive_cur;
	const bool called = class_called(rp, d, job.cls);
	const float *__restrict__ tp = reinterpret_cast<const float *>(tpos + d.tgt_off);

	uint32_t s[MULLS_NN_PTS];
	bool alive[MULLS_NN_PTS];
	float px[MULLS_NN_PTS], py[MULLS_NN_PTS], pz[MULLS_NN_PTS];
#pragma unroll
	for (int u = 0; u < MULLS_NN_PTS; u++)
	{
		s[u] = job.start + threadIdx.x + u * MULLS_NN_BLOCK;
		alive[u] = s[u] < src_n && (flag[d.src_off + s[u]] & MULLS_F_ALIVE);
		px[u] = py[u] = pz[u] = 0.0f;
		if (alive[u])
		{
			// pcl::transformPointCloudWithNormals<PointT,double> (cregistration.hpp:1690-1695)
			float4 p = spos[d.src_off + s[u]], n = snrm[d.src_off + s[u]];
			const double *T = ps.T;
			double x = p.x, y = p.y, z = p.z, nx = n.x, ny = n.y, nz = n.z;
			px[u] = (float)(T[0] * x + T[1] * y + T[2] * z + T[3]);
			py[u] = (float)(T[4] * x + T[5] * y + T[6] * z + T[7]);
			pz[u] = (float)(T[8] * x + T[9] * y + T[10] * z + T[11]);
			float onx = (float)(T[0] * nx + T[1] * ny + T[2] * nz);
			float ony = (float)(T[4] * nx + T[5] * ny + T[6] * nz);
			float onz = (float)(T[8] * nx + T[9] * ny + T[10] * nz);
			spos[d.src_off + s[u]] = make_float4(px[u], py[u], pz[u], p.w);
			snrm[d.src_off + s[u]] = make_float4(onx, ony, onz, n.w);
		}
	}
	if (!called)
		return; // correspondences of the previous iteration stay in force

	const float INF = __builtin_inff();
	float best[MULLS_NN_PTS];
	uint32_t grp[MULLS_NN_PTS]; // first target index of the winning 8-group
#pragma unroll
	for (int u = 0; u < MULLS_NN_PTS; u++)
	{
		best[u] = INF;
		grp[u] = 0;
	}

	for (uint32_t base = 0; base < tgt_n; base += MULLS_TILE)
	{
		const uint32_t nt = min((uint32_t)MULLS_TILE, tgt_n - base);
		const uint32_t nt8 = (nt + 7u) & ~7u;
		__syncthreads();
		for (uint32_t k = threadIdx.x; k < nt8; k += MULLS_NN_BLOCK)
		{
			// +inf padding keeps the unrolled scan free of tail code: (p - inf)^2 = inf never beats a finite minimum
			const float4 t = (k < nt) ? tpos[d.tgt_off + base + k] : make_float4(INF, INF, INF, 0.0f);
			tileX[k] = t.x;
			tileY[k] = t.y;
			tileZ[k] = t.z;
		}
		__syncthreads();
		for (uint32_t j = 0; j < nt8; j += 8)
			nn_scan_group<MULLS_NN_PTS>(tileX, tileY, tileZ, j, px, py, pz, best, grp, base + j);
	}

	// recover the exact index inside the winning group (bit-identical re-evaluation)
	const float r = 2.5f * ps.thr[job.cls];	 // filter_dis_times * dis_thre (float), cregistration.hpp:1745
	const double maxd = (double)r;			 // widened to the `double max_distance` parameter
	const double max_dist_sqr = maxd * maxd; // CorrespondenceEstimation::determineCorrespondences
	const bool gate = alive_cur >= 500u;	 // K_filter_distant_point, cregistration.hpp:1755
	const unsigned long long key_hi = (unsigned long long)(0xffffffffu - (rp.tick_base + (uint32_t)ps.iter)) << 32;
	uint32_t matched_cnt = 0;
#pragma unroll
	for (int u = 0; u < MULLS_NN_PTS; u++)
	{
		if (!alive[u])
			continue;
		int idx = -1;
		for (int v = 7; v >= 0; v--)
		{
			const uint32_t t = grp[u] + v;
			if (t < tgt_n)
			{
				float ddx = px[u] - tp[4 * t], ddy = py[u] - tp[4 * t + 1], ddz = pz[u] - tp[4 * t + 2];
				float dist = (ddx * ddx + ddy * ddy) + ddz * ddz;
				if (dist == best[u])
					idx = (int)t;
			}
		}
		const bool matched = idx >= 0 && !((double)best[u] > max_dist_sqr);
		nn_idx[d.src_off + s[u]] = matched ? idx : -1;
		nn_d2[d.src_off + s[u]] = best[u];
		if (matched)
		{
			matched_cnt++;
			if (gate) // duplicate rule: the lowest source index claims the target (first-come in the reference's serial walk)
				atomicMin(&winner[d.tgt_off + idx], key_hi | (unsigned long long)s[u]);
		}
	}
	for (int off = 32; off > 0; off >>= 1)
		matched_cnt += __shfl_down(matched_cnt, off);
	if ((threadIdx.x & 63) == 0 && matched_cnt)
		atomicAdd(&d.n_matched, matched_cnt);
}

// Correspondence search, global-memory tier (big_tier.h): rigid step + certificates + search against the occupancy-bitmap grid for the class clouds that do
// not fit the LDS tier.  A job is a whole source class cloud of at most 1536 points (MULLS_JOB_CLASS: the workgroup also resolves the duplicate rule and
// runs the rejection chain — no k_filter) or 512 consecutive points of a larger one, which `split` workgroups share when the launch would not fill the chip
// otherwise (k_filter finishes those clouds).  Outputs are identical to k_nn.
__global__ __launch_bounds__(MULLS_BIG_BLOCK, 4) void k_cert_big(const Job *__restrict__ jobs, CloudDesc *__restrict__ descs, const PairState *__restrict__ states, RunParams rp,
																 float4 *__restrict__ spos, float4 *__restrict__ snrm, const GridDesc *__restrict__ grids,
																 const unsigned long long *__restrict__ bm, const uint32_t *__restrict__ pf, const uint32_t *__restrict__ cs,
																 const float4 *__restrict__ tsorted, uint8_t *flag, int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2,
																 unsigned long long *__restrict__ winner, uint32_t split, const float4 *__restrict__ tpos,
																 const float4 *__restrict__ tnrm, int32_t *__restrict__ nn_hint, int32_t *__restrict__ match, float *__restrict__ wd,
																 float4 *__restrict__ mq)
{
	__shared__ BigLds<MULLS_BIG_BLOCK * 3> s_big;
	const uint32_t wg = xcd_job(blockIdx.x, gridDim.x);
	const Job job = jobs[wg / split];
	const uint32_t part = wg % split;
	const PairState &ps = states[job.pair];
	if (!ps.active || (rp.normal_shooting && (job.cls == 0 || job.cls == 2 || job.cls == 4)))
		return; // planar classes are served by k_nn_shoot while normal shooting is on
	const uint32_t ci = job.pair * MULLS_NC + job.cls;
	CloudDesc &d = descs[ci];
	const bool class_level = (job.count & MULLS_JOB_CLASS) != 0u;
	const uint32_t cnt = (job.count & ~MULLS_JOB_CLASS) ? (job.count & ~MULLS_JOB_CLASS) : (uint32_t)MULLS_SRC_PER_BLOCK;
	uint32_t q0, q1;
	if (class_level)
	{
		if (part)
			return; // a class-level job is one workgroup's
		q0 = 0u, q1 = min(d.src_n, cnt);
	}
	else
	{
		const uint32_t per = cnt / split;
		q0 = job.start + part * per, q1 = min(d.src_n, q0 + per);
		if (q0 >= q1)
			return;
	}
	const GridDesc g = grids[ci];
	const BmGrid B = {bm + g.cell_off, pf + g.cell_off, cs + d.tgt_off + ci};
	cert_big<MULLS_BIG_BLOCK, 3>(s_big, rp, ps, job.cls, q0, q1, class_level, d, g, B, tsorted + d.tgt_off, spos, snrm, flag, nn_idx, nn_d2, winner, tnrm, match, wd, tpos,
								 reinterpret_cast<int2 *>(nn_hint), mq);
}

// rejection chain of the class clouds whose correspondences are spread over several workgroups (chunk-level jobs).  big: jobs of the global-memory tier
// (duplicate rule through the batch's winner table, cropped target copies) whatever the LDS tier of the same batch does
__global__ __launch_bounds__(MULLS_BLOCK) void k_filter(const Job *__restrict__ jobs, CloudDesc *__restrict__ descs,
														 const PairState *__restrict__ states, RunParams rp,
														 const float4 *__restrict__ snrm, const float4 *__restrict__ tnrm, uint8_t *__restrict__ flag,
														 const int32_t *__restrict__ nn_idx, const float *__restrict__ nn_d2,
														 int32_t *__restrict__ match, float *__restrict__ wd,
														 const unsigned long long *__restrict__ winner, const float4 *__restrict__ tpos,
														 float4 *__restrict__ mq, uint32_t big)
{
	__shared__ uint32_t red4[4];
	const Job job = jobs[xcd_job(blockIdx.x, gridDim.x)];
	const PairState &ps = states[job.pair];
	if (!ps.active)
		return;
	CloudDesc &d = descs[job.pair * MULLS_NC + job.cls];
	if (!class_called(rp, d, job.cls))
		return;
	const float thr = ps.thr[job.cls];
	// vertex correspondences skip the direction check (cregistration.hpp:1292)
	const FilterCtx F = {d.alive_cur >= 500u, d.n_matched > 0u, job.cls != 5, rp.lds_dedup != 0u && !big, rp.rej_strict != 0, thr * thr, rp.cos_bearing,
						 (unsigned long long)(0xffffffffu - (rp.tick_base + (uint32_t)ps.iter)) << 32, big ? nullptr : rp.tgt_stage, big ? nullptr : rp.tgt_map};
	uint32_t n_alive = 0, n_valid = 0;
#pragma unroll
	for (int u = 0; u < MULLS_SRC_PER_THREAD; u++)
	{
		const uint32_t s = job.start + threadIdx.x + u * MULLS_BLOCK;
		if (s < d.src_n)
			filter_point(F, d, s, snrm, tnrm, flag, nn_idx, nn_d2, match, wd, winner, tpos, mq, n_alive, n_valid);
	}
	const uint32_t ta = block_sum_u32(n_alive, red4);
	const uint32_t tv = block_sum_u32(n_valid, red4);
	if (threadIdx.x == 0)
	{
		if (ta)
			atomicAdd(&d.alive_next, ta);
		if (tv)
			atomicAdd(&d.valid_next, tv);
	}
}


// Iteration 0 of a class-level job (`first`: the host's word that this launch set is the run's first and that the setup has applied the iteration's rigid step,
// identity_step).  No point has a hint, so the light pass would list every live point and hand the class cloud to the staged search after one walk's worth
// of loads and stores (k_cert 680 - 710 us of a 4096-pair launch, 290 us once everything certifies) — or, up to its own search budget of points, search them
// all unhinted against the grid in global memory, seven rounds of dependent gathers for a 400-point pillar cloud (470 us of a launch for those alone).  Every
// called class cloud therefore skips the light pass in iteration 0: lds_search_class(first) reads the flags and positions the setup wrote and searches every
// live point against the staged cloud.  A class that sits the iteration out has nothing to do (its points do not move; it is never called later: its sizes
// only shrink).
__device__ __forceinline__ bool first_goes_direct(const RunParams &rp, const CloudDesc &d, int cls)
{
	return class_called(rp, d, cls);
}

// One class-cloud job of the light pass: the one-pass walk for a whole class cloud that fits the lanes' registers (cert_class_flat), the general walk for
// chunk-level jobs, larger clouds and classes that sit the iteration out.  Returns (to every lane) false when the class cloud needs the heavy pass.
template <int BLK, int SMALL, bool PARK = false, int FLAT_TRIPS = (BLK == 512 ? 3 : 2)>
__device__ __forceinline__ bool cert_job(CertLds<SMALL> &CL, const RunParams &rp, const PairState &ps, const Job &job, CloudDesc &d, const GridDesc &g, uint32_t *W, float4 *__restrict__ spos,
										  float4 *__restrict__ snrm, const uint32_t *__restrict__ cell_start, const float4 *__restrict__ tsorted, uint8_t *flag,
										  int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2, unsigned long long *__restrict__ winner, const float4 *__restrict__ tnrm,
										  int32_t *__restrict__ match, float *__restrict__ wd, const float4 *__restrict__ tpos, int32_t *__restrict__ nn_hint,
										  float4 *__restrict__ mq, uint32_t *park = nullptr)
{
	const bool flat = rp.lds_dedup != 0u && rp.debug_stop != 9u && job.start == 0u && job.count >= d.src_n && d.src_n <= (uint32_t)(BLK * FLAT_TRIPS) && class_called(rp, d, job.cls);
	return flat ? cert_class_flat<BLK, FLAT_TRIPS, true, SMALL, PARK>(CL, rp, ps, job, d, g, W, spos, snrm, cell_start, tsorted, flag, nn_idx, nn_d2, winner, tnrm, match, wd, tpos, nn_hint, mq, park)
				: cert_class<BLK, true, SMALL>(CL, rp, ps, job, d, g, W, spos, snrm, cell_start, tsorted, flag, nn_idx, nn_d2, winner, tnrm, match, wd, tpos, nn_hint, mq);
}

// BLK lanes per class cloud: 512 when there are enough class clouds to give every CU several workgroups, 1024 for small batches (a class cloud of
// 1200 points is then two trips instead of three, and the launch is as long as its longest workgroup).  TRIPS: the one-pass walk takes class clouds of up to
// BLK * TRIPS source points (larger ones: the general walk).  Longer one-pass walks for the 1 537 - 3 072-point clouds of real scans — 512 lanes x 6 trips,
// 1024 lanes x 3 trips, as their own launch or for every job — were built, bit-identical, and slower than the general walk on configs[0]
// (profiles/r06_experiments.txt item 5): a real scan's time is in the leftover searches of its SHORT class clouds.
// SMALL: the leftover list a workgroup searches itself (longer ones go to the staged search).  512 entries — or 768 where four workgroups per CU still fit with them
// (768 entries + the parked words + the 16-bit duplicate table of the batch's largest target cloud <= 40 KiB, i.e. clouds of up to ~6 000 points: the reference's
// real scans, whose every iteration carries a few hundred leftover searches per class cloud: configs[0] at 512 pairs +2.4 %, profiles/r06_experiments.txt items 29, 32)
template <int BLK, int TRIPS = (BLK == 512 ? 3 : 2), int SMALL = MULLS_CERT_SMALL_LOCKSTEP>
__global__ __launch_bounds__(BLK, BLK == 512 ? 8 : 4) void k_cert(const Job *__restrict__ cjobs, CloudDesc *__restrict__ descs,
															const PairState *__restrict__ states, RunParams rp, float4 *__restrict__ spos,
															float4 *__restrict__ snrm, const GridDesc *__restrict__ grids,
															const uint32_t *__restrict__ cell_start, const float4 *__restrict__ tsorted, uint8_t *flag,
															int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2, unsigned long long *__restrict__ winner,
															const float4 *__restrict__ tnrm, int32_t *__restrict__ match, float *__restrict__ wd,
															const float4 *__restrict__ tpos, int32_t *__restrict__ nn_hint, float4 *__restrict__ mq,
															uint32_t *__restrict__ wl, uint32_t *__restrict__ wl_ctr, uint32_t parity)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	uint32_t *W = reinterpret_cast<uint32_t *>(lds_raw); // [cap / 2] lowest source index matched to each target, 16-bit entries (lds_dedup; dedup_min<true>)
	if (blockIdx.x == 0 && threadIdx.x == 0)
		wl_ctr[2u * (parity ^ 1u)] = wl_ctr[2u * (parity ^ 1u) + 1u] = 0u; // the next iteration's queue (this one's predecessor has been drained)
	const Job job = cjobs[blockIdx.x];
	const PairState &ps = states[job.pair];
	if (!ps.active || (rp.normal_shooting && (job.cls == 0 || job.cls == 2 || job.cls == 4)))
		return;
	CloudDesc &d = descs[job.pair * MULLS_NC + job.cls];
	const GridDesc g = grids[job.pair * MULLS_NC + job.cls];
	__shared__ CertLds<SMALL> s_cert;
	constexpr bool PARK = BLK == 512; // (the two-trip 1024-lane form serves small batches: two workgroups per CU, 128 registers)
	__shared__ uint32_t s_park[PARK ? 2 * TRIPS * BLK : 1];
	if (!cert_job<BLK, SMALL, PARK, TRIPS>(s_cert, rp, ps, job, d, g, W, spos, snrm, cell_start, tsorted, flag, nn_idx, nn_d2, winner, tnrm, match, wd, tpos, nn_hint, mq, s_park))
		if (threadIdx.x == 0)
			wl[atomicAdd(&wl_ctr[2u * parity], 1u)] = blockIdx.x; // k_nn_lds stages the target cloud and takes the class cloud from here
}

// The fused kernels' light pass: 1024 lanes, three trips (class clouds of up to 3072 source points: a single registration of real scans), the walk's per-point
// state parked in the dynamic LDS block behind the 16-bit duplicate table (2 * cap bytes) — fused_lds_bytes() says how much the launch needs for it
#define MULLS_FUSED_PARK_BYTES (2u * 3u * MULLS_LDS_BLOCK * 4u)
__device__ __forceinline__ uint32_t *fused_park(unsigned char *lds_raw, uint32_t cap) { return reinterpret_cast<uint32_t *>(lds_raw + (((size_t)cap * 2u + 15u) & ~(size_t)15)); }

// Light and heavy pass in one launch, for batches of at most two class clouds per CU (a single registration: three): the workgroup that finds its class cloud
// in need of the heavy pass runs it right away (1024 lanes, the whole LDS) instead of queueing it for k_nn_lds — one launch less per iteration where
// the iteration is bound by the number of launches.  Same functions, same results.
__global__ __launch_bounds__(MULLS_LDS_BLOCK) void k_cert_nn(const Job *__restrict__ cjobs, CloudDesc *__restrict__ descs, const PairState *__restrict__ states,
															  RunParams rp, float4 *__restrict__ spos, float4 *__restrict__ snrm, const GridDesc *__restrict__ grids,
															  const uint32_t *__restrict__ cell_start, const float4 *__restrict__ tsorted, uint8_t *flag,
															  int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2, unsigned long long *__restrict__ winner,
															  const float4 *__restrict__ tnrm, int32_t *__restrict__ match, float *__restrict__ wd,
															  const float4 *__restrict__ tpos, int32_t *__restrict__ nn_hint, float4 *__restrict__ mq, uint32_t cap, uint32_t first)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	const Job job = cjobs[blockIdx.x];
	const PairState &ps = states[job.pair];
	if (!ps.active || (rp.normal_shooting && (job.cls == 0 || job.cls == 2 || job.cls == 4)))
		return;
	CloudDesc &d = descs[job.pair * MULLS_NC + job.cls];
	const GridDesc g = grids[job.pair * MULLS_NC + job.cls];
	__shared__ CertLds<MULLS_CERT_SMALL> s_cert; // (the heavy pass follows in this very workgroup: a small leftover budget, and the LDS for the staged cloud)
	const bool direct = first && first_goes_direct(rp, d, job.cls); // (uniform) iteration 0: straight to the staged search
	if (!direct)
	{
		if (cert_job<MULLS_LDS_BLOCK, MULLS_CERT_SMALL, true, 3>(s_cert, rp, ps, job, d, g, reinterpret_cast<uint32_t *>(lds_raw), spos, snrm, cell_start, tsorted, flag, nn_idx, nn_d2, winner,
																 tnrm, match, wd, tpos, nn_hint, mq, fused_park(lds_raw, cap)))
			return;
		__syncthreads(); // the light pass's LDS is free
	}
	lds_search_class(rp, ps, job, d, g, lds_layout(lds_raw, cap, rp.grid_maxcells), lds_raw, spos, snrm, cell_start, tsorted, flag, nn_idx, nn_d2, winner, tnrm, match,
					 wd, tpos, nn_hint, mq, direct);
}

// A small MIXED batch (a scan against a local map whose ground class outgrew the LDS tier): the class clouds of both tiers in ONE launch.  Workgroups
// [0, n_lds) are k_cert_nn's (1024 lanes, the staged search at hand), the rest k_cert_big's (their upper eight waves leave at once; BigLds lives in the dynamic
// LDS block) — the two kernels ran one after the other although neither reads what the other writes: 14 + 8 us per iteration of a single registration.
// Same functions, same results; k_filter follows for the chunk-level jobs as before.
__global__ __launch_bounds__(MULLS_LDS_BLOCK) void k_cert_mixed(uint32_t n_lds, const Job *__restrict__ cjobs, const Job *__restrict__ bjobs, uint32_t split,
																 CloudDesc *__restrict__ descs, const PairState *__restrict__ states, RunParams rp, float4 *__restrict__ spos,
																 float4 *__restrict__ snrm, const GridDesc *__restrict__ grids, const uint32_t *__restrict__ cell_start,
																 const unsigned long long *__restrict__ bm, const uint32_t *__restrict__ pf, const uint32_t *__restrict__ cs,
																 const float4 *__restrict__ tsorted, uint8_t *flag, int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2,
																 unsigned long long *__restrict__ winner, const float4 *__restrict__ tnrm, int32_t *__restrict__ match,
																 float *__restrict__ wd, const float4 *__restrict__ tpos, int32_t *__restrict__ nn_hint, float4 *__restrict__ mq, uint32_t cap,
																 uint32_t first)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	if (blockIdx.x < n_lds)
	{
		const Job job = cjobs[blockIdx.x];
		const PairState &ps = states[job.pair];
		if (!ps.active)
			return;
		CloudDesc &d = descs[job.pair * MULLS_NC + job.cls];
		const GridDesc g = grids[job.pair * MULLS_NC + job.cls];
		__shared__ CertLds<MULLS_CERT_SMALL> s_cert;
		const bool direct = first && first_goes_direct(rp, d, job.cls); // (uniform) iteration 0: straight to the staged search
		if (!direct)
		{
			if (cert_job<MULLS_LDS_BLOCK, MULLS_CERT_SMALL, true, 3>(s_cert, rp, ps, job, d, g, reinterpret_cast<uint32_t *>(lds_raw), spos, snrm, cell_start, tsorted, flag, nn_idx, nn_d2,
																	 winner, tnrm, match, wd, tpos, nn_hint, mq, fused_park(lds_raw, cap)))
				return;
			__syncthreads(); // the light pass's LDS is free
		}
		lds_search_class(rp, ps, job, d, g, lds_layout(lds_raw, cap, rp.grid_maxcells), lds_raw, spos, snrm, cell_start, tsorted, flag, nn_idx, nn_d2, winner, tnrm, match, wd, tpos,
						 nn_hint, mq, direct);
		return;
	}
	if (threadIdx.x >= MULLS_BIG_BLOCK)
		return; // (a terminated wave does not hold the others' barriers up)
	const uint32_t wg = blockIdx.x - n_lds;
	const Job job = bjobs[wg / split];
	const uint32_t part = wg % split;
	const PairState &ps = states[job.pair];
	if (!ps.active)
		return;
	const uint32_t ci = job.pair * MULLS_NC + job.cls;
	CloudDesc &d = descs[ci];
	const bool class_level = (job.count & MULLS_JOB_CLASS) != 0u;
	const uint32_t cnt = (job.count & ~MULLS_JOB_CLASS) ? (job.count & ~MULLS_JOB_CLASS) : (uint32_t)MULLS_SRC_PER_BLOCK;
	uint32_t q0, q1;
	if (class_level)
	{
		if (part)
			return;
		q0 = 0u, q1 = min(d.src_n, cnt);
	}
	else
	{
		const uint32_t per = cnt / split;
		q0 = job.start + part * per, q1 = min(d.src_n, q0 + per);
		if (q0 >= q1)
			return;
	}
	const GridDesc g = grids[ci];
	const BmGrid B = {bm + g.cell_off, pf + g.cell_off, cs + d.tgt_off + ci};
	cert_big<MULLS_BIG_BLOCK, 3>(*reinterpret_cast<BigLds<MULLS_BIG_BLOCK * 3> *>(lds_raw), rp, ps, job.cls, q0, q1, class_level, d, g, B, tsorted + d.tgt_off, spos, snrm, flag, nn_idx,
								 nn_d2, winner, tnrm, match, wd, tpos, reinterpret_cast<int2 *>(nn_hint), mq);
}

__global__ __launch_bounds__(MULLS_LDS_BLOCK) void k_nn_lds(const Job *__restrict__ cjobs, CloudDesc *__restrict__ descs,
															 const PairState *__restrict__ states, RunParams rp, const float4 *__restrict__ spos,
															 const float4 *__restrict__ snrm, const GridDesc *__restrict__ grids,
															 const uint32_t *__restrict__ cell_start, const float4 *__restrict__ tsorted,
															 uint8_t *flag, int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2,
															 unsigned long long *__restrict__ winner, const float4 *__restrict__ tnrm, int32_t *__restrict__ match,
															 float *__restrict__ wd, const float4 *__restrict__ tpos, int32_t *__restrict__ nn_hint, float4 *__restrict__ mq,
															 uint32_t cap, const uint32_t *__restrict__ wl, uint32_t *__restrict__ wl_ctr, uint32_t parity, uint32_t first_njobs)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	const LdsLayout Y = lds_layout(lds_raw, cap, rp.grid_maxcells);

	// persistent workgroup: class clouds come from the queue k_cert filled (complete before this kernel starts) — or, iteration 0 (first_njobs != 0: the
	// table's length), straight from the job table: no light pass ran (first_goes_direct)
	const uint32_t n_queued = first_njobs ? first_njobs : wl_ctr[2u * parity];
	for (;;)
	{
		__syncthreads(); // the previous class cloud's LDS contents have been consumed
		if (threadIdx.x == 0)
			Y.HIST[65] = atomicAdd(&wl_ctr[2u * parity + 1u], 1u);
		__syncthreads();
		const uint32_t ticket = Y.HIST[65];
		if (ticket >= n_queued)
			return;
		const Job job = cjobs[first_njobs ? ticket : wl[ticket]];
		CloudDesc &d = descs[job.pair * MULLS_NC + job.cls];
		if (first_njobs && !(states[job.pair].active && first_goes_direct(rp, d, job.cls)))
			continue;
		lds_search_class(rp, states[job.pair], job, d, grids[job.pair * MULLS_NC + job.cls], Y, lds_raw, spos, snrm, cell_start, tsorted, flag, nn_idx, nn_d2, winner, tnrm,
						 match, wd, tpos, nn_hint, mq, first_njobs != 0u);
	}
}

// ---------------------------------------------------------------------------------------------------------------
// Normal-shooting correspondence search (normal_shooting_on; planar classes only: cregistration.hpp:1730-1739, PCL's
// CorrespondenceEstimationNormalShooting with k = 10).  Among the 10 nearest targets (ascending (d^2, index)) the one
// minimising |n_s x (p_t - p_s)|^2 (double) wins; it is rejected if that minimum exceeds max_distance (compared with
// r = 2.5*thr, not r^2 — PCL quirk); the stored distance is that candidate's squared Euclidean distance.  No shipped
// configuration enables the option, so this kernel is written for exactness, not speed: one query per lane, targets
// broadcast from an LDS tile, a sorted 10-entry list per lane.
__global__ __launch_bounds__(MULLS_BLOCK) void k_nn_shoot(const Job *__restrict__ jobs, CloudDesc *__restrict__ descs,
														   const PairState *__restrict__ states, RunParams rp, float4 *__restrict__ spos,
														   float4 *__restrict__ snrm, const float4 *__restrict__ tpos,
														   const uint8_t *__restrict__ flag, int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2,
														   unsigned long long *__restrict__ winner)
{
	__shared__ float4 tile[1024];
	const Job job = jobs[blockIdx.x];
	if (!(job.cls == 0 || job.cls == 2 || job.cls == 4))
		return; // pillar / beam / vertex always use the plain nearest neighbour
	const PairState &ps = states[job.pair];
	if (!ps.active)
		return;
	CloudDesc &d = descs[job.pair * MULLS_NC + job.cls];
	const uint32_t src_n = d.src_n, tgt_n = d.tgt_n, alive_cur = d.alive_cur;
	const bool called = class_called(rp, d, job.cls);
	const float r = 2.5f * ps.thr[job.cls];
	const double max_distance = (double)r;
	const bool gate = alive_cur >= 500u;
	const unsigned long long key_hi = (unsigned long long)(0xffffffffu - (rp.tick_base + (uint32_t)ps.iter)) << 32;
	uint32_t matched_cnt = 0;
	for (int u = 0; u < MULLS_SRC_PER_THREAD; u++) // uniform trip count: the tile loop below contains barriers
	{
		const uint32_t s = job.start + threadIdx.x + u * MULLS_BLOCK;
		const bool alive = s < src_n && (flag[d.src_off + s] & MULLS_F_ALIVE);
		float px = 0, py = 0, pz = 0, nx = 0, ny = 0, nz = 0;
		if (alive)
		{
			const float4 p = spos[d.src_off + s], n = snrm[d.src_off + s];
			const double *T = ps.T;
			const double x = p.x, y = p.y, z = p.z, ax = n.x, ay = n.y, az = n.z;
			px = (float)(T[0] * x + T[1] * y + T[2] * z + T[3]);
			py = (float)(T[4] * x + T[5] * y + T[6] * z + T[7]);
			pz = (float)(T[8] * x + T[9] * y + T[10] * z + T[11]);
			nx = (float)(T[0] * ax + T[1] * ay + T[2] * az);
			ny = (float)(T[4] * ax + T[5] * ay + T[6] * az);
			nz = (float)(T[8] * ax + T[9] * ay + T[10] * az);
			spos[d.src_off + s] = make_float4(px, py, pz, p.w);
			snrm[d.src_off + s] = make_float4(nx, ny, nz, n.w);
		}
		if (!called)
			continue;
		float kd[10];
		int ki[10];
#pragma unroll
		for (int j = 0; j < 10; j++)
		{
			kd[j] = __builtin_inff();
			ki[j] = 0x7fffffff;
		}
		for (uint32_t base = 0; base < tgt_n; base += 1024)
		{
			const uint32_t nt = min(1024u, tgt_n - base);
			__syncthreads();
			for (uint32_t k = threadIdx.x; k < nt; k += MULLS_BLOCK)
				tile[k] = tpos[d.tgt_off + base + k];
			__syncthreads();
			for (uint32_t j = 0; j < nt; j++)
			{
				const float4 t = tile[j];
				const float dx = px - t.x, dy = py - t.y, dz = pz - t.z;
				float dist = (dx * dx + dy * dy) + dz * dz;
				int idx = (int)(base + j);
				if (dist < kd[9]) // indices arrive in ascending order: an equal distance never displaces an earlier index
				{
#pragma unroll
					for (int q = 0; q < 10; q++) // sorted insertion by one pass of compare-exchange
					{
						const bool lt = dist < kd[q];
						const float td = kd[q];
						const int ti = ki[q];
						kd[q] = lt ? dist : td;
						ki[q] = lt ? idx : ti;
						dist = lt ? td : dist;
						idx = lt ? ti : idx;
					}
				}
			}
		}
		if (!alive)
			continue;
		double min_dist = 1.7976931348623157e308;
		int min_index = 0;
		// entries are filled from the front; with a NaN query (a singular solve upstream propagates NaNs, SURVEY B-11) no distance
		// compares below infinity and nothing is found — the oracle's kd-tree returns an empty list there, too
		int found = 0;
#pragma unroll
		for (int j = 0; j < 10; j++)
			found += ki[j] != 0x7fffffff ? 1 : 0;
#pragma unroll
		for (int j = 0; j < 10; j++)
			if (j < found)
			{
				const float4 t = tpos[d.tgt_off + (uint32_t)ki[j]];
				const float ptx = px - t.x, pty = py - t.y, ptz = pz - t.z; // PCL forms the difference in float
				const double Vx = ptx, Vy = pty, Vz = ptz, Nx = nx, Ny = ny, Nz = nz;
				const double cx = Ny * Vz - Nz * Vy, cy = Nz * Vx - Nx * Vz, cz = Nx * Vy - Ny * Vx;
				const double dist = cx * cx + cy * cy + cz * cz;
				if (dist < min_dist)
				{
					min_dist = dist;
					min_index = j;
				}
			}
		float sel_d = 0.0f;
		int sel_i = -1;
#pragma unroll
		for (int j = 0; j < 10; j++)
			if (j == min_index)
			{
				sel_d = kd[j];
				sel_i = ki[j];
			}
		const bool matched = found > 0 && !(min_dist > max_distance);
		nn_idx[d.src_off + s] = matched ? sel_i : -1;
		nn_d2[d.src_off + s] = sel_d;
		if (matched)
		{
			matched_cnt++;
			if (gate)
				atomicMin(&winner[d.tgt_off + sel_i], key_hi | (unsigned long long)s);
		}
	}
	for (int off = 32; off > 0; off >>= 1)
		matched_cnt += __shfl_down(matched_cnt, off);
	if ((threadIdx.x & 63) == 0 && matched_cnt)
		atomicAdd(&d.n_matched, matched_cnt);
}

// ---------------------------------------------------------------------------------------------------------------
// host-callable launch wrappers (the driver is plain C++ and never sees <<< >>>)
#include "launch.h"

// dynamic LDS of the fused kernels (k_cert_nn, k_cert_mixed): the staged search's block, or what the light pass in front of it parks (fused_park)
static size_t fused_lds_bytes(uint32_t cap, uint32_t maxcells, bool dedup)
{
	const size_t heavy = nn_lds_bytes(cap, maxcells, dedup), light = (((size_t)cap * 2u + 15u) & ~(size_t)15) + MULLS_FUSED_PARK_BYTES;
	return heavy > light ? heavy : light;
}
size_t nn_lds_bytes(uint32_t cap, uint32_t maxcells, bool dedup)
{
	// query block, cost-sort tables, position records + index, cell table, and (lds_dedup) the on-chip duplicate table
	return (size_t)MULLS_LDS_QCHUNK * 16u + (size_t)MULLS_LDS_AUX + (size_t)cap * 14u + (((size_t)maxcells + 8u) & ~(size_t)1) * 2u + (dedup ? (size_t)cap * 4u : 0u);
}

// one iteration of the LDS tier over the class clouds `jobs[0..njobs)`: k_cert for all of them, then k_nn_lds for the ones it
// queued in `wl` (njobs entries; wl_ctr: 4 counters = (queued, taken) x the parity of this launch and of the next one)
int launch_nn_lds(hipStream_t st, uint32_t njobs, const Job *jobs, CloudDesc *descs, const PairState *states, const RunParams &rp, float4 *spos,
				  float4 *snrm, const GridDesc *grids, const uint32_t *cell_start, const float4 *tsorted, uint8_t *flag, int32_t *nn_idx,
				  float *nn_d2, unsigned long long *winner, const float4 *tnrm, int32_t *match, float *wd, const float4 *tpos, int32_t *nn_hint, float4 *mq, uint32_t cap, uint32_t maxcells,
				  uint32_t *wl, uint32_t *wl_ctr, uint32_t parity, bool first)
{
	// first: the run's iteration 0 with the rigid step applied by the setup (identity_step; class-level jobs, rp.lds_dedup, no normal shooting: the caller's
	// conditions) — every called class cloud goes straight to the staged search (first_goes_direct)
	// per device (launch.h: DevLaunch): k_nn_lds may take the whole LDS; dyn_max[0] = the dynamic LDS k_cert_nn can have next to its static block
	const DevLaunch D = dev_launch<0>([](DevLaunch &d) {
		if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_nn_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
			return false;
		hipFuncAttributes fa;
		d.dyn_max[0] = 1;
		d.dyn_max[1] = 0; // static LDS of the light pass with the 768-entry leftover list (0: unknown — not used)
		if (hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_cert<MULLS_CERT_BLOCK, 3, 768>)) == hipSuccess)
			d.dyn_max[1] = fa.sharedSizeBytes;
		if (hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_cert_nn)) == hipSuccess && fa.sharedSizeBytes < 160u * 1024u)
		{
			const size_t room = 160u * 1024u - fa.sharedSizeBytes;
			if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_cert_nn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)room) == hipSuccess)
				d.dyn_max[0] = room;
		}
		return true;
	});
	if (!D.ok)
		return -1;
	const uint32_t n_cu = D.n_cu;
	const size_t fused_dyn_max = D.dyn_max[0];
	if (!njobs)
		return 0;
	const bool dedup = rp.lds_dedup != 0u;
	// (up to two class clouds per CU: 117 k vs 103 k registrations/s at 192 pairs, 131 k vs 119 k at 256; no gain beyond, profiles/r03_modes_fused.txt)
	if (njobs <= 2u * n_cu && rp.debug_stop != 10u && fused_lds_bytes(cap, maxcells, dedup) <= fused_dyn_max)
	{
		// light and heavy pass in one launch
		hipLaunchKernelGGL(k_cert_nn, dim3(njobs), dim3(MULLS_LDS_BLOCK), fused_lds_bytes(cap, maxcells, dedup), st, jobs, descs, states, rp, spos, snrm, grids, cell_start,
						   tsorted, flag, nn_idx, nn_d2, winner, tnrm, match, wd, tpos, nn_hint, mq, cap, first ? 1u : 0u);
		return 0;
	}
	if (first)
		; // no light pass in iteration 0 (first_goes_direct; the next iteration's queue counters are as prepare_run cleared them)
	else if (njobs <= 2u * n_cu)
		hipLaunchKernelGGL(k_cert<1024>, dim3(njobs), dim3(1024), dedup ? ((size_t)cap * 2u + 3u) & ~(size_t)3 : 0u, st, jobs, descs, states, rp, spos, snrm, grids, cell_start, tsorted,
						   flag, nn_idx, nn_d2, winner, tnrm, match, wd, tpos, nn_hint, mq, wl, wl_ctr, parity & 1u);
	else if (dedup && MULLS_CERT_BLOCK == 512 && D.dyn_max[1] && (((size_t)cap * 2u + 3u) & ~(size_t)3) + D.dyn_max[1] <= 40u * 1024u)
		// (four workgroups per CU with the longer leftover list: dyn_max[1] = the 768-entry kernel's static LDS)
		hipLaunchKernelGGL((k_cert<MULLS_CERT_BLOCK, 3, 768>), dim3(njobs), dim3(MULLS_CERT_BLOCK), ((size_t)cap * 2u + 3u) & ~(size_t)3, st, jobs, descs, states, rp, spos, snrm, grids,
						   cell_start, tsorted, flag, nn_idx, nn_d2, winner, tnrm, match, wd, tpos, nn_hint, mq, wl, wl_ctr, parity & 1u);
	else
		hipLaunchKernelGGL(k_cert<MULLS_CERT_BLOCK>, dim3(njobs), dim3(MULLS_CERT_BLOCK), dedup ? ((size_t)cap * 2u + 3u) & ~(size_t)3 : 0u, st, jobs, descs, states, rp, spos, snrm, grids,
						   cell_start, tsorted, flag, nn_idx, nn_d2, winner, tnrm, match, wd, tpos, nn_hint, mq, wl, wl_ctr, parity & 1u);
	// one workgroup per CU is all the LDS allows: they take the queued class clouds by ticket
	hipLaunchKernelGGL(k_nn_lds, dim3(njobs < n_cu ? njobs : n_cu), dim3(MULLS_LDS_BLOCK), nn_lds_bytes(cap, maxcells, dedup), st, jobs, descs, states, rp, spos, snrm,
					   grids, cell_start, tsorted, flag, nn_idx, nn_d2, winner, tnrm, match, wd, tpos, nn_hint, mq, cap, wl, wl_ctr, parity & 1u, first ? njobs : 0u);
	return 0;
}

// both tiers' class clouds of a small mixed batch in one launch (k_cert_mixed): 1 = launched, 0 = not applicable (the caller launches the tiers one after the other)
int launch_cert_mixed(hipStream_t st, uint32_t n_lds, const Job *cjobs, uint32_t n_big, const Job *bjobs, uint32_t max_wgs, uint32_t rounds, CloudDesc *descs, const PairState *states,
					  const RunParams &rp, float4 *spos, float4 *snrm, const GridDesc *grids, const uint32_t *cell_start, const unsigned long long *bm, const uint32_t *pf,
					  const uint32_t *cs, const float4 *tsorted, uint8_t *flag, int32_t *nn_idx, float *nn_d2, unsigned long long *winner, const float4 *tnrm, int32_t *match,
					  float *wd, const float4 *tpos, int32_t *nn_hint, float4 *mq, uint32_t cap, uint32_t maxcells, bool first)
{
	const DevLaunch D = dev_launch<1>([](DevLaunch &d) { // per device (launch.h): the dynamic LDS k_cert_mixed can have next to its static block, the CU count
		hipFuncAttributes fa;
		d.dyn_max[0] = 1;
		if (hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_cert_mixed)) == hipSuccess && fa.sharedSizeBytes < 160u * 1024u)
		{
			const size_t room = 160u * 1024u - fa.sharedSizeBytes;
			if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_cert_mixed), hipFuncAttributeMaxDynamicSharedMemorySize, (int)room) == hipSuccess)
				d.dyn_max[0] = room;
		}
		return true;
	});
	const size_t dyn_max = D.dyn_max[0];
	const uint32_t n_cu = D.n_cu;
	if (!n_lds || !n_big || rp.normal_shooting || rp.debug_stop == 10u || rp.debug_stop == 22u)
		return 0;
	uint32_t split = 1;
	while (split < 16 && n_big * split * 2u <= max_wgs)
		split <<= 1;
	// one workgroup per CU (each holds the whole LDS): only while every workgroup of both tiers is resident at once — or, while the chunk-level jobs still search
	// (`rounds` > 1: the run's first iterations), in up to that many rounds: the LDS tier's workgroups come first and stay for the length of their staged search,
	// the global-memory tier's jobs are shared by more workgroups, which follow each other on the CUs that are left (64 scans against 20 000-point maps: iteration 0
	// 129 -> 8x us, profiles/r06_experiments.txt item 24)
	if (n_lds + n_big > n_cu)
		rounds = 1u; // (only batches that take this launch in every iteration)
	while (split > 1 && n_lds + n_big * split > n_cu * rounds)
		split >>= 1;
	const bool dedup = rp.lds_dedup != 0u;
	const size_t dyn = fused_lds_bytes(cap, maxcells, dedup) > sizeof(BigLds<MULLS_BIG_BLOCK * 3>) ? fused_lds_bytes(cap, maxcells, dedup) : sizeof(BigLds<MULLS_BIG_BLOCK * 3>);
	if (n_lds + n_big * split > n_cu * rounds || n_lds > n_cu || dyn > dyn_max)
		return 0;
	hipLaunchKernelGGL(k_cert_mixed, dim3(n_lds + n_big * split), dim3(MULLS_LDS_BLOCK), dyn, st, n_lds, cjobs, bjobs, split, descs, states, rp, spos, snrm, grids, cell_start, bm, pf, cs,
					   tsorted, flag, nn_idx, nn_d2, winner, tnrm, match, wd, tpos, nn_hint, mq, cap, first ? 1u : 0u);
	return 1;
}

void launch_cert_big(hipStream_t st, uint32_t njobs, const Job *jobs, uint32_t max_wgs, CloudDesc *descs, const PairState *states, const RunParams &rp, float4 *spos, float4 *snrm,
					 const GridDesc *grids, const unsigned long long *bm, const uint32_t *pf, const uint32_t *cs, const float4 *tsorted, uint8_t *flag, int32_t *nn_idx,
					 float *nn_d2, unsigned long long *winner, const float4 *tpos, const float4 *tnrm, int32_t *nn_hint, int32_t *match, float *wd, float4 *mq)
{
	if (!njobs)
		return;
	// few chunk-level jobs: several workgroups share a 512-query job so that the chip is not left idle — up to max_wgs workgroups.  Splitting pays while the
	// jobs still search (first iterations: the caller allows four rounds of resident workgroups); once every point certifies a split launch is only more
	// workgroups for the same walk (a 236 k-point pair: 1860 workgroups took 30 us where 465 take 12), so later launches stop at one round
	uint32_t split = 1;
	while (split < 16 && njobs * split * 2u <= max_wgs)
		split <<= 1;
	hipLaunchKernelGGL(k_cert_big, dim3(njobs * split), dim3(MULLS_BIG_BLOCK), 0, st, jobs, descs, states, rp, spos, snrm, grids, bm, pf, cs, tsorted, flag, nn_idx, nn_d2, winner,
					   split, tpos, tnrm, nn_hint, match, wd, mq);
}

void launch_nn(hipStream_t st, uint32_t njobs, const Job *jobs, CloudDesc *descs, const PairState *states, const RunParams &rp, float4 *spos,
			   float4 *snrm, const float4 *tpos, const uint8_t *flag, int32_t *nn_idx, float *nn_d2, unsigned long long *winner)
{
	if (njobs)
		hipLaunchKernelGGL(k_nn, dim3(njobs), dim3(MULLS_NN_BLOCK), 0, st, jobs, descs, states, rp, spos, snrm, tpos, flag, nn_idx, nn_d2, winner);
}

void launch_nn_shoot(hipStream_t st, uint32_t njobs, const Job *jobs, CloudDesc *descs, const PairState *states, const RunParams &rp,
					 float4 *spos, float4 *snrm, const float4 *tpos, const uint8_t *flag, int32_t *nn_idx, float *nn_d2, unsigned long long *winner)
{
	if (njobs)
		hipLaunchKernelGGL(k_nn_shoot, dim3(njobs), dim3(MULLS_BLOCK), 0, st, jobs, descs, states, rp, spos, snrm, tpos, flag, nn_idx, nn_d2,
						   winner);
}

void launch_filter(hipStream_t st, uint32_t njobs, const Job *jobs, CloudDesc *descs, const PairState *states, const RunParams &rp,
				   const float4 *snrm, const float4 *tnrm, uint8_t *flag, const int32_t *nn_idx, const float *nn_d2, int32_t *match, float *wd,
				   const unsigned long long *winner, const float4 *tpos, float4 *mq, bool big)
{
	if (njobs)
		hipLaunchKernelGGL(k_filter, dim3(njobs), dim3(MULLS_BLOCK), 0, st, jobs, descs, states, rp, snrm, tnrm, flag, nn_idx, nn_d2, match, wd,
						   winner, tpos, mq, big ? 1u : 0u);
}
