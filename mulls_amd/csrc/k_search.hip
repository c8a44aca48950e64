// k_search.hip — correspondence search (three exact tiers + normal shooting) and the rejection chain
// (gfx950 / CDNA4, wave64; numerics policy and launch geometry: device_util.h)
#include "device_util.h"

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

// Correspondence search, global-memory grid tier (target class clouds too large for LDS).  Phase 1: the workgroup applies
// this iteration's rigid step to its slice of a 512-point job (coalesced 16-B traffic, double math once per point) and
// parks the transformed positions in LDS, together with the distance to the target the point found in the previous iteration
// (an exact upper bound, as in k_nn_lds).  Phase 2: a 16-lane sub-group owns one query at a time: the cube of that bound —
// or, without one, the own cell and then the cube of min(cell edge, distance found) —, then — while nothing lies inside the probed radius — one last cube of the distance
// found, or cubes of twice the radius up to the rejection radius; rows are swept 16 at a time with coalesced candidate
// loads, 4 xor-shuffles reduce (distance, index).  `split` workgroups share one job (batches with few jobs would leave
// most CUs idle otherwise).  Outputs are identical to k_nn.
__global__ __launch_bounds__(MULLS_BLOCK) void k_nn_grid(const Job *__restrict__ jobs, CloudDesc *__restrict__ descs,
														  const PairState *__restrict__ states, RunParams rp, float4 *__restrict__ spos,
														  float4 *__restrict__ snrm, const GridDesc *__restrict__ grids,
														  const unsigned long long *__restrict__ bm, const uint32_t *__restrict__ pf,
														  const uint32_t *__restrict__ cs, const float4 *__restrict__ tsorted,
														  const uint8_t *__restrict__ flag, int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2,
														  unsigned long long *__restrict__ winner, uint32_t split, const float4 *__restrict__ tpos,
														  int32_t *__restrict__ nn_hint, const int32_t *__restrict__ match, const float4 *__restrict__ mq)
{
	__shared__ float4 qpos[MULLS_SRC_PER_BLOCK]; // transformed query positions; w = upper bound on the squared NN distance (+inf: none), -1 = dead / out of range
	const uint32_t wg = xcd_job(blockIdx.x, gridDim.x);
	const Job job = jobs[wg / split];
	const uint32_t per = MULLS_SRC_PER_BLOCK / split, q0 = (wg % split) * per; // this workgroup's queries of the job: [q0, q0 + per)
	const PairState &ps = states[job.pair];
	if (!ps.active || (rp.normal_shooting && (job.cls == 0 || job.cls == 2 || job.cls == 4)))
		return;
	const uint32_t ci = job.pair * MULLS_NC + job.cls;
	CloudDesc &d = descs[ci];
	const uint32_t src_n = d.src_n, alive_cur = d.alive_cur;
	const bool called = class_called(rp, d, job.cls);
	// temporal coherence, as in k_nn_lds: the target a point found in the previous iteration bounds this iteration's search
	// exactly (any target is an upper bound).  Here every probe is a chain of dependent global loads (occupancy word -> rank /
	// start -> candidates), so starting with the cube of that radius instead of the own-cell probe saves a whole chain.
	const bool use_hint = called && ps.iter > 0;
	const uint32_t tgt_n = d.tgt_n;
	{
		const double *T = ps.T;
		for (uint32_t k = threadIdx.x; k < per; k += MULLS_BLOCK)
		{
			const uint32_t s = job.start + q0 + k;
			float4 out = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
			if (s < src_n && (flag[d.src_off + s] & MULLS_F_ALIVE))
			{
				// pcl::transformPointCloudWithNormals<PointT,double> (cregistration.hpp:1690-1695)
				const float4 p = spos[d.src_off + s], n = snrm[d.src_off + s];
				const double x = p.x, y = p.y, z = p.z, nx = n.x, ny = n.y, nz = n.z;
				out.x = (float)(T[0] * x + T[1] * y + T[2] * z + T[3]);
				out.y = (float)(T[4] * x + T[5] * y + T[6] * z + T[7]);
				out.z = (float)(T[8] * x + T[9] * y + T[10] * z + T[11]);
				out.w = __builtin_inff();
				if (use_hint)
				{
					const uint32_t h = (uint32_t)nn_hint[d.src_off + s];
					if (h < tgt_n)
					{
						// a point whose hint is its standing correspondence carries that target's position in its own record
						const float4 t = (int32_t)h == match[d.src_off + s] ? mq[2u * (d.src_off + s)] : tpos[d.tgt_off + h];
						const float dx = out.x - t.x, dy = out.y - t.y, dz = out.z - t.z;
						const float d0 = (dx * dx + dy * dy) + dz * dz;
						if (d0 >= 0.0f)
							out.w = d0;
					}
				}
				const float onx = (float)(T[0] * nx + T[1] * ny + T[2] * nz);
				const float ony = (float)(T[4] * nx + T[5] * ny + T[6] * nz);
				const float onz = (float)(T[8] * nx + T[9] * ny + T[10] * nz);
				spos[d.src_off + s] = make_float4(out.x, out.y, out.z, p.w);
				snrm[d.src_off + s] = make_float4(onx, ony, onz, n.w);
			}
			qpos[k] = out;
		}
	}
	if (!called)
		return; // correspondences of the previous iteration stay in force
	__syncthreads();

	const GridDesc g = grids[ci];
	const BmGrid B = {bm + g.cell_off, pf + g.cell_off, cs + d.tgt_off + ci};
	const float4 *__restrict__ ts = tsorted + d.tgt_off;
	const float r = 2.5f * ps.thr[job.cls]; // filter_dis_times * dis_thre (float), cregistration.hpp:1745
	const double maxd = (double)r;
	const double max_dist_sqr = maxd * maxd;
	const bool gate = alive_cur >= 500u;
	const unsigned long long key_hi = (unsigned long long)(0xffffffffu - (rp.tick_base + (uint32_t)ps.iter)) << 32;
	const float m = fminf(r, 0.999f * g.h - 2e-4f); // first-probe radius: its margin-inflated cube spans at most 3 cells per axis
	const uint32_t sub = threadIdx.x & (MULLS_GRID_GROUP - 1u), grp = threadIdx.x / MULLS_GRID_GROUP;
	uint32_t matched_cnt = 0;
	for (uint32_t k = grp; k < per; k += MULLS_BLOCK / MULLS_GRID_GROUP)
	{
		const uint32_t s = job.start + q0 + k;
		if (s >= src_n)
			break;
		const float4 q = qpos[k];
		if (q.w < 0.0f)
			continue;
		float best = __builtin_inff();
		int bi = -1;
		if (q.w < __builtin_inff())
		{
			// bounded by last iteration's correspondence: one sweep of the cube of that radius (it contains that target)
			grid_scan_box(g, B, ts, q.x, q.y, q.z, fminf(m, sqrtf(q.w)), sub, best, bi);
			group_min(best, bi);
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
					{
						const float dx = q.x - c[w].x, dy = q.y - c[w].y, dz = q.z - c[w].z;
						const float dist = (dx * dx + dy * dy) + dz * dz;
						const int idx = __float_as_int(c[w].w);
						if (dist < best || (dist == best && idx < bi))
						{
							best = dist;
							bi = idx;
						}
					}
			}
		}
		group_min(best, bi);
		// probe 1: the cells within min(first-probe radius, current best distance); skipped when that box is the own cell
		{
			const float R1 = bi >= 0 ? fminf(m, sqrtf(best)) : m;
			const float Rm = R1 * 1.0001f + 1e-4f;
			const bool own_only = grid_cell(q.x - Rm, g.ox, g.inv_h, g.nx) == ocx && grid_cell(q.x + Rm, g.ox, g.inv_h, g.nx) == ocx &&
								  grid_cell(q.y - Rm, g.oy, g.inv_h, g.ny) == ocy && grid_cell(q.y + Rm, g.oy, g.inv_h, g.ny) == ocy &&
								  grid_cell(q.z - Rm, g.oz, g.inv_h, g.nz) == ocz && grid_cell(q.z + Rm, g.oz, g.inv_h, g.nz) == ocz;
			if (!own_only)
			{
				grid_scan_box(g, B, ts, q.x, q.y, q.z, R1, sub, best, bi);
				group_min(best, bi);
			}
		}
		}
		// nothing inside the probed radius yet: one last probe at the distance found, else double the radius (up to r)
		float Rc = m;
		while (!(bi >= 0 && best <= Rc * Rc) && Rc < r)
		{
			const bool last = bi >= 0;
			Rc = last ? fminf(r, sqrtf(best)) : fminf(r, 2.0f * Rc);
			grid_scan_box(g, B, ts, q.x, q.y, q.z, Rc, sub, best, bi);
			group_min(best, bi);
			if (last)
				break;
		}
		if (sub == 0)
		{
			const bool matched = bi >= 0 && !((double)best > max_dist_sqr);
			nn_idx[d.src_off + s] = matched ? bi : -1;
			nn_d2[d.src_off + s] = best;
			nn_hint[d.src_off + s] = bi;
			if (matched)
			{
				matched_cnt++;
				if (gate)
					atomicMin(&winner[d.tgt_off + bi], key_hi | (unsigned long long)s);
			}
		}
	}
	for (int off = 32; off > 0; off >>= 1)
		matched_cnt += __shfl_down(matched_cnt, off);
	if ((threadIdx.x & 63) == 0 && matched_cnt)
		atomicAdd(&d.n_matched, matched_cnt);
}

// ---------------------------------------------------------------------------------------------------------------
// Rejection chain of determine_corres after the search (cregistration.hpp:1755-1830), one source point.
// `dedup_done`: the duplicate rule has been applied already (losers carry nn_idx = -1, k_nn_lds with rp.lds_dedup).
struct FilterCtx
{
	bool gate, any_match, normal_check, dedup_done;
	float max_sqr;	 // CorrespondenceRejectorDistance::setMaximumDistance (float)
	double cos_thre; // cos(angle_thre_degree / 180.0 * M_PI), evaluated on the host (:1818)
	unsigned long long key_hi;
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
			valid = !(dist > F.max_sqr);
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
					n2 = tnrm[d.tgt_off + m];
					mq[2u * g] = tpos[d.tgt_off + m];
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

__global__ __launch_bounds__(MULLS_BLOCK) void k_filter(const Job *__restrict__ jobs, CloudDesc *__restrict__ descs,
														 const PairState *__restrict__ states, RunParams rp,
														 const float4 *__restrict__ snrm, const float4 *__restrict__ tnrm, uint8_t *__restrict__ flag,
														 const int32_t *__restrict__ nn_idx, const float *__restrict__ nn_d2,
														 int32_t *__restrict__ match, float *__restrict__ wd,
														 const unsigned long long *__restrict__ winner, const float4 *__restrict__ tpos,
														 float4 *__restrict__ mq)
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
	const FilterCtx F = {d.alive_cur >= 500u, d.n_matched > 0u, job.cls != 5, rp.lds_dedup != 0u, thr * thr, rp.cos_bearing,
						 (unsigned long long)(0xffffffffu - (rp.tick_base + (uint32_t)ps.iter)) << 32};
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
__device__ __forceinline__ void row16_min(nnkey &bk)
{
	dpp_min_step<0xB1>(bk);	 // quad_perm [1,0,3,2]
	dpp_min_step<0x4E>(bk);	 // quad_perm [2,3,0,1]
#if MULLS_LDS_GROUP >= 8
	dpp_min_step<0x141>(bk); // row_half_mirror: lanes i <-> 7 - i of each 8-lane half
#endif
#if MULLS_LDS_GROUP == 16
	dpp_min_step<0x140>(bk); // row_mirror: lanes i <-> 15 - i
#endif
}
// (best key, second-best distance) over the lanes of a sub-group.  Every lane enters with the best key and the second-smallest
// distance among ITS candidates; a lane whose best lost the sub-group minimum contributes that best's distance instead.
template <int CTRL>
__device__ __forceinline__ void dpp_fmin_step(float &v)
{
	const float o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
	v = fminf(o, v);
}
__device__ __forceinline__ void row16_min2(nnkey &bk, float &sec)
{
	nnkey g = bk;
	row16_min(g);
	float c = (bk == g) ? sec : key_dist(bk);
	dpp_fmin_step<0xB1>(c);
	dpp_fmin_step<0x4E>(c);
#if MULLS_LDS_GROUP >= 8
	dpp_fmin_step<0x141>(c);
#endif
#if MULLS_LDS_GROUP == 16
	dpp_fmin_step<0x140>(c);
#endif
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

// Evaluate every target in the cells intersecting the cube [p - R, p + R] (same exactness argument as grid_scan_box).  The
// rows (x-runs of cells, contiguous in the sorted cloud) are taken two at a time: every lane of the sub-group reads their
// bounds (same addresses: LDS broadcast), the candidate ranges are laid end to end and the sub-group strides over the
// concatenation, two candidates per lane in flight — the trip count is that of the total, not the sum of the per-row
// round-ups, and the only per-row work is two table reads and a running sum.  Returns false when the cube lies inside the
// query's own cell and `own_done` says that cell has been swept already.
#define MULLS_LDS_CHUNK 2
template <class G>
__device__ __forceinline__ bool lds_scan_box(const GridDesc &g, const G &L, float px, float py, float pz, float R, uint32_t sub, nnkey &bk,
											  float &sec, uint32_t &trips, bool own_done = false)
{
	const float Rm = R * 1.0001f + 1e-4f;
	const uint32_t x0 = (uint32_t)grid_cell(px - Rm, g.ox, g.inv_h, g.nx), x1 = (uint32_t)grid_cell(px + Rm, g.ox, g.inv_h, g.nx) + 1u;
	const int y0 = grid_cell(py - Rm, g.oy, g.inv_h, g.ny), y1 = grid_cell(py + Rm, g.oy, g.inv_h, g.ny);
	const int z0 = grid_cell(pz - Rm, g.oz, g.inv_h, g.nz), z1 = grid_cell(pz + Rm, g.oz, g.inv_h, g.nz);
	if (own_done && x1 - x0 == 1u && y0 == y1 && z0 == z1)
		return false; // the cube stays inside the query's own cell, which has been swept already
	int cy = y0, cz = z0;
	while (cz <= z1)
	{
		uint32_t lo[MULLS_LDS_CHUNK], pre[MULLS_LDS_CHUNK], acc = 0;
#pragma unroll
		for (int jj = 0; jj < MULLS_LDS_CHUNK; jj++)
		{
			const bool valid = cz <= z1;
			const uint32_t row = ((uint32_t)(valid ? cz : z0) * g.ny + (uint32_t)cy) * g.nx;
			const uint32_t a = L.cs(row + x0), e = L.cs(row + x1);
			lo[jj] = a - acc; // candidate f of the concatenation lives at lo[jj] + f while f < pre[jj]
			acc += valid ? e - a : 0u;
			pre[jj] = acc;
			if (++cy > y1)
			{
				cy = y0;
				cz++;
			}
		}
		trips += (acc + MULLS_LDS_GROUP - 1u) / MULLS_LDS_GROUP;
		for (uint32_t f = sub; f < acc; f += 2 * MULLS_LDS_GROUP)
		{
			const uint32_t f2 = f + MULLS_LDS_GROUP;
			const bool ok2 = f2 < acc;
			const uint32_t ff = ok2 ? f2 : f;
#if MULLS_LDS_CHUNK == 1
			const uint32_t ta = f + lo[0], tb = ff + lo[0];
#else
			const uint32_t ta = f + (f < pre[0] ? lo[0] : lo[1]);
			const uint32_t tb = ff + (ff < pre[0] ? lo[0] : lo[1]);
#endif
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
template <class G>
__device__ __forceinline__ void search_query(const GridDesc &g, const G &L, const float4 q, float r, float m, uint32_t sub, nnkey &bk, float &sec,
											  float &Rfin, uint32_t &trips)
{
	bk = NNKEY_NONE;
	sec = __builtin_inff();
	Rfin = 0.0f;
	trips = 0u;
	// One sweep of the cube of radius R.  Its cells are a superset of every earlier sweep's cells, so its own (best, second)
	// pair replaces the standing one; only when the radius was clipped to the rejection radius can the standing best lie
	// outside — it stays the answer then, and nothing is claimed about the other targets.
	auto sweep = [&](float R, bool own_done) {
		nnkey lk = NNKEY_NONE;
		float ls = __builtin_inff();
		if (lds_scan_box(g, L, q.x, q.y, q.z, R, sub, lk, ls, trips, own_done))
		{
			row16_min2(lk, ls);
			if (bk < lk)
				sec = 0.0f;
			else
			{
				bk = lk;
				sec = ls;
			}
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
		trips += (hi - lo + MULLS_LDS_GROUP - 1u) / MULLS_LDS_GROUP;
		for (uint32_t t = lo + sub; t < hi; t += 2 * MULLS_LDS_GROUP) // two candidates in flight per lane and trip
		{
			const uint32_t t2 = t + MULLS_LDS_GROUP;
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
		row16_min2(bk, sec);
		// probe 1: every cell within min(first-probe radius, current best distance) of the query
		sweep(key_found(bk) ? fminf(m, sqrtf(key_dist(bk))) : m, true);
	}
	if (!(key_found(bk) && key_dist(bk) <= m * m))
		sweep(key_found(bk) ? fminf(r, sqrtf(key_dist(bk))) : r, false); // nothing within the first-probe radius: widen to the best distance, or to the rejection radius
}

// what the search of one class cloud needs to know about its iteration
struct ClassCtx
{
	float r, m;			 // rejection radius 2.5 * thr (filter_dis_times * dis_thre, cregistration.hpp:1745) and first-probe radius
	double max_dist_sqr; // (double)r squared: CorrespondenceEstimation's max_distance test
	bool gate, dedup;	 // >= 500 live source points: duplicate rule in force; ... and resolved in this workgroup's LDS table
	unsigned long long key_hi;
};
__device__ __forceinline__ ClassCtx class_ctx(const RunParams &rp, const PairState &ps, const GridDesc &g, int cls, uint32_t alive_cur, bool called)
{
	ClassCtx C;
	C.r = 2.5f * ps.thr[cls];
	const double maxd = (double)C.r;
	C.max_dist_sqr = maxd * maxd;
	C.m = fminf(C.r, 0.999f * g.h - 2e-4f); // first-probe radius: its margin-inflated cube spans at most 3 cells per axis
	C.gate = alive_cur >= 500u;
	C.dedup = rp.lds_dedup != 0u && called && C.gate;
	C.key_hi = (unsigned long long)(0xffffffffu - (rp.tick_base + (uint32_t)ps.iter)) << 32;
	return C;
}

// lane `sub == 0` of a sub-group commits the result of a searched query; returns whether it is a match
__device__ __forceinline__ bool commit_search(const ClassCtx &C, const CloudDesc &d, uint32_t s, nnkey bk, float sec, float Rfin, uint32_t trips,
											   int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2, int2 *__restrict__ hint2, uint32_t *W,
											   unsigned long long *__restrict__ winner)
{
	const float best = key_dist(bk);
	const int bi = (int)(uint32_t)bk; // -1: nothing found
	const bool matched = bi >= 0 && !((double)best > C.max_dist_sqr);
	nn_idx[d.src_off + s] = matched ? bi : -1;
	nn_d2[d.src_off + s] = best;
	// hint and cost class of the next iteration; every target but the one found is at least min(second, radius swept) away
	hint2[d.src_off + s] = make_int2((int32_t)(((uint32_t)bi & 0xffffu) | (min(trips, 31u) << 16)), __float_as_int(fminf(sqrtf(sec), Rfin)));
	if (matched)
	{
		if (C.dedup)
			atomicMin(&W[bi], s); // this workgroup sees every query of the class cloud: the duplicate table stays on chip
		else if (C.gate)
			atomicMin(&winner[d.tgt_off + bi], C.key_hi | (unsigned long long)s);
	}
	return matched;
}

// End of a class cloud's iteration, by the workgroup that holds all of its correspondences (k_cert when it searched the few
// leftovers itself, else k_nn_lds): duplicate rule, then the rejection chain (k_filter's work) on the results while they are
// still in cache, and the class's counters.  `red`: 3 * (BLK / 64) words of LDS.  Chunk-level jobs (no rp.lds_dedup) only
// add their matches to the class counter; k_filter does the rest.
template <int BLK>
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
				if (m >= 0 && W[m] != s)
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
	const FilterCtx F = {C.gate, total_matched > 0u, job.cls != 5, true, thr * thr, rp.cos_bearing, C.key_hi};
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
} // namespace

// nn_idx value of a live point that k_cert could not certify and left to k_nn_lds (its sweep radius waits in nn_d2)
#define MULLS_NEEDS_SEARCH (-2)

__global__ __launch_bounds__(MULLS_CERT_BLOCK) void k_cert(const Job *__restrict__ cjobs, CloudDesc *__restrict__ descs,
															const PairState *__restrict__ states, RunParams rp, float4 *__restrict__ spos,
															float4 *__restrict__ snrm, const GridDesc *__restrict__ grids,
															const uint32_t *__restrict__ cell_start, const float4 *__restrict__ tsorted, uint8_t *flag,
															int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2, unsigned long long *__restrict__ winner,
															const float4 *__restrict__ tnrm, int32_t *__restrict__ match, float *__restrict__ wd,
															const float4 *__restrict__ tpos, int32_t *__restrict__ nn_hint, float4 *__restrict__ mq,
															uint32_t *__restrict__ wl, uint32_t *__restrict__ wl_ctr, uint32_t parity)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	uint32_t *W = reinterpret_cast<uint32_t *>(lds_raw); // [cap] lowest source index matched to each target (lds_dedup)
	__shared__ float4 uq[MULLS_CERT_SMALL];				  // the few queries this workgroup searches itself
	__shared__ uint32_t us[MULLS_CERT_SMALL];
	__shared__ uint32_t ucount, red[3 * (MULLS_CERT_BLOCK / 64)];
	int2 *__restrict__ hint2 = reinterpret_cast<int2 *>(nn_hint); // per source point: (hint word, bound on every OTHER target's distance)

	if (blockIdx.x == 0 && threadIdx.x == 0)
		wl_ctr[2u * (parity ^ 1u)] = wl_ctr[2u * (parity ^ 1u) + 1u] = 0u; // the next iteration's queue (this one's predecessor has been drained)
	const Job job = cjobs[blockIdx.x];
	const PairState &ps = states[job.pair];
	if (!ps.active || (rp.normal_shooting && (job.cls == 0 || job.cls == 2 || job.cls == 4)))
		return;
	CloudDesc &d = descs[job.pair * MULLS_NC + job.cls];
	const uint32_t src_n = d.src_n, tgt_n = d.tgt_n;
	const bool called = class_called(rp, d, job.cls);
	const GridDesc g = grids[job.pair * MULLS_NC + job.cls];
	const ClassCtx C = class_ctx(rp, ps, g, job.cls, d.alive_cur, called);
	const bool have_prev = ps.iter > 0; // hint records of this run exist from its second iteration on
	const bool use_hint = called && have_prev;
	const uint32_t q_end = min(src_n, job.start + (job.count ? job.count : (uint32_t)MULLS_SRC_PER_BLOCK));
	if (C.dedup)
		for (uint32_t t = threadIdx.x; t < tgt_n; t += MULLS_CERT_BLOCK)
			W[t] = 0xffffffffu;
	if (threadIdx.x == 0)
		ucount = 0u;
	__syncthreads();

	uint32_t matched_cnt = 0;
	for (uint32_t s = job.start + threadIdx.x; s < q_end; s += MULLS_CERT_BLOCK)
	{
		const uint32_t gi = d.src_off + s;
		if (!(flag[gi] & MULLS_F_ALIVE))
			continue;
		const float4 p = spos[gi], n = snrm[gi];
		uint32_t hv = 0xffffu;
		float lb = 0.0f;
		int32_t pm = -1;
		if (have_prev)
		{
			const int2 h = hint2[gi];
			lb = __int_as_float(h.y);
			if (use_hint)
			{
				hv = (uint32_t)h.x;
				pm = match[gi];
			}
		}
		// the hinted target's position: for a point whose hint is its standing correspondence it sits in the point's own
		// record (coalesced), otherwise it is gathered
		const uint32_t hj = hv & 0xffffu;
		float4 tj = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		if (hj < tgt_n)
			tj = (int32_t)hj == pm ? mq[2u * gi] : tpos[d.tgt_off + hj];
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
							atomicMin(&W[hj], s);
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
			if (k < MULLS_CERT_SMALL)
			{
				uq[k] = out;
				us[k] = s;
			}
		}
	}
	if (!called)
		return;
	__syncthreads();
	const uint32_t U = ucount;
	if (U > MULLS_CERT_SMALL)
	{
		// too many for the global-memory walk: k_nn_lds stages the target cloud and takes the class cloud from here
		if (threadIdx.x == 0)
			wl[atomicAdd(&wl_ctr[2u * parity], 1u)] = blockIdx.x;
		// matches certified here are counted again by k_nn_lds from nn_idx; chunk-level jobs add theirs to the class counter there too
		return;
	}
	// the few leftovers against the grid where k_grid_build_sort left it (L2-resident): same sweeps, same keys
	const GlobGrid L = {tsorted + d.tgt_off, reinterpret_cast<const uint16_t *>(cell_start) + g.cell_off};
	const uint32_t sub = threadIdx.x & (MULLS_LDS_GROUP - 1u), grp = threadIdx.x / MULLS_LDS_GROUP;
	for (uint32_t i = grp; i < U; i += MULLS_CERT_BLOCK / MULLS_LDS_GROUP)
	{
		nnkey bk;
		float sec, Rfin;
		uint32_t trips;
		search_query(g, L, uq[i], C.r, C.m, sub, bk, sec, Rfin, trips);
		if (sub == 0 && commit_search(C, d, us[i], bk, sec, Rfin, trips, nn_idx, nn_d2, hint2, W, winner))
			matched_cnt++;
	}
	class_tail<MULLS_CERT_BLOCK>(rp, ps, C, d, job, q_end, matched_cnt, U, W, red, snrm, tnrm, flag, nn_idx, nn_d2, match, wd, winner, tpos, mq);
}

__global__ __launch_bounds__(MULLS_LDS_BLOCK) void k_nn_lds(const Job *__restrict__ cjobs, CloudDesc *__restrict__ descs,
															 const PairState *__restrict__ states, RunParams rp, const float4 *__restrict__ spos,
															 const float4 *__restrict__ snrm, const GridDesc *__restrict__ grids,
															 const uint32_t *__restrict__ cell_start, const float4 *__restrict__ tsorted,
															 uint8_t *flag, int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2,
															 unsigned long long *__restrict__ winner, const float4 *__restrict__ tnrm, int32_t *__restrict__ match,
															 float *__restrict__ wd, const float4 *__restrict__ tpos, int32_t *__restrict__ nn_hint, float4 *__restrict__ mq,
															 uint32_t cap, const uint32_t *__restrict__ wl, uint32_t *__restrict__ wl_ctr, uint32_t parity)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	float4 *qpos = reinterpret_cast<float4 *>(lds_raw);					  // [MULLS_LDS_QCHUNK] queries of the chunk, w = sweep radius of a hinted query / +inf
	uint32_t *HIST = reinterpret_cast<uint32_t *>(qpos + MULLS_LDS_QCHUNK); // [32] cost histogram, [32] bucket bases, [64] live queries of the chunk, [65] queue ticket
	uint16_t *ORDER = reinterpret_cast<uint16_t *>(HIST + 80);				  // [MULLS_LDS_QCHUNK] query slots, most expensive first
	float *P = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(HIST) + MULLS_LDS_AUX); // [3 * cap] x, y, z records
	uint16_t *IDX = reinterpret_cast<uint16_t *>(P + 3u * cap);			  // [cap]
	uint16_t *CS = IDX + cap;											  // [rp.grid_maxcells + 1]
	uint32_t *W = reinterpret_cast<uint32_t *>(CS + ((rp.grid_maxcells + 8u) & ~1u)); // [cap] lowest source index matched to each target (lds_dedup)
	int2 *__restrict__ hint2 = reinterpret_cast<int2 *>(nn_hint);

	// persistent workgroup: class clouds come from the queue k_cert filled (complete before this kernel starts)
	const uint32_t n_queued = wl_ctr[2u * parity];
	for (;;)
	{
		__syncthreads(); // the previous class cloud's LDS contents have been consumed
		if (threadIdx.x == 0)
			HIST[65] = atomicAdd(&wl_ctr[2u * parity + 1u], 1u);
		__syncthreads();
		const uint32_t ticket = HIST[65];
		if (ticket >= n_queued)
			return;
		const Job job = cjobs[wl[ticket]];
		const PairState &ps = states[job.pair];
		CloudDesc &d = descs[job.pair * MULLS_NC + job.cls];
		const uint32_t src_n = d.src_n, tgt_n = d.tgt_n;
		const GridDesc g = grids[job.pair * MULLS_NC + job.cls];
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
				pf_i = nn_idx[gi];
				pf_p = spos[gi];
				pf_w = nn_d2[gi];
				pf_hv = (uint32_t)hint2[gi].x;
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
		const uint32_t sub = threadIdx.x & (MULLS_LDS_GROUP - 1u), grp = threadIdx.x / MULLS_LDS_GROUP;
		uint32_t matched_cnt = 0, searched = 0;

		if (threadIdx.x < 32u)
			HIST[threadIdx.x] = 0u;
		for (uint32_t chunk = job.start; chunk < q_end; chunk += q_step)
		{
			__syncthreads(); // the previous chunk's queries have been consumed (and, first trip, the staging stores are visible below)
			uint32_t bucket = 0, rank = 0xffffffffu; // cost class of this lane's query (0 = most expensive) and its rank inside the class
			// phase 1: one source point per lane — uncertified points become the chunk's queries, certified matches enter the duplicate table
			if (threadIdx.x < MULLS_LDS_QCHUNK && (pf_f & MULLS_F_ALIVE))
			{
				if (pf_i == MULLS_NEEDS_SEARCH)
				{
					qpos[threadIdx.x] = make_float4(pf_p.x, pf_p.y, pf_p.z, pf_w);
					bucket = pf_w < __builtin_inff() ? 31u - ((pf_hv >> 16) & 31u) : 0u; // unhinted queries (the hint word is stale or absent) are the expensive ones
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
					HIST[64] = incl;
			}
			__syncthreads();
			if (rank != 0xffffffffu)
				ORDER[HIST[32u + bucket] + rank] = (uint16_t)threadIdx.x;
			__syncthreads();
			const uint32_t n_live = HIST[64];
			searched += n_live;

			// phase 2: sub-groups of MULLS_LDS_GROUP lanes, one query at a time each
			for (uint32_t i = grp; i < n_live; i += MULLS_LDS_BLOCK / MULLS_LDS_GROUP)
			{
				const uint32_t k = ORDER[i];
				nnkey bk;
				float sec, Rfin;
				uint32_t trips;
				search_query(g, L, qpos[k], C.r, C.m, sub, bk, sec, Rfin, trips);
				if (sub == 0 && commit_search(C, d, chunk + k, bk, sec, Rfin, trips, nn_idx, nn_d2, hint2, W, winner))
					matched_cnt++;
			}
		}
		uint32_t *red = reinterpret_cast<uint32_t *>(lds_raw); // the query block is free once the tail's first barrier has passed
		class_tail<MULLS_LDS_BLOCK>(rp, ps, C, d, job, q_end, matched_cnt, searched, W, red, snrm, tnrm, flag, nn_idx, nn_d2, match, wd, winner, tpos, mq);
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
				  uint32_t *wl, uint32_t *wl_ctr, uint32_t parity)
{
	static bool attr_set = false;
	static uint32_t n_cu = 256;
	if (!attr_set)
	{
		if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_nn_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
			return -1;
		int dev = 0, cus = 0;
		if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
			n_cu = (uint32_t)cus;
		attr_set = true;
	}
	if (!njobs)
		return 0;
	const bool dedup = rp.lds_dedup != 0u;
	hipLaunchKernelGGL(k_cert, dim3(njobs), dim3(MULLS_CERT_BLOCK), dedup ? (size_t)cap * 4u : 0u, st, jobs, descs, states, rp, spos, snrm, grids, cell_start, tsorted,
					   flag, nn_idx, nn_d2, winner, tnrm, match, wd, tpos, nn_hint, mq, wl, wl_ctr, parity & 1u);
	// one workgroup per CU is all the LDS allows: they take the queued class clouds by ticket
	hipLaunchKernelGGL(k_nn_lds, dim3(njobs < n_cu ? njobs : n_cu), dim3(MULLS_LDS_BLOCK), nn_lds_bytes(cap, maxcells, dedup), st, jobs, descs, states, rp, spos, snrm,
					   grids, cell_start, tsorted, flag, nn_idx, nn_d2, winner, tnrm, match, wd, tpos, nn_hint, mq, cap, wl, wl_ctr, parity & 1u);
	return 0;
}

void launch_nn_grid(hipStream_t st, uint32_t njobs, const Job *jobs, CloudDesc *descs, const PairState *states, const RunParams &rp,
					float4 *spos, float4 *snrm, const GridDesc *grids, const unsigned long long *bm, const uint32_t *pf, const uint32_t *cs,
					const float4 *tsorted, const uint8_t *flag, int32_t *nn_idx, float *nn_d2, unsigned long long *winner, const float4 *tpos,
					int32_t *nn_hint, const int32_t *match, const float4 *mq)
{
	if (!njobs)
		return;
	// few jobs (one scan against a big map): several workgroups share a 512-query job so that the chip is not left idle
	uint32_t split = 1;
	while (split < 16 && njobs * split < 1024u)
		split <<= 1;
	hipLaunchKernelGGL(k_nn_grid, dim3(njobs * split), dim3(MULLS_BLOCK), 0, st, jobs, descs, states, rp, spos, snrm, grids, bm, pf, cs, tsorted,
					   flag, nn_idx, nn_d2, winner, split, tpos, nn_hint, match, mq);
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
				   const unsigned long long *winner, const float4 *tpos, float4 *mq)
{
	if (njobs)
		hipLaunchKernelGGL(k_filter, dim3(njobs), dim3(MULLS_BLOCK), 0, st, jobs, descs, states, rp, snrm, tnrm, flag, nn_idx, nn_d2, match, wd,
						   winner, tpos, mq);
}
