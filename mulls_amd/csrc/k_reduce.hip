// k_reduce.hip — normal-equation accumulation, per-pair finish, host exchange of states / results, stage helpers
// (gfx950 / CDNA4, wave64; numerics policy and launch geometry: device_util.h)
#include "device_util.h"

// ---------------------------------------------------------------------------------------------------------------
// weight functions (cregistration.hpp:2686-2722) — float/double mix exactly as written there
namespace
{
__device__ __forceinline__ float w_dist_adaptive(float dist, int iter_num)
{
	const float unit_dist = 30.0f, b_min = 0.7f, b_max = 1.3f, b_step = 0.05f;
	float t = b_min + b_step * iter_num;
	float b_current = (t < b_max) ? t : b_max;
	float temp = (float)(b_current + (1.0 - b_current) * dist / unit_dist);
	temp = (float)((temp > 0.01) ? (double)temp : 0.01);
	return temp;
}
__device__ __forceinline__ float w_intensity(float i1, float i2)
{
	float ratio = fabsf(i1 - i2) / 255.0f;
	return (float)exp(-1.0 * ratio);
}
__device__ __forceinline__ float w_residual(float res, float thre)
{
	return (res > thre) ? ((2 * res * thre + (1 * 1 - 2 * 1) * (thre * thre)) / res / res) : 1.0f;
}
__device__ __forceinline__ int metric_of(int cls) { return (cls == 1 || cls == 3) ? 1 : (cls == 5 ? 2 : 0); }
} // namespace

// Normal-equation accumulation (active pairs) or posterior residual (pairs flagged want_residual).  27 double
// accumulators per lane -> wave64 shuffle tree -> 4-wave LDS combine -> one 27-double partial per workgroup, summed
// in fixed order by k_finish (run-to-run deterministic, unlike atomicAdd(double)).
__global__ __launch_bounds__(MULLS_ACC_BLOCK, 4) void k_accum(const Job *__restrict__ jobs, const CloudDesc *__restrict__ descs,
														const PairState *__restrict__ states, RunParams rp, const float4 *__restrict__ spos,
														const float4 *__restrict__ mq, const uint8_t *__restrict__ flag, float *__restrict__ wd,
														double *__restrict__ partial, uint32_t job_base)
{
	__shared__ double red[MULLS_ACC_BLOCK / 64][MULLS_NTERM];
	const uint32_t job_idx = xcd_job(blockIdx.x, gridDim.x);
	const Job job = jobs[job_idx];
	const PairState &ps = states[job.pair];
	if (!ps.active && !ps.want_residual)
		return;
	const CloudDesc *pd = descs + job.pair * MULLS_NC;
	const CloudDesc &d = pd[job.cls];
	const int metric = metric_of(job.cls);
	const bool residual_pass = ps.want_residual != 0;

	float class_w = 1.0f;
	if (rp.force_class_w)
		class_w = rp.class_w_value; // stage-level entry point only (mulls_stage_accumulate)
	else if (!residual_pass && rp.w_balance && (job.cls == 0 || job.cls == 4))
	{
		// w_ground = max_(0.01, z_xy * (m2 + 2*m3 - m4) / (0.0001 + 2.0*m1))   (cregistration.hpp:1886-1894)
		int cnt[MULLS_NC];
		for (int c = 0; c < MULLS_NC; c++)
			cnt[c] = (int)(class_called(rp, pd[c], c) ? pd[c].valid_next : pd[c].n_valid);
		int m1 = cnt[0] + cnt[4], m2 = cnt[2], m3 = cnt[1], m4 = cnt[3];
		double v = rp.z_xy_ratio * (m2 + 2 * m3 - m4) / (0.0001 + 2.0 * m1);
		class_w = (float)((0.01 > v) ? 0.01 : v);
	}
	const int iter_num = ps.iter;
	const bool resid_w = rp.w_resid && iter_num > rp.resid_from_iter;
	const bool dist_w = rp.w_dist, inten_w = rp.w_inten;
	const float window = metric == 0 ? rp.win_pl : (metric == 1 ? rp.win_li : rp.win_pt);

	double acc[MULLS_NTERM];
#pragma unroll
	for (int k = 0; k < MULLS_NTERM; k++)
		acc[k] = 0.0;

#pragma unroll
	for (int u = 0; u < MULLS_SRC_PER_BLOCK / MULLS_ACC_BLOCK; u++)
	{
		const uint32_t s = job.start + threadIdx.x + u * MULLS_ACC_BLOCK;
		if (s >= d.src_n)
			continue;
		const uint32_t g = d.src_off + s;
		if ((flag[g] & (MULLS_F_ALIVE | MULLS_F_VALID)) != (MULLS_F_ALIVE | MULLS_F_VALID))
			continue;
		const float4 P = spos[g], Q = mq[2u * g], N = mq[2u * g + 1u]; // the matched target's position and direction (filter_point)
		const float px = P.x, py = P.y, pz = P.z, pi = P.w;
		const float qx = Q.x, qy = Q.y, qz = Q.z, qi = Q.w;

		if (residual_pass)
		{
			const double *x = ps.x;
			const float cw = wd[g]; // pcl::Correspondence::weight — for vertex points this is still d^2
			if (metric == 0)
			{
				float ntx = N.x, nty = N.y, ntz = N.z;
				float a = ntz * py - nty * pz;
				float b = ntx * pz - ntz * px;
				float c = nty * px - ntx * py;
				float dd = ntx * qx + nty * qy + ntz * qz - ntx * px - nty * py - ntz * pz;
				float res = (float)(ntx * x[0] + nty * x[1] + ntz * x[2] + a * x[3] + b * x[4] + c * x[5] - dd);
				acc[0] += cw * res * res;
				acc[1] += 1.0;
			}
			else
			{
				float dx = px - qx, dy = py - qy, dz = pz - qz;
				double A[3][6], bb[3];
				if (metric == 1)
				{
					float vx = N.x, vy = N.y, vz = N.z;
					A[0][0] = 0;
					A[0][1] = vz;
					A[0][2] = -vy;
					A[0][3] = -vz * pz - vy * py;
					A[0][4] = vy * px;
					A[0][5] = vz * px;
					A[1][0] = -vz;
					A[1][1] = 0;
					A[1][2] = vx;
					A[1][3] = vx * py;
					A[1][4] = -vx * px - vz * pz;
					A[1][5] = vz * py;
					A[2][0] = vy;
					A[2][1] = -vx;
					A[2][2] = 0;
					A[2][3] = vx * pz;
					A[2][4] = vy * pz;
					A[2][5] = -vy * py - vx * px;
					bb[0] = -vz * dy + vy * dz;
					bb[1] = -vx * dz + vz * dx;
					bb[2] = -vy * dx + vx * dy;
				}
				else
				{
					A[0][0] = 1, A[0][1] = 0, A[0][2] = 0, A[0][3] = 0, A[0][4] = pz, A[0][5] = -py;
					A[1][0] = 0, A[1][1] = 1, A[1][2] = 0, A[1][3] = -pz, A[1][4] = 0, A[1][5] = px;
					A[2][0] = 0, A[2][1] = 0, A[2][2] = 1, A[2][3] = py, A[2][4] = -px, A[2][5] = 0;
					bb[0] = -dx, bb[1] = -dy, bb[2] = -dz;
				}
				double r[3];
				for (int k = 0; k < 3; k++)
				{
					double t = 0;
					for (int j = 0; j < 6; j++)
						t += A[k][j] * x[j];
					r[k] = t - bb[k];
				}
				acc[0] += cw * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
				acc[1] += 3.0;
			}
			continue;
		}

		const float dist = sqrtf(qx * qx + qy * qy + qz * qz);
		if (metric == 0) // pt2pl_lls_summation, cregistration.hpp:2066-2156
		{
			float ntx = N.x, nty = N.y, ntz = N.z;
			float w = class_w;
			float a = ntz * py - nty * pz;
			float b = ntx * pz - ntz * px;
			float c = nty * px - ntx * py;
			float dd = ntx * qx + nty * qy + ntz * qz - ntx * px - nty * py - ntz * pz;
			if (dist_w)
				w = w * w_dist_adaptive(dist, iter_num);
			if (resid_w)
				w = w * w_residual(fabsf(dd), window);
			if (inten_w)
				w = w * w_intensity((float)(pi + 0.0001), (float)(qi + 0.0001));
			wd[g] = w;
			acc[0] += w * ntx * ntx;
			acc[1] += w * ntx * nty;
			acc[2] += w * ntx * ntz;
			acc[3] += w * a * ntx;
			acc[4] += w * b * ntx;
			acc[5] += w * c * ntx;
			acc[6] += w * nty * nty;
			acc[7] += w * nty * ntz;
			acc[8] += w * a * nty;
			acc[9] += w * b * nty;
			acc[10] += w * c * nty;
			acc[11] += w * ntz * ntz;
			acc[12] += w * a * ntz;
			acc[13] += w * b * ntz;
			acc[14] += w * c * ntz;
			acc[15] += w * a * a;
			acc[16] += w * a * b;
			acc[17] += w * a * c;
			acc[18] += w * b * b;
			acc[19] += w * b * c;
			acc[20] += w * c * c;
			acc[21] += w * dd * ntx;
			acc[22] += w * dd * nty;
			acc[23] += w * dd * ntz;
			acc[24] += w * dd * a;
			acc[25] += w * dd * b;
			acc[26] += w * dd * c;
		}
		else if (metric == 1) // pt2li_lls_pri_direction_summation, cregistration.hpp:2160-2275
		{
			float vx = N.x, vy = N.y, vz = N.z;
			float dx = px - qx, dy = py - qy, dz = pz - qz;
			double A[3][6], bv[3];
			A[0][0] = 0;
			A[0][1] = -vz;
			A[0][2] = vy;
			A[0][3] = vy * py + vz * pz;
			A[0][4] = -vy * px;
			A[0][5] = -vz * px;
			A[1][0] = vz;
			A[1][1] = 0;
			A[1][2] = -vx;
			A[1][3] = -vx * py;
			A[1][4] = vz * pz + vx * px;
			A[1][5] = -vz * py;
			A[2][0] = -vy;
			A[2][1] = vx;
			A[2][2] = 0;
			A[2][3] = -vx * pz;
			A[2][4] = -vy * pz;
			A[2][5] = vx * px + vy * py;
			bv[0] = -vy * dz + vz * dy;
			bv[1] = -vz * dx + vx * dz;
			bv[2] = -vx * dy + vy * dx;
			float ex = (float)fabs(bv[0]), ey = (float)fabs(bv[1]), ez = (float)fabs(bv[2]);
			float ed = sqrtf(ex * ex + ey * ey + ez * ez);
			float wx = class_w;
			if (dist_w)
				wx *= w_dist_adaptive(dist, iter_num);
			if (inten_w)
				wx *= w_intensity((float)(pi + 0.0001), (float)(qi + 0.0001));
			if (resid_w)
				wx = wx * w_residual(ed, window);
			wd[g] = wx;
			const double sw = (double)sqrtf(wx);
			for (int r = 0; r < 3; r++)
			{
				for (int c = 0; c < 6; c++)
					A[r][c] = sw * A[r][c];
				bv[r] = sw * bv[r];
			}
			int k = 0;
#pragma unroll
			for (int j = 0; j < 6; j++)
#pragma unroll
				for (int c = j; c < 6; c++)
					acc[k++] += (A[0][j] * A[0][c] + A[1][j] * A[1][c]) + A[2][j] * A[2][c];
#pragma unroll
			for (int j = 0; j < 6; j++)
				acc[21 + j] += (A[0][j] * bv[0] + A[1][j] * bv[1]) + A[2][j] * bv[2];
		}
		else // pt2pt_lls_summation, cregistration.hpp:1976-2063 (never writes the correspondence weight)
		{
			float dx = px - qx, dy = py - qy, dz = pz - qz;
			float wx = class_w, wy, wz;
			if (dist_w)
				wx = wx * w_dist_adaptive(dist, iter_num);
			if (resid_w)
				wx = wx * w_residual(sqrtf(dx * dx + dy * dy + dz * dz), window);
			if (inten_w)
				wx = wx * w_intensity((float)(pi + 0.0001), (float)(qi + 0.0001));
			wy = wx;
			wz = wx;
			if (!rp.faithful)
				wd[g] = wx; // intended behaviour: weight the vertex residual by its weight, not by d^2
			acc[0] += wx;
			acc[4] += wx * pz;
			acc[5] += (-wx * py);
			acc[6] += wy;
			acc[8] += (-wy * pz);
			acc[10] += wy * px;
			acc[11] += wz;
			acc[12] += wz * py;
			acc[13] += (-wz * px);
			acc[15] += wy * pz * pz + wz * py * py;
			acc[16] += (-wz * px * py);
			acc[17] += (-wy * px * pz);
			acc[18] += wx * pz * pz + wz * px * px;
			acc[19] += (-wx * py * pz);
			acc[20] += wx * py * py + wy * px * px;
			acc[21] += (-wx * dx);
			acc[22] += (-wy * dy);
			acc[23] += (-wz * dz);
			acc[24] += wy * pz * dy - wz * py * dz;
			acc[25] += wz * px * dz - wx * pz * dx;
			acc[26] += wx * py * dx - wy * px * dy;
		}
	}

	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
	for (int k = 0; k < MULLS_NTERM; k++)
	{
		double v = acc[k];
		for (int off = 32; off > 0; off >>= 1)
			v += __shfl_down(v, off);
		if (lane == 0)
			red[wave][k] = v;
	}
	__syncthreads();
	if (threadIdx.x < MULLS_NTERM)
		partial[(size_t)(job_base + job_idx) * MULLS_NTERM + threadIdx.x] = // job_base: first job of this sub-batch in the batch-wide table
			red[0][threadIdx.x] + red[1][threadIdx.x];
}

// ---------------------------------------------------------------------------------------------------------------
// One workgroup per pair: sum the per-job partials of every class in job order, then roll the per-class counters
// over to the next iteration.
//
namespace
{
__device__ __forceinline__ void finish_pair(CloudDesc *pd, const PairState &ps, const RunParams &rp, const double *__restrict__ partial,
											 PairOut &o, const uint32_t *__restrict__ pair_bbox)
{
	if (threadIdx.x >= 192 && threadIdx.x < 198)
		o.bbox[threadIdx.x - 192] = pair_bbox[threadIdx.x - 192];
	if (threadIdx.x < MULLS_NC * MULLS_NTERM)
	{
		const int c = threadIdx.x / MULLS_NTERM, t = threadIdx.x % MULLS_NTERM;
		if (rp.used[c]) // unused classes contribute nothing: their slots are not even sent over PCIe
		{
			double sum = 0.0;
			for (uint32_t j = pd[c].job_begin; j < pd[c].job_end; j++)
				sum += partial[(size_t)j * MULLS_NTERM + t];
			o.sums[c][t] = sum;
		}
	}
	__syncthreads();
	if (rp.pull_comb && threadIdx.x >= 64 && threadIdx.x < 64 + MULLS_NTERM)
	{
		// The 6x6 the reference inverts: pt2pl / pt2pt wrote the lower triangle, pt2li the upper one, then the mirror copies
		// lower -> upper (cregistration.hpp:1924-1938).  Class order of the += chain on shared slots: ground, facade, roof (pl),
		// pillar, beam (li), vertex (pt) (:1914-1921).  Same additions in the same order as the host did them from the class
		// rows; only this row crosses PCIe (224 B instead of 224 B per used class).
		const int order[MULLS_NC] = {0, 2, 4, 1, 3, 5};
		const int t = (int)threadIdx.x - 64;
		double val;
		if (ps.want_residual)
		{
			// get_multi_metrics_lls_residual: [0] = sum of the weighted squared residuals, [1] = number of observations
			val = 0.0;
			if (t < 2)
				for (int i = 0; i < MULLS_NC; i++)
					if (rp.used[order[i]])
						val += o.sums[order[i]][t];
		}
		else if (t < 21)
		{
			int r = 0, rem = t;
			while (rem >= 6 - r)
			{
				rem -= 6 - r;
				r++;
			}
			const bool diag = rem == 0;
			double lower = 0.0, upper = 0.0;
			for (int i = 0; i < MULLS_NC; i++)
			{
				const int cls = order[i];
				if (!rp.used[cls])
					continue;
				const double v = o.sums[cls][t];
				if (metric_of(cls) == 1 && !diag)
					upper += v;
				else
					lower += v;
			}
			val = diag ? lower : (rp.faithful ? lower : lower + upper);
		}
		else
		{
			val = 0.0;
			for (int i = 0; i < MULLS_NC; i++)
				if (rp.used[order[i]])
					val += o.sums[order[i]][t];
		}
		o.comb[t] = val;
	}
	if (threadIdx.x < MULLS_NC)
	{
		const int c = threadIdx.x;
		CloudDesc &d = pd[c];
		if (ps.active)
		{
			if (class_called(rp, d, c))
			{
				d.n_valid = d.valid_next;
				d.alive_cur = d.alive_next;
			}
			d.alive_next = 0;
			d.valid_next = 0;
			d.n_matched = 0;
		}
		o.n_valid[c] = d.n_valid;
		o.n_alive[c] = d.alive_cur;
		o.src_n[c] = d.src_n;
		o.tgt_n[c] = d.tgt_n;
	}
}
} // namespace

__global__ __launch_bounds__(MULLS_BLOCK) void k_finish(CloudDesc *__restrict__ descs, const PairState *__restrict__ states, RunParams rp,
														 const double *__restrict__ partial, PairOut *__restrict__ out, const uint32_t *__restrict__ bbox,
														 uint32_t pair_base, uint4 *__restrict__ host_words, uint32_t *__restrict__ ticket,
														 volatile uint32_t *host_epoch, uint32_t epoch)
{
	const uint32_t pair = pair_base + blockIdx.x;
	const int active = states[pair].active, want_residual = states[pair].want_residual;
	if (active || want_residual) // uniform per workgroup
		finish_pair(descs + pair * MULLS_NC, states[pair], rp, partial, out[pair], bbox + pair * 6);
	if (!host_words)
		return; // k_pull_outs packs and ships the records
	// Small batches (launch_finish decides): this workgroup ships its pair's record itself — counter block (8 uint4) + combined
	// row (14 uint4), 16-B stores into pinned host memory — and the last workgroup to arrive publishes the epoch the host
	// spins on: one launch less per iteration where the iteration is launch-bound.  (With thousands of pairs the per-workgroup
	// system-scope fences make this slower than the packed k_pull_outs: 91 k vs 104 k registrations/s at 4096 pairs.)
	__syncthreads();
	const uint32_t row_words = MULLS_NTERM_PAD / 2, head_words = 8, wpp = head_words + row_words;
	if (threadIdx.x < wpp)
	{
		const uint4 *dev = reinterpret_cast<const uint4 *>(&out[pair]);
		const uint32_t src = threadIdx.x < head_words ? (MULLS_NC + 1u) * row_words + threadIdx.x : MULLS_NC * row_words + (threadIdx.x - head_words);
		host_words[(size_t)pair * wpp + threadIdx.x] = dev[src];
	}
	__threadfence_system();
	__syncthreads();
	if (threadIdx.x == 0)
	{
		const uint32_t t = atomicAdd(ticket, 1u);
		if (t == gridDim.x - 1u)
		{
			*ticket = 0u; // re-armed for the next launch (stream order: nobody else touches it before)
			__threadfence_system();
			*host_epoch = epoch;
		}
	}
}

static_assert(sizeof(PairOut) == (MULLS_NC + 1) * MULLS_NTERM_PAD * 8 + 128, "PairOut = class rows + combined row + one 128-B counter block");
__global__ __launch_bounds__(MULLS_BLOCK) void k_pull_outs(const uint4 *__restrict__ dev_words, uint4 *__restrict__ host_words, RunParams rp,
															uint32_t pair_base, uint32_t npairs, uint32_t *__restrict__ ticket,
															volatile uint32_t *host_epoch, uint32_t epoch)
{
	const uint32_t row_words = MULLS_NTERM_PAD / 2, head_words = 8, rec_words = sizeof(PairOut) / 16;
	uint32_t n_used = 0;
	for (int c = 0; c < MULLS_NC; c++)
		n_used += rp.used[c] ? 1u : 0u;
	const uint32_t wpp = head_words + row_words * (rp.pull_comb ? 1u : n_used);
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < npairs * wpp)
	{
		const uint32_t p = pair_base + i / wpp, w = i % wpp;
		uint32_t src = (MULLS_NC + 1u) * row_words + w; // counter block
		if (w >= head_words && rp.pull_comb)
			src = MULLS_NC * row_words + (w - head_words); // the combined row
		else if (w >= head_words)
		{
			uint32_t rank = (w - head_words) / row_words, cls = 0;
			for (uint32_t c = 0; c < MULLS_NC; c++)
				if (rp.used[c])
				{
					if (rank == 0)
					{
						cls = c;
						break;
					}
					rank--;
				}
			src = cls * row_words + (w - head_words) % row_words;
		}
		host_words[(size_t)p * wpp + w] = dev_words[(size_t)p * rec_words + src];
	}
	__threadfence_system();
	__syncthreads();
	if (threadIdx.x == 0)
	{
		const uint32_t t = atomicAdd(ticket, 1u);
		if (t == gridDim.x - 1u)
		{
			*ticket = 0u; // re-armed for the next launch (stream order: nobody else touches it before)
			__threadfence_system();
			*host_epoch = epoch;
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------
// Per-iteration input: the host writes the PairState records into pinned memory; this kernel pulls them into HBM with
// coalesced 16-B reads over PCIe (one read of the block instead of one per workgroup of every later kernel).
static_assert(sizeof(PairState) % 16 == 0, "PairState must be a whole number of uint4 words");
__global__ void k_push_states(const uint4 *__restrict__ host_words, uint4 *__restrict__ dev_words, uint32_t nwords)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < nwords)
		dev_words[i] = host_words[i];
}

// ---------------------------------------------------------------------------------------------------------------
// Stage-level helpers for the parity tests (mulls_stage_* in include/mulls_hip.h)
__global__ void k_transform_aos(float4 *__restrict__ recs, uint32_t n, const double *__restrict__ T /* 12 row-major */)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	float4 a = recs[(size_t)i * 3], b = recs[(size_t)i * 3 + 1];
	double x = a.x, y = a.y, z = a.z, nx = b.x, ny = b.y, nz = b.z;
	a.x = (float)(T[0] * x + T[1] * y + T[2] * z + T[3]);
	a.y = (float)(T[4] * x + T[5] * y + T[6] * z + T[7]);
	a.z = (float)(T[8] * x + T[9] * y + T[10] * z + T[11]);
	b.x = (float)(T[0] * nx + T[1] * ny + T[2] * nz);
	b.y = (float)(T[4] * nx + T[5] * ny + T[6] * nz);
	b.z = (float)(T[8] * nx + T[9] * ny + T[10] * nz);
	recs[(size_t)i * 3] = a;
	recs[(size_t)i * 3 + 1] = b;
}

// force a given correspondence list into the flag/match/wd arrays (mulls_stage_accumulate)
__global__ void k_set_corr(uint32_t src_off, const int32_t *__restrict__ cs, const int32_t *__restrict__ ct, const float *__restrict__ cd,
						   uint32_t n, uint8_t *__restrict__ flag, int32_t *__restrict__ match, float *__restrict__ wd, uint32_t tgt_off,
						   const float4 *__restrict__ tpos, const float4 *__restrict__ tnrm, float4 *__restrict__ mq)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	const uint32_t g = src_off + (uint32_t)cs[i];
	flag[g] = MULLS_F_ALIVE | MULLS_F_VALID;
	match[g] = ct[i];
	mq[2u * g] = tpos[tgt_off + (uint32_t)ct[i]];
	mq[2u * g + 1u] = tnrm[tgt_off + (uint32_t)ct[i]];
	wd[g] = cd ? cd[i] : 0.0f;
}

// ---------------------------------------------------------------------------------------------------------------
// host-callable launch wrappers (the driver is plain C++ and never sees <<< >>>)
#include "launch.h"

void launch_accum(hipStream_t st, uint32_t njobs, const Job *jobs, const CloudDesc *descs, const PairState *states, const RunParams &rp,
				  const float4 *spos, const float4 *mq, const uint8_t *flag, float *wd, double *partial, uint32_t job_base)
{
	if (njobs)
		hipLaunchKernelGGL(k_accum, dim3(njobs), dim3(MULLS_ACC_BLOCK), 0, st, jobs, descs, states, rp, spos, mq, flag, wd, partial, job_base);
}

void launch_finish(hipStream_t st, uint32_t npairs, CloudDesc *descs, const PairState *states, const RunParams &rp, const double *partial,
				   PairOut *out, PairOut *out_host, const uint32_t *bbox, uint32_t *ticket, volatile uint32_t *host_epoch, uint32_t epoch,
				   uint32_t pair_base)
{
	if (!npairs)
		return;
	const bool direct = rp.pull_comb && npairs <= 64u; // launch-bound iterations: k_finish ships the records itself
	hipLaunchKernelGGL(k_finish, dim3(npairs), dim3(MULLS_BLOCK), 0, st, descs, states, rp, partial, out, bbox, pair_base,
					   direct ? reinterpret_cast<uint4 *>(out_host) : nullptr, ticket, host_epoch, epoch);
	if (direct)
		return;
	uint32_t n_used = 0;
	for (int c = 0; c < MULLS_NC; c++)
		n_used += rp.used[c] ? 1u : 0u;
	const uint32_t nwords = npairs * (8u + (MULLS_NTERM_PAD / 2u) * (rp.pull_comb ? 1u : n_used));
	hipLaunchKernelGGL(k_pull_outs, dim3((nwords + MULLS_BLOCK - 1) / MULLS_BLOCK), dim3(MULLS_BLOCK), 0, st, reinterpret_cast<const uint4 *>(out),
					   reinterpret_cast<uint4 *>(out_host), rp, pair_base, npairs, ticket, host_epoch, epoch);
}

void launch_push_states(hipStream_t st, const PairState *host_states, PairState *dev_states, uint32_t npairs)
{
	const uint32_t nwords = npairs * (uint32_t)(sizeof(PairState) / 16);
	if (nwords)
		hipLaunchKernelGGL(k_push_states, dim3((nwords + 255) / 256), dim3(256), 0, st, reinterpret_cast<const uint4 *>(host_states),
						   reinterpret_cast<uint4 *>(dev_states), nwords);
}

void launch_transform_aos(hipStream_t st, float4 *recs, uint32_t n, const double *T12)
{
	if (n)
		hipLaunchKernelGGL(k_transform_aos, dim3((n + 255) / 256), dim3(256), 0, st, recs, n, T12);
}

void launch_set_corr(hipStream_t st, uint32_t src_off, const int32_t *cs, const int32_t *ct, const float *cd, uint32_t n, uint8_t *flag,
					 int32_t *match, float *wd, uint32_t tgt_off, const float4 *tpos, const float4 *tnrm, float4 *mq)
{
	if (n)
		hipLaunchKernelGGL(k_set_corr, dim3((n + 255) / 256), dim3(256), 0, st, src_off, cs, ct, cd, n, flag, match, wd, tgt_off, tpos, tnrm, mq);
}
