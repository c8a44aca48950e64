// accum.h — one correspondence's contribution to the normal equations (or to the posterior residual), and the fixed-order
// reduction of a class cloud's contributions: shared by k_accum (lock-step path) and k_icp (device-resident loop), so that both
// add the same numbers in the same order and produce the same bits.
#pragma once
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


// per-class constants of one iteration's accumulation
struct AccumCtx
{
	int metric, iter_num;
	bool residual_pass, dist_w, resid_w, inten_w, faithful;
	float class_w, window;
};
// w_ground = w_roof = max_(0.01, z_xy * (m2 + 2*m3 - m4) / (0.0001 + 2.0*m1)), the other classes 1   (cregistration.hpp:1886-1894);
// cnt[c] = correspondences of class c in force this iteration
__device__ __forceinline__ float class_weight(const RunParams &rp, int cls, bool residual_pass, const int cnt[MULLS_NC])
{
	if (rp.force_class_w)
		return rp.class_w_value; // stage-level entry point only (mulls_stage_accumulate)
	if (!residual_pass && rp.w_balance && (cls == 0 || cls == 4))
	{
		const int m1 = cnt[0] + cnt[4], m2 = cnt[2], m3 = cnt[1], m4 = cnt[3];
		const double v = rp.z_xy_ratio * (m2 + 2 * m3 - m4) / (0.0001 + 2.0 * m1);
		return (float)((0.01 > v) ? 0.01 : v);
	}
	return 1.0f;
}
__device__ __forceinline__ AccumCtx accum_ctx(const RunParams &rp, int cls, int iter_num, bool residual_pass, float class_w)
{
	AccumCtx A;
	A.metric = metric_of(cls);
	A.iter_num = iter_num;
	A.residual_pass = residual_pass;
	A.dist_w = rp.w_dist;
	A.resid_w = rp.w_resid && iter_num > rp.resid_from_iter;
	A.inten_w = rp.w_inten;
	A.faithful = rp.faithful;
	A.class_w = class_w;
	A.window = A.metric == 0 ? rp.win_pl : (A.metric == 1 ? rp.win_li : rp.win_pt);
	return A;
}

// One valid correspondence: source point P (current, transformed), matched target position Q and direction N (the record
// filter_point wrote).  x: the solved step (residual pass only).  wdg: pcl::Correspondence's distance / weight union of this point.
// The 27 terms are taken in two passes of MULLS_RED_TERMS = 14 and 13 (PASS = 0 / 1: terms PASS * 14 ...; the other terms'
// arithmetic is dead code in that instantiation): 14 double accumulators per lane instead of 27 — the 1024-lane workgroups that
// sum a class cloud have 128 registers per lane, and the reduction buffer holds 14 terms at a time anyway.
#define MULLS_RED_TERMS 14
#define ACC(k, v)                                    \
	do                                               \
	{                                                \
		if ((k) / MULLS_RED_TERMS == PASS)           \
			acc[(k) % MULLS_RED_TERMS] += (v);       \
	} while (0)
template <int PASS>
__device__ __forceinline__ void accum_point(const AccumCtx &A, const double *x, const float4 P, const float4 Q, const float4 N, float &wdg,
											 double acc[MULLS_RED_TERMS])
{
	const int metric = A.metric, iter_num = A.iter_num;
	const bool residual_pass = A.residual_pass, dist_w = A.dist_w, resid_w = A.resid_w, inten_w = A.inten_w, faithful = A.faithful;
	const float class_w = A.class_w, window = A.window;
	const float px = P.x, py = P.y, pz = P.z, pi = P.w;
	const float qx = Q.x, qy = Q.y, qz = Q.z, qi = Q.w;

	if (residual_pass)
	{
				const float cw = wdg; // pcl::Correspondence::weight — for vertex points this is still d^2
		if (metric == 0)
		{
			float ntx = N.x, nty = N.y, ntz = N.z;
			float a = ntz * py - nty * pz;
			float b = ntx * pz - ntz * px;
			float c = nty * px - ntx * py;
			float dd = ntx * qx + nty * qy + ntz * qz - ntx * px - nty * py - ntz * pz;
			float res = (float)(ntx * x[0] + nty * x[1] + ntz * x[2] + a * x[3] + b * x[4] + c * x[5] - dd);
			ACC(0, cw * res * res);
			ACC(1, 1.0);
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
			ACC(0, cw * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]));
			ACC(1, 3.0);
		}
		return;
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
		wdg = w;
		ACC(0, w * ntx * ntx);
		ACC(1, w * ntx * nty);
		ACC(2, w * ntx * ntz);
		ACC(3, w * a * ntx);
		ACC(4, w * b * ntx);
		ACC(5, w * c * ntx);
		ACC(6, w * nty * nty);
		ACC(7, w * nty * ntz);
		ACC(8, w * a * nty);
		ACC(9, w * b * nty);
		ACC(10, w * c * nty);
		ACC(11, w * ntz * ntz);
		ACC(12, w * a * ntz);
		ACC(13, w * b * ntz);
		ACC(14, w * c * ntz);
		ACC(15, w * a * a);
		ACC(16, w * a * b);
		ACC(17, w * a * c);
		ACC(18, w * b * b);
		ACC(19, w * b * c);
		ACC(20, w * c * c);
		ACC(21, w * dd * ntx);
		ACC(22, w * dd * nty);
		ACC(23, w * dd * ntz);
		ACC(24, w * dd * a);
		ACC(25, w * dd * b);
		ACC(26, w * dd * c);
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
		wdg = wx;
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
				{
					ACC(k, (A[0][j] * A[0][c] + A[1][j] * A[1][c]) + A[2][j] * A[2][c]);
					k++;
				}
#pragma unroll
		for (int j = 0; j < 6; j++)
			ACC(21 + j, (A[0][j] * bv[0] + A[1][j] * bv[1]) + A[2][j] * bv[2]);
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
		if (!faithful)
			wdg = wx; // intended behaviour: weight the vertex residual by its weight, not by d^2
		ACC(0, wx);
		ACC(4, wx * pz);
		ACC(5, (-wx * py));
		ACC(6, wy);
		ACC(8, (-wy * pz));
		ACC(10, wy * px);
		ACC(11, wz);
		ACC(12, wz * py);
		ACC(13, (-wz * px));
		ACC(15, wy * pz * pz + wz * py * py);
		ACC(16, (-wz * px * py));
		ACC(17, (-wy * px * pz));
		ACC(18, wx * pz * pz + wz * px * px);
		ACC(19, (-wx * py * pz));
		ACC(20, wx * py * py + wy * px * px);
		ACC(21, (-wx * dx));
		ACC(22, (-wy * dy));
		ACC(23, (-wz * dz));
		ACC(24, wy * pz * dy - wz * py * dz);
		ACC(25, wz * px * dz - wx * pz * dx);
		ACC(26, wx * py * dx - wy * px * dy);
	}
}
#undef ACC

// ---------------------------------------------------------------------------------------------------------------
// Summation order of a class cloud's row — the one order every path of the library uses (k_accum + k_finish in the lock-step
// path, k_icp in the device-resident loop), so that they produce the same bits:
//   the source slots are taken in chunks of MULLS_ACC_CHUNK = 8 x 1024; inside a chunk, virtual lane v (0..1023) adds the
//   contributions of the slots v, v + 1024, ..., v + 7168 in that order (dead / invalid slots add nothing); for each of the 27
//   terms the 1024 lane values are laid out in LDS, lane j of a wave adds the 16 values j, j + 64, ..., j + 960 in order, a
//   butterfly adds the 64 partial sums ((xor 1, xor 2, mirror 8, mirror 16) inside the 16-lane rows, then (row0 + row1) +
//   (row2 + row3)); chunk sums are added in chunk order.
#define MULLS_ACC_LANES 1024
#define MULLS_ACC_CHUNK (8u * MULLS_ACC_LANES)
#define MULLS_RED_BYTES ((size_t)MULLS_RED_TERMS * MULLS_ACC_LANES * sizeof(double))

namespace
{
template <int CTRL>
__device__ __forceinline__ double dpp_add_f64(double v)
{
	const unsigned long long b = (unsigned long long)__double_as_longlong(v);
	const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, 0xf, 0xf, false);
	const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, 0xf, 0xf, false);
	return v + __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
	const unsigned long long b = (unsigned long long)__double_as_longlong(v);
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, lane);
	const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), lane);
	return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
} // namespace

// one pass (MULLS_RED_TERMS terms) of the sums over the 1024 lanes of the workgroup (every lane calls; R: LDS, MULLS_RED_BYTES;
// out: LDS, 27 doubles)
template <int PASS>
__device__ __forceinline__ void pass_reduce(const double acc[MULLS_RED_TERMS], double *R, double *out)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int t0 = PASS * MULLS_RED_TERMS, nt = PASS ? MULLS_NTERM - MULLS_RED_TERMS : MULLS_RED_TERMS;
	__syncthreads(); // the buffer is free
#pragma unroll
	for (int k = 0; k < nt; k++)
		R[k * MULLS_ACC_LANES + threadIdx.x] = acc[k];
	__syncthreads();
	if (wave < nt)
	{
		const double *r = R + wave * MULLS_ACC_LANES;
		double s = r[lane];
#pragma unroll
		for (int k = 1; k < MULLS_ACC_LANES / 64; k++)
			s += r[lane + 64 * k];
		s = dpp_add_f64<0xB1>(s);  // quad_perm [1,0,3,2]
		s = dpp_add_f64<0x4E>(s);  // quad_perm [2,3,0,1]
		s = dpp_add_f64<0x141>(s); // row_half_mirror
		s = dpp_add_f64<0x140>(s); // row_mirror: every lane of a row holds the row's sum
		const double r0 = readlane_f64(s, 0), r1 = readlane_f64(s, 16), r2 = readlane_f64(s, 32), r3 = readlane_f64(s, 48);
		if (lane == 0)
			out[t0 + wave] = (r0 + r1) + (r2 + r3);
	}
}

template <int PASS>
__device__ __forceinline__ void chunk_pass(const AccumCtx &A, const double *x, const CloudDesc &d, uint32_t chunk, const float4 *__restrict__ spos,
											const float4 *__restrict__ mq, const uint8_t *__restrict__ flag, float *__restrict__ wd, double *R, double *part)
{
	double acc[MULLS_RED_TERMS];
#pragma unroll
	for (int k = 0; k < MULLS_RED_TERMS; k++)
		acc[k] = 0.0;
	const uint32_t end = min(d.src_n, chunk + MULLS_ACC_CHUNK);
	if (PASS == 0 || !A.residual_pass) // the residual pass has two terms only
		for (uint32_t s = chunk + threadIdx.x; s < end; s += MULLS_ACC_LANES)
		{
			const uint32_t g = d.src_off + s;
			if ((flag[g] & (MULLS_F_ALIVE | MULLS_F_VALID)) != (MULLS_F_ALIVE | MULLS_F_VALID))
				continue;
			accum_point<PASS>(A, x, spos[g], mq[2u * g], mq[2u * g + 1u], wd[g], acc);
		}
	pass_reduce<PASS>(acc, R, part);
}

// the sum of one chunk of a class cloud's source slots [chunk, chunk + MULLS_ACC_CHUNK) -> part[0..26] (LDS); 1024 lanes
__device__ __forceinline__ void chunk_sum(const AccumCtx &A, const double *x, const CloudDesc &d, uint32_t chunk, const float4 *__restrict__ spos,
										   const float4 *__restrict__ mq, const uint8_t *__restrict__ flag, float *__restrict__ wd, double *R, double *part)
{
	chunk_pass<0>(A, x, d, chunk, spos, mq, flag, wd, R, part);
	chunk_pass<1>(A, x, d, chunk, spos, mq, flag, wd, R, part);
	__syncthreads();
}

// one class cloud's whole row: chunk sums in chunk order -> row[0..26] (LDS)
__device__ __forceinline__ void class_row(const AccumCtx &A, const double *x, const CloudDesc &d, const float4 *__restrict__ spos,
										   const float4 *__restrict__ mq, const uint8_t *__restrict__ flag, float *__restrict__ wd, double *R, double *row)
{
	__shared__ double part[MULLS_NTERM_PAD];
	const uint32_t src_n = d.src_n;
	uint32_t chunk = 0;
	do
	{
		chunk_sum(A, x, d, chunk, spos, mq, flag, wd, R, part);
		if (threadIdx.x < MULLS_NTERM)
			row[threadIdx.x] = chunk ? row[threadIdx.x] + part[threadIdx.x] : 0.0 + part[threadIdx.x]; // as k_finish adds the chunk partials to 0.0
		__syncthreads();
		chunk += MULLS_ACC_CHUNK;
	} while (chunk < src_n);
}

// Term t (0..26) of the one system the reference solves, from the class rows.  The 6x6 the reference inverts: pt2pl / pt2pt
// wrote the lower triangle, pt2li the upper one, then the mirror copies lower -> upper (cregistration.hpp:1924-1938).  Class
// order of the += chain on shared slots: ground, facade, roof (pl), pillar, beam (li), vertex (pt) (:1914-1921) — the same
// additions in the same order.  Residual pass: [0] = sum of the weighted squared residuals, [1] = number of observations
// (get_multi_metrics_lls_residual).
__device__ __forceinline__ void combine_rows(const RunParams &rp, bool want_residual, const double (*sums)[MULLS_NTERM_PAD], double *comb, int t)
{
	const int order[MULLS_NC] = {0, 2, 4, 1, 3, 5};
	double val;
	if (want_residual)
	{
		val = 0.0;
		if (t < 2)
			for (int i = 0; i < MULLS_NC; i++)
				if (rp.used[order[i]])
					val += sums[order[i]][t];
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
			const double v = sums[cls][t];
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
				val += sums[order[i]][t];
	}
	comb[t] = val;
}
