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


// get_weight_by_intensity of one correspondence (source intensity P.w, target intensity Q.w), cregistration.hpp:2710-2715
__device__ __forceinline__ float point_wi(const float4 P, const float4 Q) { return w_intensity((float)(P.w + 0.0001), (float)(Q.w + 0.0001)); }

// per-class constants of one iteration's accumulation
struct AccumCtx
{
	int metric, iter_num;
	bool residual_pass, dist_w, resid_w, inten_w, faithful;
	bool li_diag; // point-to-line class, faithful mode, only the combined system is wanted: the mirror (cregistration.hpp:1924-1938) throws the
				  // 15 off-diagonal terms of these classes away, so only the 6 diagonal terms and the right-hand side are summed
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
	A.li_diag = A.metric == 1 && rp.faithful && rp.pull_comb && !residual_pass;
	A.class_w = class_w;
	A.window = A.metric == 0 ? rp.win_pl : (A.metric == 1 ? rp.win_li : rp.win_pt);
	return A;
}

// The weight of one valid correspondence (normal-equation pass): class weight x distance weight x residual weight x intensity weight, multiplied in the order the
// metric's summation function does (pt2pl cregistration.hpp:2103-2113, pt2li :2215-2224, pt2pt :2003-2012).  One function for every caller: k_accum_wave evaluates it
// ONCE per slot and hands it to the term windows (the weights sit under run-time flags, and the compiler does not merge the copies of a conditional computation the
// windows would otherwise carry: three double divisions, six float divisions and three square roots per slot instead of one, two and one).
__device__ __forceinline__ float corr_weight(const AccumCtx &A, int metric, const float4 P, const float4 Q, const float4 N, float wi)
{
	const float px = P.x, py = P.y, pz = P.z;
	const float qx = Q.x, qy = Q.y, qz = Q.z;
	const float dist = sqrtf(qx * qx + qy * qy + qz * qz);
	if (metric == 0)
	{
		float ntx = N.x, nty = N.y, ntz = N.z;
		float w = A.class_w;
		float dd = ntx * qx + nty * qy + ntz * qz - ntx * px - nty * py - ntz * pz;
		if (A.dist_w)
			w = w * w_dist_adaptive(dist, A.iter_num);
		if (A.resid_w)
			w = w * w_residual(fabsf(dd), A.window);
		if (A.inten_w)
			w = w * wi;
		return w;
	}
	if (metric == 1)
	{
		float vx = N.x, vy = N.y, vz = N.z;
		float dx = px - qx, dy = py - qy, dz = pz - qz;
		const double b0 = -vy * dz + vz * dy, b1 = -vz * dx + vx * dz, b2 = -vx * dy + vy * dx;
		float ex = (float)fabs(b0), ey = (float)fabs(b1), ez = (float)fabs(b2);
		float ed = sqrtf(ex * ex + ey * ey + ez * ez);
		float wx = A.class_w;
		if (A.dist_w)
			wx *= w_dist_adaptive(dist, A.iter_num);
		if (A.inten_w)
			wx *= wi;
		if (A.resid_w)
			wx = wx * w_residual(ed, A.window);
		return wx;
	}
	float dx = px - qx, dy = py - qy, dz = pz - qz;
	float wx = A.class_w;
	if (A.dist_w)
		wx = wx * w_dist_adaptive(dist, A.iter_num);
	if (A.resid_w)
		wx = wx * w_residual(sqrtf(dx * dx + dy * dy + dz * dz), A.window);
	if (A.inten_w)
		wx = wx * wi;
	return wx;
}

// One valid correspondence: source point P (current, transformed), matched target position Q and direction N (the record
// filter_point wrote).  x: the solved step (residual pass only).  wdg: pcl::Correspondence's distance / weight union of this point.
// Writes the terms T0 .. T0 + NT - 1 of the point's contribution into t[] (terms the metric does not have stay as they are: the
// caller zeroes t[]); the arithmetic of the other terms is dead code in that instantiation.  wi: the point's intensity weight
// (point_wi; evaluated by the caller, once per point and away from the terms' registers — a double-precision exp).  TS = float where every term is a
// float expression of the reference (point-to-plane and point-to-point normal equations) — it converts to double exactly when
// the sum is taken, as the reference's `double += float expression` does — and double elsewhere.
// wpre: the correspondence's weight if the caller has evaluated it already (corr_weight: the same value), else nullptr.
// MODE 1 (point-to-line classes with AccumCtx::li_diag): the six diagonal terms (0, 6, 11, 15, 18, 20) and the six right-hand-side
// terms (21..26) are numbered 0..11, and T0 / NT select among those
__device__ __forceinline__ constexpr int li_slot(int k) { return k == 0 ? 0 : (k == 6 ? 1 : (k == 11 ? 2 : (k == 15 ? 3 : (k == 18 ? 4 : (k == 20 ? 5 : (k >= 21 ? k - 15 : -1)))))); }
__device__ __forceinline__ constexpr int li_term(int slot) { return slot == 0 ? 0 : (slot == 1 ? 6 : (slot == 2 ? 11 : (slot == 3 ? 15 : (slot == 4 ? 18 : (slot == 5 ? 20 : slot + 15))))); }
#define ACC(k, v)                                              \
	do                                                         \
	{                                                          \
		if (MODE == 1)                                         \
		{                                                      \
			if (li_slot(k) >= T0 && li_slot(k) < T0 + NT)      \
				t[li_slot(k) >= T0 ? li_slot(k) - T0 : 0] = (TS)(v); \
		}                                                      \
		else if ((k) >= T0 && (k) < T0 + NT)                   \
			t[(k)-T0] = (TS)(v);                               \
	} while (0)
// METRIC >= 0: the caller knows the metric at compile time (k_accum_wave: the other metrics' arithmetic is not even compiled — their temporaries would set the
// kernel's register count)
template <typename TS, int T0, int NT, int MODE = 0, int METRIC = -1>
__device__ __forceinline__ void point_terms(const AccumCtx &A, const double *x, const float4 P, const float4 Q, const float4 N, float wi, float &wdg, TS t[NT],
											 const float *wpre = nullptr)
{
	const int metric = METRIC >= 0 ? METRIC : A.metric;
	const bool residual_pass = A.residual_pass, faithful = A.faithful;
	const float px = P.x, py = P.y, pz = P.z;
	const float qx = Q.x, qy = Q.y, qz = Q.z;

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

	if (metric == 0) // pt2pl_lls_summation, cregistration.hpp:2066-2156
	{
		float ntx = N.x, nty = N.y, ntz = N.z;
		float a = ntz * py - nty * pz;
		float b = ntx * pz - ntz * px;
		float c = nty * px - ntx * py;
		float dd = ntx * qx + nty * qy + ntz * qz - ntx * px - nty * py - ntz * pz;
		const float w = wpre ? *wpre : corr_weight(A, 0, P, Q, N, wi);
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
		const float wx = wpre ? *wpre : corr_weight(A, 1, P, Q, N, wi); // (in front of the local matrix A)
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
		const float wx = wpre ? *wpre : corr_weight(A, 2, P, Q, N, wi);
		const float wy = wx, wz = wx;
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
//   the source slots are taken in trips of MULLS_ACC_LANES = 1024; the term of slot trip * 1024 + v is value v of the trip (0 for
//   a dead / invalid slot); for each of the 27 terms lane j of a wave adds the 16 values j, j + 64, ..., j + 960 in that order
//   (in double), a butterfly adds the 64 partial sums ((xor 1, xor 2, mirror 8, mirror 16) inside the 16-lane rows, then
//   (row0 + row1) + (row2 + row3)); the trip sums are added to 0.0 in trip order.
#define MULLS_ACC_LANES 1024
#define MULLS_RED_BYTES ((size_t)27 * MULLS_ACC_LANES * sizeof(float)) // the LDS term buffer: 27 float terms, or 13 double terms, of 1024 slots
#define MULLS_RED_BYTES_HALF ((size_t)14 * MULLS_ACC_LANES * sizeof(float)) // ... in the two-halves mode: 14 float terms (or 7 double terms)

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
// ... through a DPP control that reaches only the rows of ROWS (row_bcast:15 / :31): the other rows add +0.0
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_add_f64_rows(double v)
{
	const unsigned long long b = (unsigned long long)__double_as_longlong(v);
	const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, ROWS, 0xf, false);
	const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, ROWS, 0xf, false);
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

// Terms T0 .. T0 + NT - 1 of one slot (lane) into the LDS term buffer R (NT x 1024 values of TS): zeros unless `valid`.
// LANES: the slots the trip can have (1024; k_accum's workgroups for short trips use 512 / 256 — the slots beyond are +0.0 in the sum anyway)
template <typename TS, int T0, int NT, int MODE = 0, int LANES = MULLS_ACC_LANES>
__device__ __forceinline__ void slot_terms(const AccumCtx &A, const double *x, bool valid, const float4 P, const float4 Q, const float4 N, float wi, float &wdg, TS *R)
{
	static_assert(sizeof(TS) * NT * LANES <= MULLS_RED_BYTES, "the term buffer holds NT terms of LANES slots");
	TS t[NT];
#pragma unroll
	for (int k = 0; k < NT; k++)
		t[k] = (TS)0;
	if (valid)
		point_terms<TS, T0, NT, MODE>(A, x, P, Q, N, wi, wdg, t);
#pragma unroll
	for (int k = 0; k < NT; k++)
		R[k * LANES + threadIdx.x] = t[k];
}
// ... and their sums over the 1024 slots -> part[T0 ..] (LDS).  The caller puts a barrier between slot_terms and reduce_terms,
// and another one before the buffer is written again.
template <typename TS, int T0, int NT, int MODE = 0, int LANES = MULLS_ACC_LANES>
__device__ __forceinline__ void reduce_terms(const TS *R, double *part)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
	for (int k0 = 0; k0 < NT; k0 += LANES / 64)
	{
		const int k = k0 + wave;
		if (k < NT)
		{
			const TS *r = R + k * LANES;
			double sum = (double)r[lane];
#pragma unroll
			for (int i = 1; i < LANES / 64; i++) // all loads in flight, then the chain of adds (values beyond the cloud are +0.0: with fewer LANES they are not
												 // added at all — the same sum, and a -0.0 that survives only that way is added to +0.0 by k_finish)
				sum += (double)r[lane + 64 * i];
			sum = dpp_add_f64<0xB1>(sum);	// quad_perm [1,0,3,2]
			sum = dpp_add_f64<0x4E>(sum);	// quad_perm [2,3,0,1]
			sum = dpp_add_f64<0x141>(sum); // row_half_mirror
			sum = dpp_add_f64<0x140>(sum); // row_mirror: every lane of a row holds the row's sum
			const double r0 = readlane_f64(sum, 0), r1 = readlane_f64(sum, 16), r2 = readlane_f64(sum, 32), r3 = readlane_f64(sum, 48);
			if (lane == 0)
				part[MODE == 1 ? li_term(T0 + k) : T0 + k] = (r0 + r1) + (r2 + r3);
		}
	}
}

// The 27 sums of one trip of 1024 slots whose data the lanes hold in registers (valid, P, Q, N, wdg as slot_terms) -> part[0..26]
// (LDS; every entry written).  R: LDS, MULLS_RED_BYTES.  Every lane calls; ends with a barrier.
template <bool HALF = false, int LANES = MULLS_ACC_LANES>
__device__ __forceinline__ void trip_sum_regs(const AccumCtx &A, const double *x, bool valid, const float4 P, const float4 Q, const float4 N, float &wdg, void *R,
											   double *part)
{
	const float wi = (valid && A.inten_w && !A.residual_pass) ? point_wi(P, Q) : 1.0f;
	if (A.residual_pass)
	{
		__syncthreads(); // the buffer is free
		slot_terms<double, 0, 2, 0, LANES>(A, x, valid, P, Q, N, wi, wdg, static_cast<double *>(R)); // sum of w * r^2, number of observations
		__syncthreads();
		reduce_terms<double, 0, 2, 0, LANES>(static_cast<const double *>(R), part);
		if (threadIdx.x >= 2 && threadIdx.x < MULLS_NTERM)
			part[threadIdx.x] = 0.0;
	}
	else if (A.li_diag)
	{
		// point-to-line, faithful: diagonal + right-hand side (the off-diagonal sums would be dropped by the mirror); the other entries are 0
		__syncthreads();
		if (threadIdx.x < 21 && li_slot((int)threadIdx.x) < 0)
			part[threadIdx.x] = 0.0;
		if (HALF)
		{
			slot_terms<double, 0, 6, 1, LANES>(A, x, valid, P, Q, N, wi, wdg, static_cast<double *>(R));
			__syncthreads();
			reduce_terms<double, 0, 6, 1, LANES>(static_cast<const double *>(R), part);
			__syncthreads();
			slot_terms<double, 6, 6, 1, LANES>(A, x, valid, P, Q, N, wi, wdg, static_cast<double *>(R));
			__syncthreads();
			reduce_terms<double, 6, 6, 1, LANES>(static_cast<const double *>(R), part);
		}
		else
		{
			slot_terms<double, 0, 12, 1, LANES>(A, x, valid, P, Q, N, wi, wdg, static_cast<double *>(R));
			__syncthreads();
			reduce_terms<double, 0, 12, 1, LANES>(static_cast<const double *>(R), part);
		}
	}
	else if (A.metric == 1 && HALF)
	{
		__syncthreads();
		slot_terms<double, 0, 7, 0, LANES>(A, x, valid, P, Q, N, wi, wdg, static_cast<double *>(R));
		__syncthreads();
		reduce_terms<double, 0, 7, 0, LANES>(static_cast<const double *>(R), part);
		__syncthreads();
		slot_terms<double, 7, 7, 0, LANES>(A, x, valid, P, Q, N, wi, wdg, static_cast<double *>(R));
		__syncthreads();
		reduce_terms<double, 7, 7, 0, LANES>(static_cast<const double *>(R), part);
		__syncthreads();
		slot_terms<double, 14, 7, 0, LANES>(A, x, valid, P, Q, N, wi, wdg, static_cast<double *>(R));
		__syncthreads();
		reduce_terms<double, 14, 7, 0, LANES>(static_cast<const double *>(R), part);
		__syncthreads();
		slot_terms<double, 21, 6, 0, LANES>(A, x, valid, P, Q, N, wi, wdg, static_cast<double *>(R));
		__syncthreads();
		reduce_terms<double, 21, 6, 0, LANES>(static_cast<const double *>(R), part);
	}
	else if (A.metric == 1)
	{
		// point-to-line: the terms are products of doubles
		__syncthreads();
		slot_terms<double, 0, 13, 0, LANES>(A, x, valid, P, Q, N, wi, wdg, static_cast<double *>(R));
		__syncthreads();
		reduce_terms<double, 0, 13, 0, LANES>(static_cast<const double *>(R), part);
		__syncthreads();
		slot_terms<double, 13, 13, 0, LANES>(A, x, valid, P, Q, N, wi, wdg, static_cast<double *>(R));
		__syncthreads();
		reduce_terms<double, 13, 13, 0, LANES>(static_cast<const double *>(R), part);
		__syncthreads();
		slot_terms<double, 26, 1, 0, LANES>(A, x, valid, P, Q, N, wi, wdg, static_cast<double *>(R));
		__syncthreads();
		reduce_terms<double, 26, 1, 0, LANES>(static_cast<const double *>(R), part);
	}
	else if (HALF)
	{
		// two halves through a buffer of 14 float terms (56 KiB: two workgroups per CU, k_accum); same sums, term by term
		__syncthreads();
		slot_terms<float, 0, 14, 0, LANES>(A, x, valid, P, Q, N, wi, wdg, static_cast<float *>(R));
		__syncthreads();
		reduce_terms<float, 0, 14, 0, LANES>(static_cast<const float *>(R), part);
		__syncthreads();
		slot_terms<float, 14, 13, 0, LANES>(A, x, valid, P, Q, N, wi, wdg, static_cast<float *>(R));
		__syncthreads();
		reduce_terms<float, 14, 13, 0, LANES>(static_cast<const float *>(R), part);
	}
	else
	{
		__syncthreads();
		slot_terms<float, 0, 27, 0, LANES>(A, x, valid, P, Q, N, wi, wdg, static_cast<float *>(R));
		__syncthreads();
		reduce_terms<float, 0, 27, 0, LANES>(static_cast<const float *>(R), part);
	}
	__syncthreads();
}

// the 27 sums of one trip of a class cloud's source slots [trip0, trip0 + 1024), read from memory (all loads issued before the
// validity test: one memory round trip) -> part[0..26]
template <bool HALF = false, int LANES = MULLS_ACC_LANES>
__device__ __forceinline__ void trip_sum(const AccumCtx &A, const double *x, const CloudDesc &d, uint32_t trip0, const float4 *__restrict__ spos,
										  const float4 *__restrict__ mq, const uint8_t *__restrict__ flag, float *__restrict__ wd, void *R, double *part)
{
	const uint32_t s = trip0 + threadIdx.x;
	bool valid = false;
	float4 P = make_float4(0.0f, 0.0f, 0.0f, 0.0f), Q = P, N = P;
	float w = 0.0f, w0 = 0.0f;
	uint32_t g = 0;
	if (s < d.src_n)
	{
		g = d.src_off + s;
		const uint32_t f = flag[g];
		P = spos[g], Q = mq[2u * g], N = mq[2u * g + 1u];
		w = w0 = wd[g];
		valid = (f & (MULLS_F_ALIVE | MULLS_F_VALID)) == (MULLS_F_ALIVE | MULLS_F_VALID);
	}
	trip_sum_regs<HALF, LANES>(A, x, valid, P, Q, N, w, R, part);
	if (valid && __float_as_uint(w) != __float_as_uint(w0))
		wd[g] = w; // pcl::Correspondence::weight
}

// one class cloud's whole row: trip sums in trip order -> row[0..26] (LDS)
__device__ __forceinline__ void class_row(const AccumCtx &A, const double *x, const CloudDesc &d, const float4 *__restrict__ spos,
										   const float4 *__restrict__ mq, const uint8_t *__restrict__ flag, float *__restrict__ wd, void *R, double *row)
{
	__shared__ double part[MULLS_NTERM_PAD];
	const uint32_t src_n = d.src_n;
	uint32_t trip0 = 0;
	do
	{
		trip_sum(A, x, d, trip0, spos, mq, flag, wd, R, part);
		if (threadIdx.x < MULLS_NTERM)
			row[threadIdx.x] = trip0 ? row[threadIdx.x] + part[threadIdx.x] : 0.0 + part[threadIdx.x]; // as k_finish adds the trip partials to 0.0
		__syncthreads();
		trip0 += MULLS_ACC_LANES;
	} while (trip0 < src_n);
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
