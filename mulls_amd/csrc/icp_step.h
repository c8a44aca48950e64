// icp_step.h — the O(1)-per-iteration half of mm_lls_icp (cregistration.hpp:1296-1401): correspondence-count test, threshold
// update, 6x6 solve, step-size and convergence tests, posterior sigma and information matrix.  Host AND device code: the
// lock-step driver (driver.cpp, host_step) and the device-resident loop (k_icp.hip) both call these functions, and every
// operation in them is IEEE arithmetic in a fixed order (hostmath.h / detmath.h), so the two paths produce the same bits.
#pragma once
#include <stdint.h>

#include "hostmath.h"

namespace mulls
{

// run-wide constants, derived from mulls_params once per run (the float conversions are the reference's, :1150-1157)
struct IcpConst
{
	int32_t max_iter_num;
	float converge_translation;
	float converge_rotation;		// converge_rotation_d / 180 * pi, stored as float
	float max_bearable_translation; // 2.0 * dis_thre_unit, stored as float
	float max_bearable_rotation;	// max_bearable_rotation_d / 180 * pi, stored as float
	float dis_thre_unit, dis_thre_min, dis_thre_update_rate;
	float min_neccessary_corr_ratio, sigma_thre;
};

// the life of one pair through the loop
struct PairIter
{
	Mat4 guess, temp; // initial_guess accumulated so far; TempTran of the last solved step
	Mat6 cofactor, info;
	double x[6];
	double sigma2;
	float thr[6];
	float ratio;
	int32_t code, iters, src_feature_count, singular;
	int32_t active, want_residual, done;
};

// Lock-step loop with this half of the iteration on the device (k_reduce.hip: k_finish_step): what the host otherwise keeps per pair between
// the launches, resident in HBM instead
struct StepState
{
	PairIter h;
	uint32_t alive_prev[6];						  // live source points per class before the iteration's search
	uint32_t ncorr[6], nsrc0[6], ntgt0[6], bbox[6]; // what the result record reports: last |Corr_f|, post-filter cloud sizes, crop box keys
	uint32_t first;									  // the first iteration's bookkeeping (cloud sizes, crop box, source_feature_points_count) is still to do
	uint32_t pad_;
	unsigned long long src_pts, tgt_pts, tgt_job_pts, pair_evals, corr_pts; // profile counters, summed over the iterations
};

MULLS_HD inline void pair_iter_init(PairIter &h, const double guess_rows[12], const IcpConst &K)
{
	for (int r = 0; r < 3; r++)
		for (int c = 0; c < 4; c++)
			h.guess.at(r, c) = guess_rows[r * 4 + c];
	h.guess.at(3, 0) = h.guess.at(3, 1) = h.guess.at(3, 2) = 0.0;
	h.guess.at(3, 3) = 1.0;
	h.temp = Mat4::identity();
	h.cofactor = Mat6::identity();
	h.info = Mat6::identity();
	for (int c = 0; c < 6; c++)
	{
		h.x[c] = 0.0;
		h.thr[c] = K.dis_thre_unit;
	}
	h.sigma2 = 1.0;
	h.ratio = 1.0f;
	h.code = 0;
	h.iters = 0;
	h.src_feature_count = 0;
	h.singular = 0;
	h.active = K.max_iter_num > 0;
	h.want_residual = 0;
	h.done = !h.active;
}

// After this iteration's search: the correspondence-count test (:1296-1311) and, if it passes, the threshold update that
// follows it (update_corr_dist_thre, :1855-1866).  n_valid: |Corr_f| per class, index = used_feature_type character index
// (ground, pillar, facade, beam, roof, vertex).  Returns false on process code -2.
MULLS_HD inline bool step_counts(PairIter &h, const IcpConst &K, const uint32_t n_valid[6])
{
	int total = 0;
	for (int c = 0; c < 6; c++)
		total += (int)n_valid[c];
	const int necessary = (int)(n_valid[1] + n_valid[3] + n_valid[2]); // pillar + beam + facade
	h.ratio = (float)(1.0 * necessary / h.src_feature_count);
	if (total < 40 || necessary < 20 || h.ratio < K.min_neccessary_corr_ratio)
	{
		h.code = -2;
		h.temp = Mat4::identity();
		h.active = 0;
		h.done = 1;
		return false;
	}
	for (int c = 0; c < 6; c++)
	{
		const double v = 1.0 * h.thr[c] / K.dis_thre_update_rate;
		h.thr[c] = (float)((v > K.dis_thre_min) ? v : (double)K.dis_thre_min);
	}
	return true;
}

// packed index of (r,c), r <= c, in the row-major-upper enumeration of the accumulated terms
MULLS_HD inline int packed_index(int r, int c) { return r * 6 - r * (r - 1) / 2 + (c - r); }

// The 6x6 the reference inverts and its right-hand side from the combined row (21 packed terms + 6): the lower / upper
// triangle bookkeeping of cregistration.hpp:1914-1938 has been applied when the row was combined.
MULLS_HD inline void normal_from_row(const double comb[27], Mat6 &N, double b[6])
{
	for (int r = 0; r < 6; r++)
		for (int c = r; c < 6; c++)
		{
			const double val = comb[packed_index(r, c)];
			N.at(c, r) = val;
			N.at(r, c) = val;
		}
	for (int j = 0; j < 6; j++)
		b[j] = comb[21 + j];
}

// Solve (:1924-1964), step-size test (:1344-1354) and convergence test (:1357) of iteration i.
// Afterwards: h.done (code -1), h.want_residual (converged or last iteration: the residual pass comes next), or h.guess advanced.
MULLS_HD inline void step_solve(PairIter &h, const IcpConst &K, const Mat6 &N, const double b[6], int i)
{
	if (!solve_step(N, b, h.x, h.cofactor))
		h.singular = 1;
	h.temp = euler_step_to_matrix(h.x);
	const double tsn = std::sqrt(h.x[0] * h.x[0] + h.x[1] * h.x[1] + h.x[2] * h.x[2]);
	const double rsa = rotation_angle(h.temp);
	if (tsn > K.max_bearable_translation || std::fabs(rsa) > K.max_bearable_rotation)
	{
		h.code = -1;
		h.temp = Mat4::identity();
		h.active = 0;
		h.done = 1;
		return;
	}
	if (i == K.max_iter_num - 1 || (i > 2 && tsn < K.converge_translation && std::fabs(rsa) < K.converge_rotation))
	{
		h.active = 0;
		h.want_residual = 1; // the residual pass runs before anything else touches this pair
		return;
	}
	h.guess = h.temp * h.guess; // :1400
}

// get_multi_metrics_lls_residual's result (:2518-2544) -> sigma, process code, information matrix (:1357-1395)
MULLS_HD inline void step_residual(PairIter &h, const IcpConst &K, double VTPV, double observations)
{
	const long obs = (long)observations;
	h.sigma2 = VTPV / (double)((int)obs - 6);
	h.code = (std::sqrt(h.sigma2) < (double)K.sigma_thre) ? 1 : -3;
	MULLS_WORK Mat6 cinv;
	invert6(h.cofactor, cinv);
	for (int k = 0; k < 36; k++)
		h.info.v[k] = (1.0 / h.sigma2) * cinv.v[k];
	h.want_residual = 0;
	h.done = 1;
}

} // namespace mulls
