// solve_wave.h — icp_step.h's step_solve spread over the 64 lanes of one wave.  Device code shared by the device-resident loop (k_icp.hip) and
// the lock-step loop's device step (k_reduce.hip: k_finish_step).
#pragma once
#include "icp_step.h"

// icp_step.h's step_solve, by the 64 lanes of ONE wave (the other waves of the workgroup wait at the next barrier): the same
// operations on the same operands in the same order as the host functions it mirrors (hostmath.h: invert6, solve_step,
// quat_euler_jacobian, euler_step_to_matrix, operator*), only spread over lanes wherever the host loops over independent
// elements — so the result is bit-identical to the host's (tests/test_gpu_icp.py::test_resident_loop_equals_lock_step compares
// every iteration's system, step and transform with the lock-step path's).  All data goes through LDS; a wave's LDS traffic is
// executed in order, so a wavefront-scope fence between dependent steps is all the synchronisation needed.
struct SolveWs
{
	double N[36], a[36], inv[36], b[6], sc[6], J[9], tmp[9], newg[16];
	float scf[6];
	int row_of[6], p, regular;
};
#define WSYNC() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")
#ifndef MULLS_SOLVE_MARK // tools/gpu_solve_bench.hip defines it to time the sections; nothing in the library
#define MULLS_SOLVE_MARK(k)
#endif
__device__ __forceinline__ void solve_wave(mulls::PairIter &h, const mulls::IcpConst &K, const double *comb, int i, SolveWs &w)
{
	const int l = (int)threadIdx.x; // 0..63
	// normal_from_row
	if (l < 36)
	{
		const int r = l % 6, c = l / 6;
		const double val = comb[mulls::packed_index(r < c ? r : c, r < c ? c : r)];
		w.N[l] = val;
		w.a[l] = val;
	}
	if (l < 6)
	{
		w.b[l] = comb[21 + l];
		w.row_of[l] = l;
	}
	if (l == 0)
		w.regular = 1;
	WSYNC();
	MULLS_SOLVE_MARK(0);
	// invert6: row-pivoted LU ...
	for (int col = 0; col < 6; col++)
	{
		if (l == 0)
		{
			int p = col;
			double big = fabs(w.a[col + 6 * col]);
			for (int r = col + 1; r < 6; r++)
				if (fabs(w.a[r + 6 * col]) > big)
				{
					big = fabs(w.a[r + 6 * col]);
					p = r;
				}
			if (big == 0.0)
				w.regular = 0;
			w.p = p;
		}
		WSYNC();
		const int p = w.p;
		if (p != col)
		{
			if (l < 6)
			{
				const double t = w.a[col + 6 * l];
				w.a[col + 6 * l] = w.a[p + 6 * l];
				w.a[p + 6 * l] = t;
			}
			if (l == 6)
			{
				const int t = w.row_of[col];
				w.row_of[col] = w.row_of[p];
				w.row_of[p] = t;
			}
		}
		WSYNC();
		const double piv = w.a[col + 6 * col];
		if (l < 6 && l > col)
			w.a[l + 6 * col] /= piv;
		WSYNC();
		if (l < 36)
		{
			const int r = l % 6, c = l / 6;
			if (r > col && c > col)
				w.a[r + 6 * c] -= w.a[r + 6 * col] * w.a[col + 6 * c];
		}
		WSYNC();
	}
	MULLS_SOLVE_MARK(1);
	// ... solved against the identity column by column (one lane per column, the factors in registers)
	if (l < 6)
	{
		double A[36], y[6];
#pragma unroll
		for (int k = 0; k < 36; k++)
			A[k] = w.a[k];
#pragma unroll
		for (int r = 0; r < 6; r++)
			y[r] = (w.row_of[r] == l) ? 1.0 : 0.0;
#pragma unroll
		for (int r = 1; r < 6; r++)
#pragma unroll
			for (int k = 0; k < r; k++)
				y[r] -= A[r + 6 * k] * y[k];
#pragma unroll
		for (int r = 5; r >= 0; r--)
		{
#pragma unroll
			for (int k = r + 1; k < 6; k++)
				y[r] -= A[r + 6 * k] * y[k];
			y[r] /= A[r + 6 * r];
		}
#pragma unroll
		for (int r = 0; r < 6; r++)
			w.inv[r + 6 * l] = y[r];
	}
	WSYNC();
	MULLS_SOLVE_MARK(2);
	// solve_step: x = N^-1 b
	if (l < 6)
	{
		double acc = 0.0;
		for (int c = 0; c < 6; c++)
			acc += w.inv[l + 6 * c] * w.b[c];
		h.x[l] = acc;
	}
	WSYNC();
	MULLS_SOLVE_MARK(3);
	// quat_euler_jacobian's half-angle sines / cosines (float locals in the reference, :2797-2804) and euler_step_to_matrix's
	if (l < 12)
	{
		const int q = l < 6 ? l : l - 6;
		const double v = mulls::det::trig(l < 6 ? 0.5 * h.x[3 + (q >> 1)] : h.x[3 + (q >> 1)], q & 1);
		if (l < 6)
			w.scf[q] = (float)v; // sr cr sp cp sy cy
		else
			w.sc[q] = v; // sa ca sb cb sg cg
	}
	WSYNC();
	MULLS_SOLVE_MARK(4);
	if (l == 0)
	{
		const float sr = w.scf[0], cr = w.scf[1], sp = w.scf[2], cp = w.scf[3], sy = w.scf[4], cy = w.scf[5];
		w.J[0] = 0.5 * (cr * cp * cy + sr * sp * sy);
		w.J[1] = 0.5 * (-sr * sp * cy - cr * cp * sy);
		w.J[2] = 0.5 * (-sr * cp * sy - cr * sp * cy);
		w.J[3] = 0.5 * (-sr * sp * cy + cr * cp * sy);
		w.J[4] = 0.5 * (cr * cp * cy - sr * sp * sy);
		w.J[5] = 0.5 * (-cr * sp * sy + sr * cp * cy);
		w.J[6] = 0.5 * (-sr * cp * sy - cr * sp * cy);
		w.J[7] = 0.5 * (-cr * sp * sy - sr * cp * cy);
		w.J[8] = 0.5 * (cr * cp * cy + sr * sp * sy);
	}
	if (l == 1)
	{
		const double sa = w.sc[0], ca = w.sc[1], sb = w.sc[2], cb = w.sc[3], sg = w.sc[4], cg = w.sc[5];
		mulls::Mat4 &m = h.temp;
		for (int k = 0; k < 16; k++)
			m.v[k] = 0.0;
		m.at(0, 0) = cg * cb;
		m.at(0, 1) = -sg * ca + cg * sb * sa;
		m.at(0, 2) = sg * sa + cg * sb * ca;
		m.at(1, 0) = sg * cb;
		m.at(1, 1) = cg * ca + sg * sb * sa;
		m.at(1, 2) = -cg * sa + sg * sb * ca;
		m.at(2, 0) = -sb;
		m.at(2, 1) = cb * sa;
		m.at(2, 2) = cb * ca;
		m.at(0, 3) = h.x[0];
		m.at(1, 3) = h.x[1];
		m.at(2, 3) = h.x[2];
		m.at(3, 3) = 1.0;
	}
	WSYNC();
	MULLS_SOLVE_MARK(5);
	// cofactor = N^-1 with its rotational blocks propagated to quaternion space
	if (l < 36)
		h.cofactor.v[l] = w.inv[l];
	if (l < 9)
	{
		const int r = l / 3, c = l % 3;
		w.tmp[l] = w.J[r * 3 + 0] * w.inv[(3 + 0) + 6 * (3 + c)] + w.J[r * 3 + 1] * w.inv[(3 + 1) + 6 * (3 + c)] + w.J[r * 3 + 2] * w.inv[(3 + 2) + 6 * (3 + c)];
	}
	WSYNC();
	if (l < 27)
	{
		const int blk = l / 9, r = (l % 9) / 3, c = l % 3;
		if (blk == 0)
			h.cofactor.v[(3 + r) + 6 * (3 + c)] = w.tmp[r * 3 + 0] * w.J[c * 3 + 0] + w.tmp[r * 3 + 1] * w.J[c * 3 + 1] + w.tmp[r * 3 + 2] * w.J[c * 3 + 2];
		else if (blk == 1)
			h.cofactor.v[r + 6 * (3 + c)] = w.inv[r + 6 * (3 + 0)] * w.J[c * 3 + 0] + w.inv[r + 6 * (3 + 1)] * w.J[c * 3 + 1] + w.inv[r + 6 * (3 + 2)] * w.J[c * 3 + 2];
		else
			h.cofactor.v[(3 + r) + 6 * c] = w.J[r * 3 + 0] * w.inv[(3 + 0) + 6 * c] + w.J[r * 3 + 1] * w.inv[(3 + 1) + 6 * c] + w.J[r * 3 + 2] * w.inv[(3 + 2) + 6 * c];
	}
	WSYNC();
	MULLS_SOLVE_MARK(6);
	// step-size and convergence tests (icp_step.h: step_solve)
	if (l == 0)
	{
		bool ok = w.regular != 0;
		for (int k = 0; k < 6; k++)
			ok = ok && std::isfinite(h.x[k]);
		if (!ok)
			h.singular = 1;
		const double tsn = std::sqrt(h.x[0] * h.x[0] + h.x[1] * h.x[1] + h.x[2] * h.x[2]);
		const double rsa = mulls::rotation_angle(h.temp);
		w.p = 0; // 1: advance the guess
		if (tsn > K.max_bearable_translation || std::fabs(rsa) > K.max_bearable_rotation)
		{
			h.code = -1;
			h.temp = mulls::Mat4::identity();
			h.active = 0;
			h.done = 1;
		}
		else if (i == K.max_iter_num - 1 || (i > 2 && tsn < K.converge_translation && std::fabs(rsa) < K.converge_rotation))
		{
			h.active = 0;
			h.want_residual = 1;
		}
		else
			w.p = 1;
	}
	WSYNC();
	MULLS_SOLVE_MARK(7);
	if (w.p == 1) // initial_guess = TempTran * initial_guess (:1400)
	{
		if (l < 16)
		{
			const int row = l % 4, col = l / 4;
			double acc = 0.0;
			for (int k = 0; k < 4; k++)
				acc += h.temp.at(row, k) * h.guess.at(k, col);
			w.newg[l] = acc;
		}
		WSYNC();
		if (l < 16)
			h.guess.v[l] = w.newg[l];
	}
	WSYNC();
	MULLS_SOLVE_MARK(8);
}
