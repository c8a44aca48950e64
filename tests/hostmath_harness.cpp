// Test harness: C entry points around mulls_amd/csrc/hostmath.h (the host half of an ICP iteration in the product), so that
// tests/test_hostmath.py can compare it with the oracle's own implementations of the same reference lines.
#include <cstring>

#include "../mulls_amd/csrc/hostmath.h"

extern "C"
{
	void hm_construct_trans(const double x[6], double T[16])
	{
		const mulls::Mat4 M = mulls::euler_step_to_matrix(x);
		std::memcpy(T, M.v, sizeof(M.v));
	}
	int hm_solve(const double atpa[36], const double atpb[6], double x[6], double cof[36])
	{
		mulls::Mat6 N, C;
		std::memcpy(N.v, atpa, sizeof(N.v));
		const bool ok = mulls::solve_step(N, atpb, x, C);
		std::memcpy(cof, C.v, sizeof(C.v));
		return ok ? 1 : 0;
	}
	double hm_rotation_angle(const double T[16])
	{
		mulls::Mat4 M;
		std::memcpy(M.v, T, sizeof(M.v));
		return mulls::rotation_angle(M);
	}
	void hm_invert4(const double A[16], double out[16])
	{
		mulls::Mat4 M;
		std::memcpy(M.v, A, sizeof(M.v));
		const mulls::Mat4 R = mulls::invert4(M);
		std::memcpy(out, R.v, sizeof(R.v));
	}
}
