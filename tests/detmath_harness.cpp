// Test harness around mulls_amd/csrc/detmath.h (vectorised entry points for tests/test_detmath.py)
#include "../mulls_amd/csrc/detmath.h"
extern "C"
{
	void dm_sin(const double *x, double *out, long n) { for (long i = 0; i < n; i++) out[i] = mulls::det::sin_cr(x[i]); }
	void dm_cos(const double *x, double *out, long n) { for (long i = 0; i < n; i++) out[i] = mulls::det::cos_cr(x[i]); }
	void dm_atan2(const double *y, const double *x, double *out, long n) { for (long i = 0; i < n; i++) out[i] = mulls::det::atan2_cr(y[i], x[i]); }
}
