// detmath.h — the three transcendental functions of the per-iteration algebra (sin, cos, atan2), written out so that the
// host driver and the device-resident ICP loop compute THE SAME BITS: plain IEEE-754 double operations in a fixed order
// (the library is built -ffp-contract=off; the only fused operation is the explicit fma of the exact product), no libm.
//
// Evaluation is in double-double (~104 bits) and the result is the rounding of that value to double, i.e. the correctly
// rounded function value except when the true value lies within ~2^-100 of a rounding boundary.  In front of it sits a quick
// phase in Ziv's manner for the arguments an ICP step produces (|angle| <= 0.5 rad, tan <= 1/8): the leading correction term in
// double-double, the tail of the series in double, an explicit bound on what that leaves out — the value is returned only when
// rounding it gives the same double with the bound added and subtracted, i.e. when it IS the correctly rounded result; otherwise
// (a few calls in a million at ICP angles) the full evaluation runs.  Same bits either way, a tenth of the operations.  The reference calls glibc's
// sin / cos / atan2 (construct_trans_a cregistration.hpp:2740-2764, get_quat_euler_jacobi :2795-2819, Eigen::AngleAxisd :1345);
// glibc 2.35 documents an error below 1 ulp for them, not 0.5: its results equal the values computed here except at the rare
// arguments where glibc itself misrounds (tests/test_detmath.py counts them over the ICP range of angles).
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define MULLS_HD __host__ __device__
#else
#define MULLS_HD
#endif
// Work arrays of the per-iteration algebra: plain locals on the host; on the device they live in LDS (the algebra is run by one
// lane of a workgroup — k_icp — and dynamically indexed locals would otherwise sit in scratch memory, ~10x the latency).
#if defined(__HIP_DEVICE_COMPILE__)
#define MULLS_WORK static __shared__
#else
#define MULLS_WORK
#endif

namespace mulls
{
namespace det
{
struct dd
{
	double hi, lo;
};
MULLS_HD inline dd two_sum(double a, double b)
{
	const double s = a + b, bb = s - a;
	return {s, (a - (s - bb)) + (b - bb)};
}
MULLS_HD inline dd fast_two_sum(double a, double b) // |a| >= |b| (or a == 0)
{
	const double s = a + b;
	return {s, b - (s - a)};
}
MULLS_HD inline dd two_prod(double a, double b)
{
	const double p = a * b;
	return {p, __builtin_fma(a, b, -p)}; // exact: a*b = p + e
}
MULLS_HD inline dd neg(dd a) { return {-a.hi, -a.lo}; }
MULLS_HD inline dd add(dd a, dd b)
{
	dd s = two_sum(a.hi, b.hi);
	const dd t = two_sum(a.lo, b.lo);
	s.lo += t.hi;
	s = fast_two_sum(s.hi, s.lo);
	s.lo += t.lo;
	return fast_two_sum(s.hi, s.lo);
}
MULLS_HD inline dd add(dd a, double b)
{
	dd s = two_sum(a.hi, b);
	s.lo += a.lo;
	return fast_two_sum(s.hi, s.lo);
}
MULLS_HD inline dd mul(dd a, dd b)
{
	dd p = two_prod(a.hi, b.hi);
	p.lo += a.hi * b.lo + a.lo * b.hi;
	return fast_two_sum(p.hi, p.lo);
}
MULLS_HD inline dd mul(dd a, double b)
{
	dd p = two_prod(a.hi, b);
	p.lo += a.lo * b;
	return fast_two_sum(p.hi, p.lo);
}
MULLS_HD inline dd div(dd a, dd b)
{
	const double q1 = a.hi / b.hi;
	dd r = add(a, neg(mul(b, q1)));
	const double q2 = r.hi / b.hi;
	r = add(r, neg(mul(b, q2)));
	const double q3 = r.hi / b.hi;
	return add(fast_two_sum(q1, q2), q3);
}
MULLS_HD inline dd sqrt_dd(dd a) // a > 0
{
	const double s = std::sqrt(a.hi);
	const dd r = add(a, neg(two_prod(s, s)));
	return fast_two_sum(s, r.hi / (2.0 * s));
}

// (+-) 1/k!, k = 0..33, with the sign the Taylor series of sin (odd k) and cos (even k) gives the term
MULLS_HD inline dd inv_fact(int k)
{
	static const dd t[34] = {
		{0x1.0000000000000p+0, 0x0.0p+0},
		{0x1.0000000000000p+0, 0x0.0p+0},
		{-0x1.0000000000000p-1, 0x0.0p+0},
		{-0x1.5555555555555p-3, -0x1.5555555555555p-57},
		{0x1.5555555555555p-5, 0x1.5555555555555p-59},
		{0x1.1111111111111p-7, 0x1.1111111111111p-63},
		{-0x1.6c16c16c16c17p-10, 0x1.f49f49f49f49fp-65},
		{-0x1.a01a01a01a01ap-13, -0x1.a01a01a01a01ap-73},
		{0x1.a01a01a01a01ap-16, 0x1.a01a01a01a01ap-76},
		{0x1.71de3a556c734p-19, -0x1.c154f8ddc6c00p-73},
		{-0x1.27e4fb7789f5cp-22, -0x1.cbbc05b4fa99ap-76},
		{-0x1.ae64567f544e4p-26, 0x1.c062e06d1f209p-80},
		{0x1.1eed8eff8d898p-29, -0x1.2aec959e14c06p-83},
		{0x1.6124613a86d09p-33, 0x1.f28e0cc748ebep-87},
		{-0x1.93974a8c07c9dp-37, -0x1.05d6f8a2efd1fp-92},
		{-0x1.ae7f3e733b81fp-41, -0x1.1d8656b0ee8cbp-97},
		{0x1.ae7f3e733b81fp-45, 0x1.1d8656b0ee8cbp-101},
		{0x1.952c77030ad4ap-49, 0x1.ac981465ddc6cp-103},
		{-0x1.6827863b97d97p-53, -0x1.eec01221a8b0bp-107},
		{-0x1.2f49b46814157p-57, -0x1.2650f61dbdcb4p-112},
		{0x1.e542ba4020225p-62, 0x1.ea72b4afe3c2fp-120},
		{0x1.71b8ef6dcf572p-66, -0x1.d043ae40c4647p-120},
		{-0x1.0ce396db7f853p-70, 0x1.aebcdbd20331cp-124},
		{-0x1.761b41316381ap-75, 0x1.3423c7d91404fp-130},
		{0x1.f2cf01972f578p-80, -0x1.9ada5fcc1ab14p-135},
		{0x1.3f3ccdd165fa9p-84, -0x1.58ddadf344487p-139},
		{-0x1.88e85fc6a4e5ap-89, 0x1.71c37ebd16540p-143},
		{-0x1.d1ab1c2dccea3p-94, -0x1.054d0c78aea14p-149},
		{0x1.0a18a2635085dp-98, 0x1.b9e2e28e1aa54p-153},
		{0x1.259f98b4358adp-103, 0x1.eaf8c39dd9bc5p-157},
		{-0x1.3932c5047d60ep-108, -0x1.832b7b530a627p-162},
		{-0x1.434d2e783f5bcp-113, -0x1.0b87b91be9affp-167},
		{0x1.434d2e783f5bcp-118, 0x1.0b87b91be9affp-172},
		{0x1.3981254dd0d52p-123, -0x1.2b1f4c8015a2fp-177}};
	return t[k];
}
// (-1)^k / (2k+1), k = 0..20 (atan)
MULLS_HD inline dd inv_odd(int k)
{
	static const dd t[21] = {
		{0x1.0000000000000p+0, 0x0.0p+0},
		{-0x1.5555555555555p-2, -0x1.5555555555555p-56},
		{0x1.999999999999ap-3, -0x1.999999999999ap-57},
		{-0x1.2492492492492p-3, -0x1.2492492492492p-57},
		{0x1.c71c71c71c71cp-4, 0x1.c71c71c71c71cp-58},
		{-0x1.745d1745d1746p-4, 0x1.745d1745d1746p-59},
		{0x1.3b13b13b13b14p-4, -0x1.3b13b13b13b14p-58},
		{-0x1.1111111111111p-4, -0x1.1111111111111p-60},
		{0x1.e1e1e1e1e1e1ep-5, 0x1.e1e1e1e1e1e1ep-61},
		{-0x1.af286bca1af28p-5, -0x1.af286bca1af28p-59},
		{0x1.8618618618618p-5, 0x1.8618618618618p-59},
		{-0x1.642c8590b2164p-5, -0x1.642c8590b2164p-60},
		{0x1.47ae147ae147bp-5, -0x1.eb851eb851eb8p-61},
		{-0x1.2f684bda12f68p-5, -0x1.2f684bda12f68p-59},
		{0x1.1a7b9611a7b96p-5, 0x1.1a7b9611a7b96p-61},
		{-0x1.0842108421084p-5, -0x1.0842108421084p-60},
		{0x1.f07c1f07c1f08p-6, -0x1.f07c1f07c1f08p-61},
		{-0x1.d41d41d41d41dp-6, -0x1.0750750750750p-60},
		{0x1.bacf914c1bad0p-6, -0x1.bacf914c1bad0p-60},
		{-0x1.a41a41a41a41ap-6, -0x1.0690690690690p-60},
		{0x1.8f9c18f9c18fap-6, -0x1.f3831f3831f38p-61}};
	return t[k];
}

// sin (which = 0) or cos (which = 1) of x
MULLS_HD inline double trig(double x, int which)
{
	if (!(std::fabs(x) <= 0x1p40)) // NaN, +-inf — and arguments beyond 2^40 rad, where the 159-bit pi/2 below no longer reduces exactly: defined
		return x - x + (x - x) / (x - x); // as NaN (an Euler angle of 1e12 rad is a diverged solve; its translation fails the step test first)
	if (x == 0.0)
		return which ? 1.0 : x; // sin keeps the sign of zero
	const double ax = std::fabs(x);
#ifndef MULLS_DETMATH_NO_QUICK // (tests/test_detmath.py builds the header both ways and demands equal bits)
	if (ax <= 0.5 && ax >= 0x1p-300)
	{
		// quick phase: x^2 exactly (s), the series' first correction in double-double (c1), the rest in double (c2)
		const dd s = two_prod(x, x);
		const double s1 = s.hi;
		double hi, lo, err;
		if (which == 0)
		{
			dd t = two_prod(x, s.hi); // x^3
			t.lo += x * s.lo;
			dd c1 = two_prod(t.hi, -0x1.5555555555555p-3); // -x^3 / 6
			c1.lo += t.hi * -0x1.5555555555555p-57 + t.lo * -0x1.5555555555555p-3;
			const double p = 0x1.1111111111111p-7 + s1 * (-0x1.a01a01a01a01ap-13 + s1 * (0x1.71de3a556c734p-19 + s1 * (-0x1.ae64567f544e4p-26 + s1 * (0x1.6124613a86d09p-33 + s1 * (-0x1.ae7f3e733b81fp-41 + s1 * 0x1.952c77030ad4ap-49)))));
			const double c2 = (t.hi * s1) * p; // x^5 / 5! - x^7 / 7! + ... + x^17 / 17!
			const dd y = fast_two_sum(x, c1.hi);
			hi = y.hi;
			lo = y.lo + (c1.lo + c2);
			err = std::fabs(c2) * 0x1p-48 + ax * 0x1p-100;
		}
		else
		{
			const double q = 0x1.5555555555555p-5 + s1 * (-0x1.6c16c16c16c17p-10 + s1 * (0x1.a01a01a01a01ap-16 + s1 * (-0x1.27e4fb7789f5cp-22 + s1 * (0x1.1eed8eff8d898p-29 + s1 * (-0x1.93974a8c07c9dp-37 + s1 * (0x1.ae7f3e733b81fp-45 + s1 * -0x1.6827863b97d97p-53))))));
			const double c2 = (s1 * s1) * q; // x^4 / 4! - x^6 / 6! + ... - x^18 / 18!
			const dd y = fast_two_sum(1.0, -0.5 * s.hi); // -x^2 / 2 is exact in (s.hi, s.lo)
			hi = y.hi;
			lo = y.lo + (-0.5 * s.lo + c2);
			err = c2 * 0x1p-48 + 0x1p-100;
		}
		const double out = hi + lo;
		if (out == hi + (lo + err) && out == hi + (lo - err))
			return out;
	}
#endif
	dd r = {x, 0.0};
	int quad = 0;
	if (ax > 0.78539816339744828) // beyond pi/4: r = x - k * pi/2 with pi/2 = P1 + P2 + P3 (159 bits)
	{
		const double k = std::nearbyint(x * 0x1.45f306dc9c883p-1); // round to nearest even, like the default rounding mode everywhere else
		const dd t1 = two_prod(k, 0x1.921fb54442d18p+0);
		r = two_sum(x, -t1.hi);
		r = add(r, -t1.lo);
		r = add(r, neg(two_prod(k, 0x1.1a62633145c07p-54)));
		r = add(r, -(k * -0x1.f1976b7ed8fbcp-110));
		const double km = k - 4.0 * std::floor(k * 0.25);
		quad = (int)km;
	}
	// sin(x) = (+sin r, +cos r, -sin r, -cos r)[quad], cos(x) = (+cos r, -sin r, -cos r, +sin r)[quad]
	const int sel = (quad + which) & 3; // 0: +sin r, 1: +cos r, 2: -sin r, 3: -cos r
	const bool want_cos = (sel & 1) != 0;
	int e;
	(void)std::frexp(r.hi, &e); // |r| < 2^e
	const int K = e <= -20 ? 7 : (e <= -10 ? 13 : (e <= -5 ? 19 : (e <= -2 ? 27 : 33))); // highest degree: the rest is below 2^-110 of the value
	const dd r2 = mul(r, r);
	dd v;
	if (want_cos)
	{
		int k = K & ~1; // highest even degree
		dd p = inv_fact(k);
		for (k -= 2; k >= 2; k -= 2)
			p = add(mul(p, r2), inv_fact(k));
		v = add(mul(p, r2), 1.0); // 1 - r^2/2 + ...
	}
	else
	{
		int k = (K & 1) ? K : K - 1; // highest odd degree
		dd p = inv_fact(k);
		for (k -= 2; k >= 3; k -= 2)
			p = add(mul(p, r2), inv_fact(k));
		v = add(mul(mul(p, r2), r), r); // r - r^3/6 + ...
	}
	const double out = v.hi + v.lo; // v is normalised: this is v.hi, the rounding of the double-double value
	return (sel & 2) ? -out : out;
}
MULLS_HD inline double sin_cr(double x) { return trig(x, 0); }
MULLS_HD inline double cos_cr(double x) { return trig(x, 1); }

// atan2(y, x) with the IEEE special cases of the C library function
MULLS_HD inline double atan2_cr(double y, double x)
{
	if (x != x || y != y)
		return x + y;
	const double PI_HI = 0x1.921fb54442d18p+1, PIO2_HI = 0x1.921fb54442d18p+0, PIO4_HI = 0x1.921fb54442d18p-1;
	const double ax = std::fabs(x), ay = std::fabs(y);
	const bool xneg = std::signbit(x);
	const double inf = 1.0 / 0.0;
	double res;
	if (ay == 0.0)
		res = xneg ? PI_HI : 0.0;
	else if (ax == 0.0)
		res = PIO2_HI;
	else if (ax == inf || ay == inf)
		res = (ax == inf && ay == inf) ? (xneg ? 3.0 * PIO4_HI : PIO4_HI) : (ay == inf ? PIO2_HI : (xneg ? PI_HI : 0.0));
	else
	{
#ifndef MULLS_DETMATH_NO_QUICK
		if (!xneg && ay <= 0.125 * ax && ay >= 0x1p-300 && ax >= 0x1p-300 && ax <= 0x1p300)
		{
			// quick phase (see trig): z = ay / ax as quotient + remainder, z^3 / 3 in double-double, the rest of the series in double
			const double q = ay / ax;
			if (q >= 0x1p-200)
			{
				const double ql = __builtin_fma(-q, ax, ay) / ax; // z = q + ql
				dd s = two_prod(q, q);						   // z^2
				s.lo += 2.0 * q * ql;
				dd t = two_prod(q, s.hi); // z^3
				t.lo += q * s.lo + ql * s.hi;
				dd c1 = two_prod(t.hi, -0x1.5555555555555p-2); // -z^3 / 3
				c1.lo += t.hi * -0x1.5555555555555p-56 + t.lo * -0x1.5555555555555p-2;
				const double s1 = s.hi;
				const double p = 0x1.999999999999ap-3 + s1 * (-0x1.2492492492492p-3 + s1 * (0x1.c71c71c71c71cp-4 + s1 * (-0x1.745d1745d1746p-4 + s1 * (0x1.3b13b13b13b14p-4 + s1 * (-0x1.1111111111111p-4 + s1 * (0x1.e1e1e1e1e1e1ep-5 + s1 * (-0x1.af286bca1af28p-5 + s1 * (0x1.8618618618618p-5 + s1 * (-0x1.642c8590b2164p-5 + s1 * (0x1.47ae147ae147bp-5 + s1 * -0x1.2f684bda12f68p-5))))))))));
				const double c2 = (t.hi * s1) * p; // z^5 / 5 - z^7 / 7 + ... - z^27 / 27
				const dd yv = fast_two_sum(q, c1.hi);
				const double lo = yv.lo + ((ql + c1.lo) + c2);
				const double err = c2 * 0x1p-48 + q * 0x1p-100;
				const double out = yv.hi + lo;
				if (out == yv.hi + (lo + err) && out == yv.hi + (lo - err))
					return std::signbit(y) ? -out : out;
			}
		}
#endif
		const bool swap = ay > ax;
		dd z = div(dd{swap ? ax : ay, 0.0}, dd{swap ? ay : ax, 0.0}); // in [0, 1]
		// halve the angle until tan is small: atan z = 2 atan(z / (1 + sqrt(1 + z^2)))
		int m = 0;
		while (z.hi > 0.1 && m < 4)
		{
			z = div(z, add(sqrt_dd(add(mul(z, z), 1.0)), 1.0));
			m++;
		}
		dd a;
		if (z.hi < 1e-150)
			a = z; // atan z = z to every bit we keep (and z^2 would underflow)
		else
		{
			int e;
			(void)std::frexp(z.hi, &e);
			const int n = e <= -20 ? 3 : (e <= -10 ? 6 : (e <= -5 ? 12 : 20));
			const dd z2 = mul(z, z);
			dd p = inv_odd(n);
			for (int k = n - 1; k >= 1; k--)
				p = add(mul(p, z2), inv_odd(k));
			a = add(mul(mul(p, z2), z), z); // z - z^3/3 + ...
		}
		for (int i = 0; i < m; i++)
			a = dd{2.0 * a.hi, 2.0 * a.lo};
		const dd pio2 = {0x1.921fb54442d18p+0, 0x1.1a62633145c07p-54}, pi = {0x1.921fb54442d18p+1, 0x1.1a62633145c07p-53};
		if (swap)
			a = add(pio2, neg(a));
		if (xneg)
			a = add(pi, neg(a));
		res = a.hi + a.lo;
	}
	return std::signbit(y) ? -res : res;
}
} // namespace det
} // namespace mulls
