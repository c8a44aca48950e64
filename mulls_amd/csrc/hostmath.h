// hostmath.h — the O(1)-per-iteration host algebra of MULLS-ICP (the 6x6 solve stays on the host by design).
//
// Reference semantics (include/common/cregistration.hpp): mirror + `ATPA.inverse() * ATPb` + cofactor Jacobian
// propagation :1924-1964, construct_trans_a :2740-2764, get_quat_euler_jacobi :2795-2819, the Eigen::AngleAxisd
// step-size test :1344-1348, information matrix :1386.  Eigen is not a dependency of this library: the few fixed-size
// operations are written out (column-major storage, partial-pivot LU like Eigen 3.3's PartialPivLU).
#pragma once
#include <cmath>
#include <cstring>

#include "detmath.h" // sin / cos / atan2 shared bit for bit by the host driver and the device-resident loop

namespace mulls
{

struct Mat4 // column-major 4x4
{
	double v[16];
	MULLS_HD double &at(int r, int c) { return v[r + 4 * c]; }
	MULLS_HD double at(int r, int c) const { return v[r + 4 * c]; }
	MULLS_HD static Mat4 identity()
	{
		Mat4 m;
		std::memset(m.v, 0, sizeof(m.v));
		m.v[0] = m.v[5] = m.v[10] = m.v[15] = 1.0;
		return m;
	}
};

MULLS_HD inline Mat4 operator*(const Mat4 &a, const Mat4 &b)
{
	Mat4 c;
	for (int col = 0; col < 4; col++)
		for (int row = 0; row < 4; row++)
		{
			double acc = 0.0;
			for (int k = 0; k < 4; k++)
				acc += a.at(row, k) * b.at(k, col);
			c.at(row, col) = acc;
		}
	return c;
}

struct Mat6 // column-major 6x6
{
	double v[36];
	MULLS_HD double &at(int r, int c) { return v[r + 6 * c]; }
	MULLS_HD double at(int r, int c) const { return v[r + 6 * c]; }
	MULLS_HD static Mat6 identity()
	{
		Mat6 m;
		std::memset(m.v, 0, sizeof(m.v));
		for (int i = 0; i < 6; i++)
			m.v[7 * i] = 1.0;
		return m;
	}
};

// inverse through a row-pivoted LU factorisation, solved against the identity column by column.
// Returns false when a zero pivot was met (the result then carries inf/NaN exactly like Eigen's would).
MULLS_HD inline bool invert6(const Mat6 &in, Mat6 &out)
{
	const int n = 6;
	MULLS_WORK double a[36];
	MULLS_WORK int row_of[6];
	for (int k = 0; k < 36; k++)
		a[k] = in.v[k];
	for (int i = 0; i < n; i++)
		row_of[i] = i;
	bool regular = true;
	for (int col = 0; col < n; col++)
	{
		int p = col;
		double big = std::fabs(a[col + n * col]);
		for (int r = col + 1; r < n; r++)
			if (std::fabs(a[r + n * col]) > big)
			{
				big = std::fabs(a[r + n * col]);
				p = r;
			}
		if (big == 0.0)
			regular = false;
		if (p != col)
		{
			for (int c = 0; c < n; c++)
			{
				double t = a[col + n * c];
				a[col + n * c] = a[p + n * c];
				a[p + n * c] = t;
			}
			int t = row_of[col];
			row_of[col] = row_of[p];
			row_of[p] = t;
		}
		const double piv = a[col + n * col];
		for (int r = col + 1; r < n; r++)
			a[r + n * col] /= piv;
		for (int c = col + 1; c < n; c++)
		{
			const double top = a[col + n * c];
			for (int r = col + 1; r < n; r++)
				a[r + n * c] -= a[r + n * col] * top;
		}
	}
	MULLS_WORK double y[6];
	for (int c = 0; c < n; c++)
	{
		for (int r = 0; r < n; r++)
			y[r] = (row_of[r] == c) ? 1.0 : 0.0;
		for (int r = 1; r < n; r++)
			for (int k = 0; k < r; k++)
				y[r] -= a[r + n * k] * y[k];
		for (int r = n - 1; r >= 0; r--)
		{
			for (int k = r + 1; k < n; k++)
				y[r] -= a[r + n * k] * y[k];
			y[r] /= a[r + n * r];
		}
		for (int r = 0; r < n; r++)
			out.v[r + n * c] = y[r];
	}
	return regular;
}

// general 4x4 inverse by the same row-pivoted LU (used for inverse(initial_guess) in the undistortion branch)
MULLS_HD inline Mat4 invert4(const Mat4 &in)
{
	const int n = 4;
	double a[16];
	int row_of[4];
	std::memcpy(a, in.v, sizeof(a));
	for (int i = 0; i < n; i++)
		row_of[i] = i;
	for (int col = 0; col < n; col++)
	{
		int p = col;
		double big = std::fabs(a[col + n * col]);
		for (int r = col + 1; r < n; r++)
			if (std::fabs(a[r + n * col]) > big)
			{
				big = std::fabs(a[r + n * col]);
				p = r;
			}
		if (p != col)
		{
			for (int c = 0; c < n; c++)
			{
				double t = a[col + n * c];
				a[col + n * c] = a[p + n * c];
				a[p + n * c] = t;
			}
			int t = row_of[col];
			row_of[col] = row_of[p];
			row_of[p] = t;
		}
		const double piv = a[col + n * col];
		for (int r = col + 1; r < n; r++)
			a[r + n * col] /= piv;
		for (int c = col + 1; c < n; c++)
		{
			const double top = a[col + n * c];
			for (int r = col + 1; r < n; r++)
				a[r + n * c] -= a[r + n * col] * top;
		}
	}
	Mat4 out;
	for (int c = 0; c < n; c++)
	{
		double y[4];
		for (int r = 0; r < n; r++)
			y[r] = (row_of[r] == c) ? 1.0 : 0.0;
		for (int r = 1; r < n; r++)
			for (int k = 0; k < r; k++)
				y[r] -= a[r + n * k] * y[k];
		for (int r = n - 1; r >= 0; r--)
		{
			for (int k = r + 1; k < n; k++)
				y[r] -= a[r + n * k] * y[k];
			y[r] /= a[r + n * r];
		}
		for (int r = 0; r < n; r++)
			out.v[r + n * c] = y[r];
	}
	return out;
}

// inverse of a 3x3 (column-major) by cofactors over the determinant, as Eigen 3.3 does for fixed 3x3 matrices
MULLS_HD inline void invert3(const double m[9], double out[9])
{
	auto M = [&](int r, int c) { return m[r + 3 * c]; };
	auto cof = [&](int i, int j) {
		const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
		return M(i1, j1) * M(i2, j2) - M(i1, j2) * M(i2, j1);
	};
	const double c0 = cof(0, 0), c1 = cof(1, 0), c2 = cof(2, 0);
	const double det = (c0 * M(0, 0) + c1 * M(1, 0)) + c2 * M(2, 0);
	const double invdet = 1.0 / det;
	out[0] = c0 * invdet, out[3] = c1 * invdet, out[6] = c2 * invdet;
	out[1] = cof(0, 1) * invdet, out[4] = cof(1, 1) * invdet, out[7] = cof(2, 1) * invdet;
	out[2] = cof(0, 2) * invdet, out[5] = cof(1, 2) * invdet, out[8] = cof(2, 2) * invdet;
}

// unit quaternion (w,x,y,z) of the upper-left 3x3 (Shepperd branches, as Eigen::Quaterniond(Matrix3d))
MULLS_HD inline void rotation_quaternion(const Mat4 &T, double q[4])
{
	const double r00 = T.at(0, 0), r11 = T.at(1, 1), r22 = T.at(2, 2);
	const double tr = r00 + r11 + r22;
	if (tr > 0.0)
	{
		double s = std::sqrt(tr + 1.0);
		q[0] = 0.5 * s;
		s = 0.5 / s;
		q[1] = (T.at(2, 1) - T.at(1, 2)) * s;
		q[2] = (T.at(0, 2) - T.at(2, 0)) * s;
		q[3] = (T.at(1, 0) - T.at(0, 1)) * s;
	}
	else
	{
		int i = 0;
		if (r11 > r00)
			i = 1;
		if (r22 > T.at(i, i))
			i = 2;
		const int j = (i + 1) % 3, k = (j + 1) % 3;
		double s = std::sqrt(T.at(i, i) - T.at(j, j) - T.at(k, k) + 1.0);
		double v[3];
		v[i] = 0.5 * s;
		s = 0.5 / s;
		q[0] = (T.at(k, j) - T.at(j, k)) * s;
		v[j] = (T.at(j, i) + T.at(i, j)) * s;
		v[k] = (T.at(k, i) + T.at(i, k)) * s;
		q[1] = v[0];
		q[2] = v[1];
		q[3] = v[2];
	}
}

// tx ty tz roll pitch yaw -> [Rz(yaw) Ry(pitch) Rx(roll) | t]
MULLS_HD inline Mat4 euler_step_to_matrix(const double x[6])
{
	const double sa = det::sin_cr(x[3]), ca = det::cos_cr(x[3]);
	const double sb = det::sin_cr(x[4]), cb = det::cos_cr(x[4]);
	const double sg = det::sin_cr(x[5]), cg = det::cos_cr(x[5]);
	Mat4 m;
	std::memset(m.v, 0, sizeof(m.v));
	m.at(0, 0) = cg * cb;
	m.at(0, 1) = -sg * ca + cg * sb * sa;
	m.at(0, 2) = sg * sa + cg * sb * ca;
	m.at(1, 0) = sg * cb;
	m.at(1, 1) = cg * ca + sg * sb * sa;
	m.at(1, 2) = -cg * sa + sg * sb * ca;
	m.at(2, 0) = -sb;
	m.at(2, 1) = cb * sa;
	m.at(2, 2) = cb * ca;
	m.at(0, 3) = x[0];
	m.at(1, 3) = x[1];
	m.at(2, 3) = x[2];
	m.at(3, 3) = 1.0;
	return m;
}

// rotation angle in [0, pi] of the upper-left 3x3, through the unit quaternion (what Eigen::AngleAxisd(R).angle() does)
MULLS_HD inline double rotation_angle(const Mat4 &T)
{
	const double r00 = T.at(0, 0), r11 = T.at(1, 1), r22 = T.at(2, 2);
	double qw, qx, qy, qz;
	const double tr = r00 + r11 + r22;
	if (tr > 0.0)
	{
		double s = std::sqrt(tr + 1.0);
		qw = 0.5 * s;
		s = 0.5 / s;
		qx = (T.at(2, 1) - T.at(1, 2)) * s;
		qy = (T.at(0, 2) - T.at(2, 0)) * s;
		qz = (T.at(1, 0) - T.at(0, 1)) * s;
	}
	else
	{
		int i = 0;
		if (r11 > r00)
			i = 1;
		if (r22 > T.at(i, i))
			i = 2;
		const int j = (i + 1) % 3, k = (j + 1) % 3;
		double s = std::sqrt(T.at(i, i) - T.at(j, j) - T.at(k, k) + 1.0);
		double q[3];
		q[i] = 0.5 * s;
		s = 0.5 / s;
		qw = (T.at(k, j) - T.at(j, k)) * s;
		q[j] = (T.at(j, i) + T.at(i, j)) * s;
		q[k] = (T.at(k, i) + T.at(i, k)) * s;
		qx = q[0];
		qy = q[1];
		qz = q[2];
	}
	const double vn = std::sqrt(qx * qx + qy * qy + qz * qz);
	return vn != 0.0 ? 2.0 * det::atan2_cr(vn, std::fabs(qw)) : 0.0;
}

// d(q1,q2,q3)/d(roll,pitch,yaw); the half-angle sines/cosines are float locals in the reference (:2797-2804)
MULLS_HD inline void quat_euler_jacobian(const double e[3], double J[3][3])
{
	const float sr = (float)det::sin_cr(0.5 * e[0]), sp = (float)det::sin_cr(0.5 * e[1]), sy = (float)det::sin_cr(0.5 * e[2]);
	const float cr = (float)det::cos_cr(0.5 * e[0]), cp = (float)det::cos_cr(0.5 * e[1]), cy = (float)det::cos_cr(0.5 * e[2]);
	J[0][0] = 0.5 * (cr * cp * cy + sr * sp * sy);
	J[0][1] = 0.5 * (-sr * sp * cy - cr * cp * sy);
	J[0][2] = 0.5 * (-sr * cp * sy - cr * sp * cy);
	J[1][0] = 0.5 * (-sr * sp * cy + cr * cp * sy);
	J[1][1] = 0.5 * (cr * cp * cy - sr * sp * sy);
	J[1][2] = 0.5 * (-cr * sp * sy + sr * cp * cy);
	J[2][0] = 0.5 * (-sr * cp * sy - cr * sp * cy);
	J[2][1] = 0.5 * (-cr * sp * sy - sr * cp * cy);
	J[2][2] = 0.5 * (cr * cp * cy + sr * sp * sy);
}

// x = N^-1 b ; cofactor = N^-1 with its rotational blocks propagated to quaternion space.  Returns false if the
// solve produced a non-finite step.
MULLS_HD inline bool solve_step(const Mat6 &N, const double b[6], double x[6], Mat6 &cofactor)
{
	MULLS_WORK Mat6 Ninv;
	bool ok = invert6(N, Ninv);
	for (int r = 0; r < 6; r++)
	{
		double acc = 0.0;
		for (int c = 0; c < 6; c++)
			acc += Ninv.at(r, c) * b[c];
		x[r] = acc;
	}
	MULLS_WORK double J[3][3];
	quat_euler_jacobian(x + 3, J);
	cofactor = Ninv;
	MULLS_WORK double rr[3][3], tr[3][3], rt[3][3], tmp[3][3];
	for (int r = 0; r < 3; r++)
		for (int c = 0; c < 3; c++)
		{
			rr[r][c] = Ninv.at(3 + r, 3 + c);
			tr[r][c] = Ninv.at(r, 3 + c);
			rt[r][c] = Ninv.at(3 + r, c);
		}
	for (int r = 0; r < 3; r++)
		for (int c = 0; c < 3; c++)
			tmp[r][c] = J[r][0] * rr[0][c] + J[r][1] * rr[1][c] + J[r][2] * rr[2][c];
	for (int r = 0; r < 3; r++)
		for (int c = 0; c < 3; c++)
		{
			cofactor.at(3 + r, 3 + c) = tmp[r][0] * J[c][0] + tmp[r][1] * J[c][1] + tmp[r][2] * J[c][2];
			cofactor.at(r, 3 + c) = tr[r][0] * J[c][0] + tr[r][1] * J[c][1] + tr[r][2] * J[c][2];
			cofactor.at(3 + r, c) = J[r][0] * rt[0][c] + J[r][1] * rt[1][c] + J[r][2] * rt[2][c];
		}
	for (int i = 0; i < 6; i++)
		ok = ok && std::isfinite(x[i]);
	return ok;
}

} // namespace mulls
