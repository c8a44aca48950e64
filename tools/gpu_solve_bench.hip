// gpu_solve_bench.hip — where the per-iteration 6x6 step (solve_wave.h: LU, back substitution, trigonometry, cofactor, step tests)
// spends its time on one wave.  Diagnostics only: built by tools/build_tools.sh into tools/_bin/solve_bench, run on the GPU box.
//
//   tools/_bin/solve_bench [workgroups] [repetitions]
//
// Every workgroup is one wave that solves the same well-conditioned system `repetitions` times (the step angles of a converging
// registration: a few milliradians); lane 0 charges the wall clock (100 MHz, wall_clock64) between the marks of solve_wave to nine sections.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define MULLS_SOLVE_MARK(k)                                               \
	do                                                                    \
	{                                                                     \
		if (threadIdx.x == 0)                                             \
		{                                                                 \
			const unsigned long long now_ = wall_clock64(); \
			s_sec[k] += now_ - s_last;                                    \
			s_last = now_;                                                \
		}                                                                 \
	} while (0)
__shared__ unsigned long long s_sec[9], s_last;
#include "../mulls_amd/csrc/solve_wave.h"

__global__ __launch_bounds__(64) void k_bench(const double *comb_in, int reps, double angle_scale, unsigned long long *out, double *x_out)
{
	__shared__ mulls::PairIter h;
	__shared__ SolveWs ws;
	__shared__ double comb[28];
	mulls::IcpConst K;
	K.max_iter_num = 1000000;
	K.converge_translation = 0.0f;
	K.converge_rotation = 0.0f;
	K.max_bearable_translation = 1e9f;
	K.max_bearable_rotation = 1e9f;
	K.dis_thre_unit = 1.0f;
	K.dis_thre_min = 0.5f;
	K.dis_thre_update_rate = 1.1f;
	K.min_neccessary_corr_ratio = 0.0f;
	K.sigma_thre = 1.0f;
	if (threadIdx.x < 27)
		comb[threadIdx.x] = comb_in[threadIdx.x] * (threadIdx.x >= 21 ? angle_scale : 1.0);
	if (threadIdx.x == 0)
	{
		double g[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
		mulls::pair_iter_init(h, g, K);
		for (int k = 0; k < 9; k++)
			s_sec[k] = 0;
	}
	__syncthreads();
	const unsigned long long t0 = wall_clock64();
	if (threadIdx.x == 0)
		s_last = t0;
	for (int r = 0; r < reps; r++)
	{
		solve_wave(h, K, comb, 3, ws);
		if (threadIdx.x == 0)
		{
			h.active = 1;
			h.want_residual = 0;
			h.done = 0;
			s_last = wall_clock64();
		}
	}
	const unsigned long long t1 = wall_clock64();
	if (threadIdx.x == 0)
	{
		for (int k = 0; k < 9; k++)
			out[(size_t)blockIdx.x * 10 + k] = s_sec[k];
		out[(size_t)blockIdx.x * 10 + 9] = t1 - t0;
		if (blockIdx.x == 0)
			for (int k = 0; k < 6; k++)
				x_out[k] = h.x[k];
	}
}

int main(int argc, char **argv)
{
	const int wgs = argc > 1 ? std::atoi(argv[1]) : 256, reps = argc > 2 ? std::atoi(argv[2]) : 200;
	// a symmetric positive definite system in the packed row-major-upper form of the library (21 + 6 terms)
	double comb[27];
	int k = 0;
	for (int r = 0; r < 6; r++)
		for (int c = r; c < 6; c++)
			comb[k++] = r == c ? 1000.0 + 37.0 * r : 11.0 / (1.0 + r + c);
	const double rhs[6] = {4.0, -2.5, 1.0, 3.1, -2.2, 5.7}; // steps of a few millimetres / milliradians
	for (int j = 0; j < 6; j++)
		comb[21 + j] = rhs[j];
	double *d_comb, *d_x;
	unsigned long long *d_out;
	if (hipMalloc(&d_comb, sizeof(comb)) != hipSuccess || hipMalloc(&d_x, 6 * sizeof(double)) != hipSuccess ||
		hipMalloc(&d_out, sizeof(unsigned long long) * 10 * (size_t)wgs) != hipSuccess || hipMemcpy(d_comb, comb, sizeof(comb), hipMemcpyHostToDevice) != hipSuccess)
	{
		std::printf("no device\n");
		return 1;
	}
	const char *names[9] = {"normal_from_row", "LU (6 pivots)", "back substitution", "x = inv * b", "trigonometry (12 lanes)", "J + TempTran", "cofactor",
							"rotation angle + tests", "guess update"};
	for (double scale : {1.0, 100.0})
	{
		hipLaunchKernelGGL(k_bench, dim3(wgs), dim3(64), 0, 0, d_comb, 5, scale, d_out, d_x); // warm-up
		hipLaunchKernelGGL(k_bench, dim3(wgs), dim3(64), 0, 0, d_comb, reps, scale, d_out, d_x);
		if (hipDeviceSynchronize() != hipSuccess)
		{
			std::printf("kernel failed\n");
			return 1;
		}
		std::vector<unsigned long long> out((size_t)wgs * 10);
		double x[6];
		(void)hipMemcpy(out.data(), d_out, out.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
		(void)hipMemcpy(x, d_x, sizeof(x), hipMemcpyDeviceToHost);
		// wall_clock64: the constant 100 MHz counter (10 ns ticks)
		std::printf("step angles %.4g %.4g %.4g rad, %d workgroups x %d solves\n", x[3], x[4], x[5], wgs, reps);
		double tot = 0;
		for (int s = 0; s < 10; s++)
		{
			double sum = 0;
			for (int w = 0; w < wgs; w++)
				sum += (double)out[(size_t)w * 10 + s];
			const double cyc = sum / wgs / reps;
			if (s < 9)
			{
				tot += cyc;
				std::printf("  %-26s %7.3f us\n", names[s], cyc / 100.0);
			}
			else
				std::printf("  %-26s %7.3f us (sections: %.3f)\n", "whole solve_wave", cyc / 100.0, tot / 100.0);
		}
	}
	return 0;
}
