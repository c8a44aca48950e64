// kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) for the MULLS-ICP hot path.
//
// Reference semantics implemented here (citations relative to the MULLS tree, include/common/):
//   k_clone_src   cloudblock_t::clone_feature + batch_transform_feature_points(initial_guess)   utility.hpp:524-550, cregistration.hpp:1183
//   k_crop        intersection_filter / bbx_filter (stable compaction of all 12 clouds)          cregistration.hpp:2894-2922, cfilter.hpp:950-981
//   k_nn          batch_transform_feature_points(TempTran) fused with the exact 1-NN search      cregistration.hpp:1260, :1740-1747
//   k_filter      duplicate rule, permanent source compaction, distance + direction rejectors    cregistration.hpp:1755-1830
//   k_accum       pt2pl / pt2li / pt2pt normal-equation terms, and the posterior residual pass   cregistration.hpp:1976-2275, :2546-2677
//   k_finish      fixed-order reduction of per-workgroup partials, per-iteration bookkeeping
//
// Numerics policy: every float/double operation order follows the reference's C++ expressions; the translation unit
// is compiled with -ffp-contract=off (the reference build has no FMA: CMakeLists.txt:43, no -march), float sqrt and
// division are IEEE-correct (hipcc default), accumulators are double.  No MFMA: this is a search plus a reduction.
//
// Launch geometry: 256-thread workgroups (4 wave64).  The search / filter / accumulate kernels share one static job
// table (one job = 512 consecutive source points of one feature class of one pair); dead source points keep their
// slot and are masked by a flag byte instead of being physically compacted, which makes the job table iteration-
// invariant and the whole batch advance with three launches per ICP iteration.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.h"

// Workgroups are handed to the 8 XCDs round-robin by blockIdx, and every XCD has its own L2.  Job tables are sorted by
// pair, so consecutive jobs gather from the same target clouds: this bijection gives XCD x the x-th eighth of the table,
// in order, instead of every eighth job — the jobs of one pair then share one L2 (affinity only, never correctness).
__device__ __forceinline__ uint32_t xcd_job(uint32_t b, uint32_t n)
{
	const uint32_t q = n >> 3, r = n & 7u, x = b & 7u;
	return x * q + min(x, r) + (b >> 3);
}

#pragma clang fp contract(off)

namespace
{

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t f2ord(float f)
{
	uint32_t u = __float_as_uint(f);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t k)
{
	uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
	return __uint_as_float(u);
}

__device__ __forceinline__ bool class_called(const RunParams &rp, const CloudDesc &d, int cls)
{
	// `if (used[c] && src.size() > 0) determine_corres(...)` (cregistration.hpp:1272-1292) combined with the
	// K_min = 3 early return inside it (:1727-1728, :1832-1833)
	return rp.used[cls] && d.alive_cur >= 3u && d.tgt_n >= 3u;
}

// wave64 + 4-wave workgroup sum of an unsigned count; result valid in every thread
__device__ __forceinline__ uint32_t block_sum_u32(uint32_t v, uint32_t *lds4)
{
	for (int off = 32; off > 0; off >>= 1)
		v += __shfl_down(v, off);
	int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	__syncthreads();
	if (lane == 0)
		lds4[wave] = v;
	__syncthreads();
	return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

} // namespace

// ---------------------------------------------------------------------------------------------------------------
// Setup 1: clone the staged source clouds into SoA and apply the initial guess (double math, float store); reduce
// the bounding box of the transformed ground / pillar / facade source clouds (cregistration.hpp:2912-2915).
__global__ __launch_bounds__(MULLS_BLOCK) void k_clone_src(const Job *__restrict__ jobs, const CloudDesc *__restrict__ descs,
															const PairSetup *__restrict__ setup, const float4 *__restrict__ stage,
															float4 *__restrict__ tmp_pos, float4 *__restrict__ tmp_nrm,
															uint32_t *__restrict__ bbox /* [pair][6] ordered keys */, RunParams rp)
{
	const Job job = jobs[blockIdx.x];
	const CloudDesc &d = descs[job.pair * MULLS_NC + job.cls];
	const uint32_t s = job.start + threadIdx.x;
	// motion undistortion regenerates the five non-vertex clouds from block2->pc_*_down (cregistration.hpp:1251-1253)
	const bool regen = rp.undistort && job.cls != 5;
	const uint32_t n_in = regen ? d.sd_n0 : d.src_n0;
	const bool in = s < n_in;
	const PairSetup &su = setup[job.pair];
	const double *G = su.guess;
	float x = 0, y = 0, z = 0;
	if (in)
	{
		const float4 *rec = stage + (size_t)((regen ? d.sd_stage : d.src_stage) + s) * 3;
		float4 a = rec[0], b = rec[1], c = rec[2]; // (x y z _) (nx ny nz _) (intensity curvature _ _)
		if (regen)
		{
			// CFilter::apply_motion_compensation(in, out, inverse(initial_guess)) (cfilter.hpp:493-516): the point is moved
			// by the fraction `curvature` (its time stamp in [0,1]) of the inverse guess — slerp from the identity
			// quaternion, linear translation — in double, stored as float; normals are copied unrotated.
			const float sc = c.y;
			if (!(sc < 0.0f || (double)sc > 1.0 - 0.0f))
			{
				const double t = (double)sc, one = 1.0 - 2.220446049250313e-16;
				const double dq = su.inv_q[0], absD = fabs(dq);
				double s0, s1;
				if (absD >= one)
				{
					s0 = 1.0 - t;
					s1 = t;
				}
				else
				{
					const double theta = acos(absD), sinTheta = sin(theta);
					s0 = sin((1.0 - t) * theta) / sinTheta;
					s1 = sin((t * theta)) / sinTheta;
				}
				if (dq < 0)
					s1 = -s1;
				const double qw = s0 + s1 * su.inv_q[0], qx = s1 * su.inv_q[1], qy = s1 * su.inv_q[2], qz = s1 * su.inv_q[3];
				const double vx = a.x, vy = a.y, vz = a.z;
				const double uvx = 2.0 * (qy * vz - qz * vy), uvy = 2.0 * (qz * vx - qx * vz), uvz = 2.0 * (qx * vy - qy * vx);
				const double rx = vx + qw * uvx + (qy * uvz - qz * uvy);
				const double ry = vy + qw * uvy + (qz * uvx - qx * uvz);
				const double rz = vz + qw * uvz + (qx * uvy - qy * uvx);
				a.x = (float)(rx + t * su.inv_t[0]);
				a.y = (float)(ry + t * su.inv_t[1]);
				a.z = (float)(rz + t * su.inv_t[2]);
			}
		}
		const int reps = (rp.undistort && job.cls == 5) ? 2 : 1; // the vertex cloud is not regenerated: it receives the guess twice
		float onx = b.x, ony = b.y, onz = b.z;					  // (cregistration.hpp:1183 and :1257)
		x = a.x, y = a.y, z = a.z;
		for (int rep = 0; rep < reps; rep++)
		{
			const double px = x, py = y, pz = z, nx = onx, ny = ony, nz = onz;
			x = (float)(G[0] * px + G[1] * py + G[2] * pz + G[3]);
			y = (float)(G[4] * px + G[5] * py + G[6] * pz + G[7]);
			z = (float)(G[8] * px + G[9] * py + G[10] * pz + G[11]);
			onx = (float)(G[0] * nx + G[1] * ny + G[2] * nz);
			ony = (float)(G[4] * nx + G[5] * ny + G[6] * nz);
			onz = (float)(G[8] * nx + G[9] * ny + G[10] * nz);
		}
		tmp_pos[d.src_off + s] = make_float4(x, y, z, c.x);
		tmp_nrm[d.src_off + s] = make_float4(onx, ony, onz, c.y);
	}
	if (job.cls == 0 || job.cls == 1 || job.cls == 2)
	{
		uint32_t k[6];
		k[0] = in ? f2ord(x) : 0xffffffffu;
		k[1] = in ? f2ord(y) : 0xffffffffu;
		k[2] = in ? f2ord(z) : 0xffffffffu;
		k[3] = in ? f2ord(x) : 0u;
		k[4] = in ? f2ord(y) : 0u;
		k[5] = in ? f2ord(z) : 0u;
		for (int off = 32; off > 0; off >>= 1)
			for (int j = 0; j < 3; j++)
			{
				k[j] = min(k[j], (uint32_t)__shfl_down(k[j], off));
				k[3 + j] = max(k[3 + j], (uint32_t)__shfl_down(k[3 + j], off));
			}
		if ((threadIdx.x & 63) == 0)
			for (int j = 0; j < 3; j++)
			{
				atomicMin(&bbox[job.pair * 6 + j], k[j]);
				atomicMax(&bbox[job.pair * 6 + 3 + j], k[3 + j]);
			}
	}
}

// the intersection box of get_cloud_pair_intersection (cfilter.hpp:2613-2655): union box of the transformed source clouds
// (ordered keys from k_clone_src) against block1->local_bound, padded by 1 m
namespace
{
__device__ __forceinline__ void crop_box(uint32_t pair, const uint32_t *__restrict__ bbox, const PairSetup *__restrict__ setup, double lo[3],
										  double hi[3])
{
	for (int k = 0; k < 3; k++)
	{
		uint32_t kmin = bbox[pair * 6 + k], kmax = bbox[pair * 6 + 3 + k];
		// an empty union keeps (+DBL_MAX, -DBL_MAX) like CloudUtility::merge_bbx (utility.hpp:867-884)
		double mmin = (kmin == 0xffffffffu && kmax == 0u) ? 1.7976931348623157e308 : (double)ord2f(kmin);
		double mmax = (kmin == 0xffffffffu && kmax == 0u) ? -1.7976931348623157e308 : (double)ord2f(kmax);
		double b1min = setup[pair].tgt_bound[k], b1max = setup[pair].tgt_bound[3 + k];
		const float pad = 1.0f;
		lo[k] = ((b1min > mmin) ? b1min : mmin) - pad; // get_intersection_bbx, utility.hpp:857-865
		hi[k] = ((b1max < mmax) ? b1max : mmax) + pad;
	}
}
// strict inequalities, float coordinate promoted to double (cfilter.hpp:959-961)
__device__ __forceinline__ bool crop_keep(const float4 &p, const double lo[3], const double hi[3])
{
	return (double)p.x > lo[0] && (double)p.x < hi[0] && (double)p.y > lo[1] && (double)p.y < hi[1] && (double)p.z > lo[2] && (double)p.z < hi[2];
}
// cells along one axis for an extent and a cell edge — the same float expression as grid_cell(), so that the largest
// coordinate lands in the last cell; an absurd extent (the bounding box only sees coordinates within 1e18 m)
// saturates instead of overflowing the conversion, and the caller then grows the cell edge until the grid fits
__device__ __forceinline__ uint32_t grid_dim(float extent, float inv_h)
{
	const float c = floorf(extent * inv_h);
	return c >= 0.0f ? (uint32_t)fminf(c, 4.0e9f) + 1u : 1u; // also false for NaN
}
// grid descriptor of a cropped target cloud with bounding box [lo3, hi3] and `running` points
__device__ __forceinline__ GridDesc make_grid(const float lo3[3], const float hi3[3], uint32_t running, const RunParams &rp, uint32_t pair,
											   uint32_t cls)
{
	GridDesc g;
	g.ox = lo3[0], g.oy = lo3[1], g.oz = lo3[2];
	g.h = rp.grid_h0;
	g.nx = g.ny = g.nz = 1;
	g.ncell = 0;
	g.wpr = 1;
	g.nocc = 0;
	uint32_t nwords = 1;
	if (running > 0 && rp.bm_h0 > 0.0f)
	{
		// global-memory tier: occupancy bitmap over fine cells; rows are padded to whole 64-cell words
		g.h = rp.bm_h0;
		if (rp.bm_auto) // points lie on surfaces: mean spacing ~ sqrt(footprint / count); measured optimum 0.25 m (1 M points) .. 0.7 m (5 k)
			g.h = fminf(fmaxf(sqrtf((hi3[0] - lo3[0]) * (hi3[1] - lo3[1]) / (float)running), rp.bm_h0), 2.8f * rp.bm_h0);
		for (;;)
		{
			g.inv_h = 1.0f / g.h;
			g.nx = grid_dim(hi3[0] - g.ox, g.inv_h);
			g.ny = grid_dim(hi3[1] - g.oy, g.inv_h);
			g.nz = grid_dim(hi3[2] - g.oz, g.inv_h);
			g.wpr = (g.nx + 63u) >> 6;
			if ((unsigned long long)g.ny * g.nz * g.wpr <= (unsigned long long)rp.grid_maxcells)
				break;
			g.h *= 1.25f;
		}
		nwords = g.ny * g.nz * g.wpr;
	}
	else if (running > 0)
		for (;;)
		{
			g.inv_h = 1.0f / g.h;
			// same float expression as grid_cell() so that the largest coordinate lands in the last cell
			g.nx = grid_dim(hi3[0] - g.ox, g.inv_h);
			g.ny = grid_dim(hi3[1] - g.oy, g.inv_h);
			g.nz = grid_dim(hi3[2] - g.oz, g.inv_h);
			if ((unsigned long long)g.nx * g.ny * g.nz <= (unsigned long long)rp.grid_maxcells)
				break;
			g.h *= 1.25f;
		}
	else
		g.inv_h = 1.0f;
	// cell tables are laid out over the USED classes only: slot = pair * n_used + rank of this class among them
	uint32_t n_used = 0, rank = 0;
	for (uint32_t c = 0; c < MULLS_NC; c++)
	{
		if (c < cls && rp.used[c])
			rank++;
		n_used += rp.used[c] ? 1u : 0u;
	}
	g.ncell = (running > 0 && rp.used[cls]) ? (rp.bm_h0 > 0.0f ? nwords : g.nx * g.ny * g.nz) : 0u;
	g.cell_off = (pair * n_used + rank) * (rp.cell_stride);
	return g;
}
} // namespace

// Setup 2: order-preserving compaction of one cloud by the intersection box.  One workgroup per (pair, class, side);
// side 0 = source (reads the SoA written by k_clone_src), side 1 = target (reads the staged AoS records).
__global__ __launch_bounds__(MULLS_BLOCK) void k_crop(CloudDesc *__restrict__ descs, const PairSetup *__restrict__ setup,
													   const uint32_t *__restrict__ bbox, const float4 *__restrict__ stage,
													   const float4 *__restrict__ tmp_pos, const float4 *__restrict__ tmp_nrm,
													   float4 *__restrict__ spos, float4 *__restrict__ snrm, float4 *__restrict__ tpos,
													   float4 *__restrict__ tnrm, uint8_t *__restrict__ flag, int32_t *__restrict__ match,
													   float *__restrict__ wd, RunParams rp, GridDesc *__restrict__ grids, uint32_t *__restrict__ big_box)
{
	const int crop = rp.crop;
	__shared__ uint32_t wave_cnt[4];
	__shared__ float box_red[4][6];
	const uint32_t pair = blockIdx.x / (MULLS_NC * 2);
	const uint32_t cls = (blockIdx.x / 2) % MULLS_NC;
	const uint32_t side = blockIdx.x & 1;
	CloudDesc &d = descs[pair * MULLS_NC + cls];
	const uint32_t n0 = side ? d.tgt_n0 : ((rp.undistort && cls != 5) ? d.sd_n0 : d.src_n0);
	const uint32_t off = side ? d.tgt_off : d.src_off;
	if (side && d.big_slot)
	{
		// cropped by k_crop_big_* (many workgroups); this one only arms the cloud's bounding-box keys
		if (threadIdx.x < 6)
			big_box[(d.big_slot - 1u) * 6u + threadIdx.x] = threadIdx.x < 3 ? 0xffffffffu : 0u;
		return;
	}
	double lo[3], hi[3];
	if (crop)
		crop_box(pair, bbox, setup, lo, hi);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t running = 0;
	float bmin[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, bmax[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
	for (uint32_t base = 0; base < n0; base += MULLS_BLOCK)
	{
		const uint32_t i = base + threadIdx.x;
		const bool in = i < n0;
		float4 p = make_float4(0, 0, 0, 0), q = make_float4(0, 0, 0, 0);
		if (in)
		{
			if (side)
			{
				const float4 *rec = stage + (size_t)(d.tgt_stage + i) * 3;
				float4 a = rec[0], b = rec[1], c = rec[2];
				p = make_float4(a.x, a.y, a.z, c.x);
				q = make_float4(b.x, b.y, b.z, c.y);
			}
			else
			{
				p = tmp_pos[off + i];
				q = tmp_nrm[off + i];
			}
		}
		bool keep = in;
		if (crop && in)
			keep = crop_keep(p, lo, hi);
		const unsigned long long bal = __ballot(keep);
		const uint32_t before = __popcll(bal & ((1ull << lane) - 1ull));
		__syncthreads();
		if (lane == 0)
			wave_cnt[wave] = __popcll(bal);
		__syncthreads();
		uint32_t wbase = 0, total = 0;
		for (int w = 0; w < 4; w++)
		{
			if (w < wave)
				wbase += wave_cnt[w];
			total += wave_cnt[w];
		}
		if (keep)
		{
			const uint32_t dst = off + running + wbase + before;
			if (side)
			{
				tpos[dst] = p;
				tnrm[dst] = q;
				if (fabsf(p.x) <= 1.0e18f && fabsf(p.y) <= 1.0e18f && fabsf(p.z) <= 1.0e18f) // the grid covers the finite points; others clamp into its border cells
				{
					bmin[0] = fminf(bmin[0], p.x), bmin[1] = fminf(bmin[1], p.y), bmin[2] = fminf(bmin[2], p.z);
					bmax[0] = fmaxf(bmax[0], p.x), bmax[1] = fmaxf(bmax[1], p.y), bmax[2] = fmaxf(bmax[2], p.z);
				}
			}
			else
			{
				spos[dst] = p;
				snrm[dst] = q;
				flag[dst] = MULLS_F_ALIVE;
				match[dst] = -1;
				wd[dst] = 0.0f;
			}
		}
		running += total;
	}
	if (side && grids)
	{
		// bounding box of the surviving target points -> uniform grid descriptor for the grid search tier
		for (int k = 0; k < 3; k++)
			for (int off = 32; off > 0; off >>= 1)
			{
				bmin[k] = fminf(bmin[k], __shfl_down(bmin[k], off));
				bmax[k] = fmaxf(bmax[k], __shfl_down(bmax[k], off));
			}
		__syncthreads();
		if (lane == 0)
			for (int k = 0; k < 3; k++)
			{
				box_red[wave][k] = bmin[k];
				box_red[wave][3 + k] = bmax[k];
			}
		__syncthreads();
		if (threadIdx.x == 0)
		{
			float lo3[3], hi3[3];
			for (int k = 0; k < 3; k++)
			{
				lo3[k] = fminf(fminf(box_red[0][k], box_red[1][k]), fminf(box_red[2][k], box_red[3][k]));
				hi3[k] = fmaxf(fmaxf(box_red[0][3 + k], box_red[1][3 + k]), fmaxf(box_red[2][3 + k], box_red[3][3 + k]));
			}
			grids[pair * MULLS_NC + cls] = make_grid(lo3, hi3, running, rp, pair, cls);
		}
	}
	if (threadIdx.x == 0)
	{
		if (side)
			d.tgt_n = running;
		else
		{
			d.src_n = running;
			d.alive_cur = running;
			d.alive_next = 0;
			d.n_matched = 0;
			d.valid_next = 0;
			d.n_valid = 0;
		}
	}
}

// Setup 2, target class clouds beyond MULLS_BIG_CLOUD points (scan-to-map against a large local map): one workgroup per
// 4096-point segment instead of one per cloud (a single CU's memory bandwidth made a 400 k-point crop take 1.7 ms).
// count -> per-cloud scan of the segment counts (+ bounding box -> grid descriptor) -> scatter; same stable order.
__global__ __launch_bounds__(MULLS_BLOCK) void k_crop_big_count(const Job *__restrict__ segs, const CloudDesc *__restrict__ descs,
																 const PairSetup *__restrict__ setup, const uint32_t *__restrict__ bbox,
																 const float4 *__restrict__ stage, RunParams rp, uint32_t *__restrict__ seg_cnt,
																 uint32_t *__restrict__ big_box)
{
	__shared__ uint32_t red4[4];
	const Job sg = segs[blockIdx.x]; // count = big slot
	const CloudDesc &d = descs[sg.pair * MULLS_NC + sg.cls];
	double lo[3], hi[3];
	if (rp.crop)
		crop_box(sg.pair, bbox, setup, lo, hi);
	uint32_t mine = 0;
	float bmin[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, bmax[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
	for (uint32_t k = 0; k < MULLS_SEG; k += MULLS_BLOCK)
	{
		const uint32_t i = sg.start + k + threadIdx.x;
		if (i < d.tgt_n0)
		{
			const float4 p = stage[(size_t)(d.tgt_stage + i) * 3];
			if (!rp.crop || crop_keep(p, lo, hi))
			{
				mine++;
				if (fabsf(p.x) <= 1.0e18f && fabsf(p.y) <= 1.0e18f && fabsf(p.z) <= 1.0e18f) // the grid covers the finite points; others clamp into its border cells
				{
					bmin[0] = fminf(bmin[0], p.x), bmin[1] = fminf(bmin[1], p.y), bmin[2] = fminf(bmin[2], p.z);
					bmax[0] = fmaxf(bmax[0], p.x), bmax[1] = fmaxf(bmax[1], p.y), bmax[2] = fmaxf(bmax[2], p.z);
				}
			}
		}
	}
	for (int k = 0; k < 3; k++)
	{
		for (int off = 32; off > 0; off >>= 1)
		{
			bmin[k] = fminf(bmin[k], __shfl_down(bmin[k], off));
			bmax[k] = fmaxf(bmax[k], __shfl_down(bmax[k], off));
		}
		if ((threadIdx.x & 63) == 0 && bmin[k] <= bmax[k])
		{
			atomicMin(&big_box[sg.count * 6u + k], f2ord(bmin[k]));
			atomicMax(&big_box[sg.count * 6u + 3 + k], f2ord(bmax[k]));
		}
	}
	const uint32_t total = block_sum_u32(mine, red4);
	if (threadIdx.x == 0)
		seg_cnt[blockIdx.x] = total;
}

// one wave per big cloud: segment counts -> segment bases, cloud size, grid descriptor
__global__ __launch_bounds__(64) void k_crop_big_scan(const Job *__restrict__ clouds, CloudDesc *__restrict__ descs, RunParams rp,
													   uint32_t *__restrict__ seg_cnt, const uint32_t *__restrict__ big_box,
													   GridDesc *__restrict__ grids)
{
	const Job bc = clouds[blockIdx.x]; // start = first segment, count = number of segments
	uint32_t running = 0;
	for (uint32_t base = 0; base < bc.count; base += 64)
	{
		const uint32_t s = base + threadIdx.x;
		const uint32_t v = s < bc.count ? seg_cnt[bc.start + s] : 0u;
		uint32_t incl = v;
		for (int off = 1; off < 64; off <<= 1)
		{
			const uint32_t o = __shfl_up(incl, off);
			if ((int)threadIdx.x >= off)
				incl += o;
		}
		if (s < bc.count)
			seg_cnt[bc.start + s] = running + incl - v;
		running += __shfl(incl, 63);
	}
	if (threadIdx.x == 0)
	{
		descs[bc.pair * MULLS_NC + bc.cls].tgt_n = running;
		if (grids)
		{
			float lo3[3], hi3[3];
			for (int k = 0; k < 3; k++)
			{
				lo3[k] = running ? ord2f(big_box[blockIdx.x * 6u + k]) : __builtin_inff();
				hi3[k] = running ? ord2f(big_box[blockIdx.x * 6u + 3 + k]) : -__builtin_inff();
			}
			grids[bc.pair * MULLS_NC + bc.cls] = make_grid(lo3, hi3, running, rp, bc.pair, bc.cls);
		}
	}
}

__global__ __launch_bounds__(MULLS_BLOCK) void k_crop_big_scatter(const Job *__restrict__ segs, const CloudDesc *__restrict__ descs,
																   const PairSetup *__restrict__ setup, const uint32_t *__restrict__ bbox,
																   const float4 *__restrict__ stage, RunParams rp,
																   const uint32_t *__restrict__ seg_base, float4 *__restrict__ tpos,
																   float4 *__restrict__ tnrm)
{
	__shared__ uint32_t wave_cnt[4];
	const Job sg = segs[blockIdx.x];
	const CloudDesc &d = descs[sg.pair * MULLS_NC + sg.cls];
	double lo[3], hi[3];
	if (rp.crop)
		crop_box(sg.pair, bbox, setup, lo, hi);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t running = seg_base[blockIdx.x];
	for (uint32_t k = 0; k < MULLS_SEG; k += MULLS_BLOCK)
	{
		const uint32_t i = sg.start + k + threadIdx.x;
		bool keep = false;
		float4 p = make_float4(0, 0, 0, 0), q = make_float4(0, 0, 0, 0);
		if (i < d.tgt_n0)
		{
			const float4 *rec = stage + (size_t)(d.tgt_stage + i) * 3;
			const float4 a = rec[0], b = rec[1], c = rec[2];
			p = make_float4(a.x, a.y, a.z, c.x);
			q = make_float4(b.x, b.y, b.z, c.y);
			keep = !rp.crop || crop_keep(p, lo, hi);
		}
		const unsigned long long bal = __ballot(keep);
		const uint32_t before = __popcll(bal & ((1ull << lane) - 1ull));
		__syncthreads();
		if (lane == 0)
			wave_cnt[wave] = __popcll(bal);
		__syncthreads();
		uint32_t wbase = 0, total = 0;
		for (int w = 0; w < 4; w++)
		{
			if (w < wave)
				wbase += wave_cnt[w];
			total += wave_cnt[w];
		}
		if (keep)
		{
			tpos[d.tgt_off + running + wbase + before] = p;
			tnrm[d.tgt_off + running + wbase + before] = q;
		}
		running += total;
	}
}

// Setup 2b (keep_less_source_points only): order-preserving in-place compaction of one cloud by a host-made keep mask
// (cregistration.hpp:2866-2892).  One workgroup per (pair, class, side); destinations never overtake unread sources.
__global__ __launch_bounds__(MULLS_BLOCK) void k_thin(CloudDesc *__restrict__ descs, const uint8_t *__restrict__ src_keep,
													   const uint8_t *__restrict__ tgt_keep, float4 *__restrict__ spos, float4 *__restrict__ snrm,
													   float4 *__restrict__ tpos, float4 *__restrict__ tnrm)
{
	__shared__ uint32_t wave_cnt[4];
	const uint32_t pair = blockIdx.x / (MULLS_NC * 2);
	const uint32_t cls = (blockIdx.x / 2) % MULLS_NC;
	const uint32_t side = blockIdx.x & 1;
	CloudDesc &d = descs[pair * MULLS_NC + cls];
	const uint32_t n0 = side ? d.tgt_n : d.src_n;
	const uint32_t off = side ? d.tgt_off : d.src_off;
	const uint8_t *keepm = (side ? tgt_keep : src_keep) + off;
	float4 *pos = side ? tpos : spos, *nrm = side ? tnrm : snrm;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t running = 0;
	for (uint32_t base = 0; base < n0; base += MULLS_BLOCK)
	{
		const uint32_t i = base + threadIdx.x;
		const bool in = i < n0;
		float4 p = make_float4(0, 0, 0, 0), q = make_float4(0, 0, 0, 0);
		bool keep = false;
		if (in)
		{
			p = pos[off + i];
			q = nrm[off + i];
			keep = keepm[i] != 0;
		}
		const unsigned long long bal = __ballot(keep);
		const uint32_t before = __popcll(bal & ((1ull << lane) - 1ull));
		__syncthreads(); // every load of this chunk has completed before any store of it
		if (lane == 0)
			wave_cnt[wave] = __popcll(bal);
		__syncthreads();
		uint32_t wbase = 0, total = 0;
		for (int w = 0; w < 4; w++)
		{
			if (w < wave)
				wbase += wave_cnt[w];
			total += wave_cnt[w];
		}
		if (keep)
		{
			const uint32_t dst = off + running + wbase + before;
			pos[dst] = p;
			nrm[dst] = q;
		}
		running += total;
	}
	if (threadIdx.x == 0)
	{
		if (side)
			d.tgt_n = running;
		else
		{
			d.src_n = running;
			d.alive_cur = running;
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------
// Correspondence search.  A workgroup is 2 wave64 (MULLS_NN_BLOCK = 128 lanes); each lane owns MULLS_NN_PTS = 4
// source points of the job's 512-point slice: it applies this iteration's rigid step (double math, float store, in
// place — the reference accumulates float rounding the same way), then scans the whole target class cloud.  Targets
// are streamed from HBM with coalesced 16-B loads and staged in LDS as three planar arrays X[], Y[], Z[], so that one
// ds_read_b128 (wave-uniform address -> broadcast) delivers one coordinate of FOUR targets: 6 LDS reads per 8
// targets against ~290 VALU instructions, which keeps the LDS pipe ~10 % busy and the kernel VALU-bound.
// Distances are FLANN's L2_Simple<float>: ((dx*dx)+(dy*dy))+(dz*dz), no FMA.  Per source point the scan keeps the
// running minimum and the first index of the 8-target group that produced it (one v_min3 chain + one compare per
// group instead of a compare/select pair per target); the exact target index — lowest index among bit-equal
// distances — is recovered afterwards by re-evaluating that one group from L2.
template <int NPTS>
__device__ __forceinline__ void nn_scan_group(const float *__restrict__ X, const float *__restrict__ Y, const float *__restrict__ Z, uint32_t j,
											   const float (&px)[NPTS], const float (&py)[NPTS], const float (&pz)[NPTS], float (&best)[NPTS],
											   uint32_t (&grp)[NPTS], uint32_t gidx)
{
	const float4 x0 = *reinterpret_cast<const float4 *>(X + j), x1 = *reinterpret_cast<const float4 *>(X + j + 4);
	const float4 y0 = *reinterpret_cast<const float4 *>(Y + j), y1 = *reinterpret_cast<const float4 *>(Y + j + 4);
	const float4 z0 = *reinterpret_cast<const float4 *>(Z + j), z1 = *reinterpret_cast<const float4 *>(Z + j + 4);
	const float tx[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
	const float ty[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
	const float tz[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
#pragma unroll
	for (int u = 0; u < NPTS; u++)
	{
		float dd[8];
#pragma unroll
		for (int v = 0; v < 8; v++)
		{
			const float dx = px[u] - tx[v], dy = py[u] - ty[v], dz = pz[u] - tz[v];
			dd[v] = (dx * dx + dy * dy) + dz * dz;
		}
		const float m = fminf(fminf(fminf(dd[0], dd[1]), fminf(dd[2], dd[3])), fminf(fminf(dd[4], dd[5]), fminf(dd[6], dd[7])));
		const bool upd = m < best[u]; // strict: an equal later distance never replaces an earlier one
		best[u] = upd ? m : best[u];
		grp[u] = upd ? gidx : grp[u];
	}
}

__global__ __launch_bounds__(MULLS_NN_BLOCK) void k_nn(const Job *__restrict__ jobs, CloudDesc *__restrict__ descs,
														const PairState *__restrict__ states, RunParams rp, float4 *__restrict__ spos,
														float4 *__restrict__ snrm, const float4 *__restrict__ tpos,
														const uint8_t *__restrict__ flag, int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2,
														unsigned long long *__restrict__ winner)
{
	__shared__ __attribute__((aligned(16))) float tileX[MULLS_TILE];
	__shared__ __attribute__((aligned(16))) float tileY[MULLS_TILE];
	__shared__ __attribute__((aligned(16))) float tileZ[MULLS_TILE];
	const Job job = jobs[blockIdx.x];
	const PairState &ps = states[job.pair];
	if (!ps.active || (rp.normal_shooting && (job.cls == 0 || job.cls == 2 || job.cls == 4)))
		return; // planar classes are served by k_nn_shoot while normal shooting is on
	CloudDesc &d = descs[job.pair * MULLS_NC + job.cls];
	const uint32_t src_n = d.src_n, tgt_n = d.tgt_n, alive_cur = d.alive_cur;
	const bool called = class_called(rp, d, job.cls);
	const float *__restrict__ tp = reinterpret_cast<const float *>(tpos + d.tgt_off);

	uint32_t s[MULLS_NN_PTS];
	bool alive[MULLS_NN_PTS];
	float px[MULLS_NN_PTS], py[MULLS_NN_PTS], pz[MULLS_NN_PTS];
#pragma unroll
	for (int u = 0; u < MULLS_NN_PTS; u++)
	{
		s[u] = job.start + threadIdx.x + u * MULLS_NN_BLOCK;
		alive[u] = s[u] < src_n && (flag[d.src_off + s[u]] & MULLS_F_ALIVE);
		px[u] = py[u] = pz[u] = 0.0f;
		if (alive[u])
		{
			// pcl::transformPointCloudWithNormals<PointT,double> (cregistration.hpp:1690-1695)
			float4 p = spos[d.src_off + s[u]], n = snrm[d.src_off + s[u]];
			const double *T = ps.T;
			double x = p.x, y = p.y, z = p.z, nx = n.x, ny = n.y, nz = n.z;
			px[u] = (float)(T[0] * x + T[1] * y + T[2] * z + T[3]);
			py[u] = (float)(T[4] * x + T[5] * y + T[6] * z + T[7]);
			pz[u] = (float)(T[8] * x + T[9] * y + T[10] * z + T[11]);
			float onx = (float)(T[0] * nx + T[1] * ny + T[2] * nz);
			float ony = (float)(T[4] * nx + T[5] * ny + T[6] * nz);
			float onz = (float)(T[8] * nx + T[9] * ny + T[10] * nz);
			spos[d.src_off + s[u]] = make_float4(px[u], py[u], pz[u], p.w);
			snrm[d.src_off + s[u]] = make_float4(onx, ony, onz, n.w);
		}
	}
	if (!called)
		return; // correspondences of the previous iteration stay in force

	const float INF = __builtin_inff();
	float best[MULLS_NN_PTS];
	uint32_t grp[MULLS_NN_PTS]; // first target index of the winning 8-group
#pragma unroll
	for (int u = 0; u < MULLS_NN_PTS; u++)
	{
		best[u] = INF;
		grp[u] = 0;
	}

	for (uint32_t base = 0; base < tgt_n; base += MULLS_TILE)
	{
		const uint32_t nt = min((uint32_t)MULLS_TILE, tgt_n - base);
		const uint32_t nt8 = (nt + 7u) & ~7u;
		__syncthreads();
		for (uint32_t k = threadIdx.x; k < nt8; k += MULLS_NN_BLOCK)
		{
			// +inf padding keeps the unrolled scan free of tail code: (p - inf)^2 = inf never beats a finite minimum
			const float4 t = (k < nt) ? tpos[d.tgt_off + base + k] : make_float4(INF, INF, INF, 0.0f);
			tileX[k] = t.x;
			tileY[k] = t.y;
			tileZ[k] = t.z;
		}
		__syncthreads();
		for (uint32_t j = 0; j < nt8; j += 8)
			nn_scan_group<MULLS_NN_PTS>(tileX, tileY, tileZ, j, px, py, pz, best, grp, base + j);
	}

	// recover the exact index inside the winning group (bit-identical re-evaluation)
	const float r = 2.5f * ps.thr[job.cls];	 // filter_dis_times * dis_thre (float), cregistration.hpp:1745
	const double maxd = (double)r;			 // widened to the `double max_distance` parameter
	const double max_dist_sqr = maxd * maxd; // CorrespondenceEstimation::determineCorrespondences
	const bool gate = alive_cur >= 500u;	 // K_filter_distant_point, cregistration.hpp:1755
	const unsigned long long key_hi = (unsigned long long)(0xffffffffu - (rp.tick_base + (uint32_t)ps.iter)) << 32;
	uint32_t matched_cnt = 0;
#pragma unroll
	for (int u = 0; u < MULLS_NN_PTS; u++)
	{
		if (!alive[u])
			continue;
		int idx = -1;
		for (int v = 7; v >= 0; v--)
		{
			const uint32_t t = grp[u] + v;
			if (t < tgt_n)
			{
				float ddx = px[u] - tp[4 * t], ddy = py[u] - tp[4 * t + 1], ddz = pz[u] - tp[4 * t + 2];
				float dist = (ddx * ddx + ddy * ddy) + ddz * ddz;
				if (dist == best[u])
					idx = (int)t;
			}
		}
		const bool matched = idx >= 0 && !((double)best[u] > max_dist_sqr);
		nn_idx[d.src_off + s[u]] = matched ? idx : -1;
		nn_d2[d.src_off + s[u]] = best[u];
		if (matched)
		{
			matched_cnt++;
			if (gate) // duplicate rule: the lowest source index claims the target (first-come in the reference's serial walk)
				atomicMin(&winner[d.tgt_off + idx], key_hi | (unsigned long long)s[u]);
		}
	}
	for (int off = 32; off > 0; off >>= 1)
		matched_cnt += __shfl_down(matched_cnt, off);
	if ((threadIdx.x & 63) == 0 && matched_cnt)
		atomicAdd(&d.n_matched, matched_cnt);
}

// ---------------------------------------------------------------------------------------------------------------
// Exact fixed-radius search tier on a uniform grid.  The reference discards every match farther than 2.5*dis_thre
// (cregistration.hpp:1745), so visiting only the cells that intersect the search ball returns the same nearest
// neighbour as the kd-tree — provided no candidate inside the ball can be skipped.  That holds by construction:
// the cell coordinate is a monotone function of the float coordinate (subtract origin, scale, floor, clamp), so every
// target with |t.x - p.x| <= R lies in a cell between cell(p.x - R) and cell(p.x + R); R carries a relative 1e-4 and
// an absolute 1e-4 m margin over the float rounding of the distance arithmetic.  Ties on the float distance resolve to
// the lowest original target index explicitly (the cell-sorted order is not index order).
namespace
{
__device__ __forceinline__ int grid_cell(float v, float o, float inv_h, uint32_t n)
{
	const float c = floorf((v - o) * inv_h);
	return (int)fminf(fmaxf(c, 0.0f), (float)(n - 1u));
}
__device__ __forceinline__ uint32_t grid_cell_id(const GridDesc &g, float x, float y, float z)
{
	return ((uint32_t)grid_cell(z, g.oz, g.inv_h, g.nz) * g.ny + (uint32_t)grid_cell(y, g.oy, g.inv_h, g.ny)) * g.nx +
		   (uint32_t)grid_cell(x, g.ox, g.inv_h, g.nx);
}
// Evaluate every target in the cells intersecting the cube [p - R, p + R].  The MULLS_GRID_GROUP (= 16) lanes of a
// sub-group share one query.  Rows (fixed cy,cz; cells x0..x1 are one contiguous range of the cell-sorted array) are
// taken 16 at a time: lane j fetches the bounds of row j (occupancy words and ranks, then — only for rows that hold points —
// the two start positions: 16 rows per two memory latencies), then every lane issues its
// first candidate load of 8 rows back to back (coalesced 256-B segments), and only rows holding more than 16
// candidates loop further.  Chunks whose 16 rows are all empty cost one latency and no candidate work.
struct BmGrid
{
	const unsigned long long *bm; // [nw] occupancy words of this cloud
	const uint32_t *pf;			  // [nw] occupied cells before each word
	const uint32_t *cs;			  // [nocc + 1] first sorted position of every occupied cell
};
// sorted candidates [lo, hi) of cells xa..xb of the row whose first word is `rowword` (cells of a row are consecutive in the
// cell-sorted order, occupied or not): ranks of the first and one-past-last occupied cell, then their start positions
__device__ __forceinline__ void bm_row_range(const BmGrid &B, uint32_t rowword, uint32_t xa, uint32_t xb, uint32_t &lo, uint32_t &hi)
{
	const uint32_t wa = rowword + (xa >> 6), wb = rowword + (xb >> 6);
	const unsigned long long A = B.bm[wa], Z = B.bm[wb];
	const uint32_t r0 = B.pf[wa] + (uint32_t)__popcll(A & ((1ull << (xa & 63u)) - 1ull));
	const uint32_t r1 = B.pf[wb] + (uint32_t)__popcll(Z & (~0ull >> (63u - (xb & 63u))));
	lo = hi = 0;
	if (r1 > r0)
	{
		lo = B.cs[r0];
		hi = B.cs[r1];
	}
}
__device__ __forceinline__ void grid_scan_box(const GridDesc &g, const BmGrid &B, const float4 *__restrict__ ts,
											   float px, float py, float pz, float R, uint32_t sub, float &best, int &bi)
{
	const float Rm = R * 1.0001f + 1e-4f;
	const int x0 = grid_cell(px - Rm, g.ox, g.inv_h, g.nx), x1 = grid_cell(px + Rm, g.ox, g.inv_h, g.nx);
	const int y0 = grid_cell(py - Rm, g.oy, g.inv_h, g.ny), y1 = grid_cell(py + Rm, g.oy, g.inv_h, g.ny);
	const int z0 = grid_cell(pz - Rm, g.oz, g.inv_h, g.nz), z1 = grid_cell(pz + Rm, g.oz, g.inv_h, g.nz);
	const int nyc = y1 - y0 + 1, nrows = nyc * (z1 - z0 + 1);
	const uint32_t gshift = (threadIdx.x & 63u) & ~(MULLS_GRID_GROUP - 1u); // first lane of this sub-group inside its wave
	for (int base = 0; base < nrows; base += (int)MULLS_GRID_GROUP)
	{
		const int j = base + (int)sub;
		uint32_t lo = 0, hi = 0;
		if (j < nrows)
		{
			const uint32_t rowword = ((uint32_t)(z0 + j / nyc) * g.ny + (uint32_t)(y0 + j % nyc)) * g.wpr;
			bm_row_range(B, rowword, (uint32_t)x0, (uint32_t)x1, lo, hi);
		}
		const uint32_t nonempty = (uint32_t)(__ballot(hi > lo) >> gshift) & 0xffffu; // per sub-group: which of its 16 rows hold points
		if (!nonempty)
			continue;
#pragma unroll
		for (int half = 0; half < 2; half++) // 8 rows at a time keeps the register footprint at 8 float4 of loads in flight
		{
			if (!((nonempty >> (8 * half)) & 0xffu))
				continue;
			float4 q[8];
			bool ok[8];
#pragma unroll
			for (int jj = 0; jj < 8; jj++)
			{
				const uint32_t t = __shfl(lo, 8 * half + jj, MULLS_GRID_GROUP) + sub;
				ok[jj] = t < __shfl(hi, 8 * half + jj, MULLS_GRID_GROUP);
				if (ok[jj])
					q[jj] = ts[t];
			}
#pragma unroll
			for (int jj = 0; jj < 8; jj++)
				if (ok[jj])
				{
					const float dx = px - q[jj].x, dy = py - q[jj].y, dz = pz - q[jj].z;
					const float dist = (dx * dx + dy * dy) + dz * dz; // L2_Simple<float>, no FMA
					const int idx = __float_as_int(q[jj].w);
					if (dist < best || (dist == best && idx < bi))
					{
						best = dist;
						bi = idx;
					}
				}
		}
		// rows holding more than 16 candidates: four loads in flight per lane and trip
		for (uint32_t rem = nonempty; rem; rem &= rem - 1u)
		{
			const int jj = __ffs((int)rem) - 1;
			const uint32_t lo_j = __shfl(lo, jj, MULLS_GRID_GROUP), hi_j = __shfl(hi, jj, MULLS_GRID_GROUP);
			for (uint32_t t = lo_j + sub + MULLS_GRID_GROUP; t < hi_j; t += 4 * MULLS_GRID_GROUP)
			{
				float4 c[4];
				bool v[4];
#pragma unroll
				for (int w = 0; w < 4; w++)
				{
					v[w] = t + w * MULLS_GRID_GROUP < hi_j;
					if (v[w])
						c[w] = ts[t + w * MULLS_GRID_GROUP];
				}
#pragma unroll
				for (int w = 0; w < 4; w++)
					if (v[w])
					{
						const float dx = px - c[w].x, dy = py - c[w].y, dz = pz - c[w].z;
						const float dist = (dx * dx + dy * dy) + dz * dz;
						const int idx = __float_as_int(c[w].w);
						if (dist < best || (dist == best && idx < bi))
						{
							best = dist;
							bi = idx;
						}
					}
			}
		}
	}
}
// lexicographic (distance, original index) minimum over the lanes of one sub-group; result in every lane
__device__ __forceinline__ void group_min(float &best, int &bi)
{
#pragma unroll
	for (int mask = MULLS_GRID_GROUP / 2; mask > 0; mask >>= 1)
	{
		const float ob = __shfl_xor(best, mask, MULLS_GRID_GROUP);
		const int oi = __shfl_xor(bi, mask, MULLS_GRID_GROUP);
		const bool take = oi >= 0 && (bi < 0 || ob < best || (ob == best && oi < bi));
		best = take ? ob : best;
		bi = take ? oi : bi;
	}
}
} // namespace

// Global-memory tier grid build (clouds too large for LDS).  Fine cells make a dense cell table impossible (a 1 M-point
// map at 0.35 m cells spans ~50 M cells, of which <1 % hold points), so the grid is an occupancy bitmap + ranks:
//   bm  one bit per cell, 64 cells of a row per word          k_bm_mark   (atomicOr per point)
//   pf  per word: occupied cells before it                    k_bm_scan   (one workgroup per cloud)
//   cs  per occupied cell: first sorted position (+ end)      k_bm_count -> k_bm_starts -> k_bm_scatter (counting sort by rank)
// cs / cnt are indexed from cs_off = tgt_off + cloud index (every cloud needs tgt_n + 1 entries).
namespace
{
__device__ __forceinline__ uint32_t bm_bit(const GridDesc &g, float x, float y, float z)
{
	const uint32_t row = (uint32_t)grid_cell(z, g.oz, g.inv_h, g.nz) * g.ny + (uint32_t)grid_cell(y, g.oy, g.inv_h, g.ny);
	return row * (g.wpr * 64u) + (uint32_t)grid_cell(x, g.ox, g.inv_h, g.nx);
}
__device__ __forceinline__ uint32_t bm_rank(const unsigned long long *bm, const uint32_t *pf, uint32_t bit)
{
	return pf[bit >> 6] + (uint32_t)__popcll(bm[bit >> 6] & ((1ull << (bit & 63u)) - 1ull));
}
} // namespace

// only the words a cloud's grid really uses are cleared (the arena reserves the worst case per cloud)
__global__ __launch_bounds__(MULLS_BLOCK) void k_bm_clear(const GridDesc *__restrict__ grids, RunParams rp, unsigned long long *__restrict__ bm)
{
	if (!rp.used[blockIdx.x % MULLS_NC])
		return;
	const GridDesc g = grids[blockIdx.x];
	for (uint32_t w = blockIdx.y * MULLS_BLOCK + threadIdx.x; w < g.ncell; w += gridDim.y * MULLS_BLOCK)
		bm[g.cell_off + w] = 0ull;
}

__global__ __launch_bounds__(MULLS_BLOCK) void k_bm_mark(const Job *__restrict__ tjobs, const CloudDesc *__restrict__ descs,
														  const GridDesc *__restrict__ grids, const float4 *__restrict__ tpos,
														  unsigned long long *__restrict__ bm)
{
	const Job job = tjobs[blockIdx.x];
	const CloudDesc &d = descs[job.pair * MULLS_NC + job.cls];
	const uint32_t t = job.start + threadIdx.x;
	if (t >= d.tgt_n)
		return;
	const GridDesc g = grids[job.pair * MULLS_NC + job.cls];
	const float4 p = tpos[d.tgt_off + t];
	const uint32_t bit = bm_bit(g, p.x, p.y, p.z);
	unsigned long long *word = &bm[g.cell_off + (bit >> 6)];
	const unsigned long long b = 1ull << (bit & 63u);
	if (!(__builtin_nontemporal_load(word) & b)) // dense maps put tens of points in a cell: most find their bit set already
		atomicOr(word, b);
}

// exclusive scan of a uint32 sequence produced by `value(i)`, one 1024-lane workgroup, four items per lane and trip
template <typename F, typename G>
__device__ __forceinline__ uint32_t block_scan_1024(uint32_t n, F value, G store)
{
	__shared__ uint32_t wave_tot[16];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t running = 0;
	for (uint32_t base = 0; base < n; base += 4096u)
	{
		const uint32_t i0 = base + threadIdx.x * 4u;
		uint32_t v[4], sum = 0;
		for (int k = 0; k < 4; k++)
		{
			v[k] = (i0 + k < n) ? value(i0 + k) : 0u;
			sum += v[k];
		}
		uint32_t incl = sum;
		for (int off = 1; off < 64; off <<= 1)
		{
			const uint32_t o = __shfl_up(incl, off);
			if (lane >= off)
				incl += o;
		}
		__syncthreads();
		if (lane == 63)
			wave_tot[wave] = incl;
		__syncthreads();
		uint32_t wbase = 0, total = 0;
		for (int w = 0; w < 16; w++)
		{
			if (w < wave)
				wbase += wave_tot[w];
			total += wave_tot[w];
		}
		uint32_t ex = running + wbase + incl - sum;
		for (int k = 0; k < 4; k++)
			if (i0 + k < n)
			{
				store(i0 + k, ex);
				ex += v[k];
			}
		running += total;
	}
	return running;
}

__global__ __launch_bounds__(1024) void k_bm_scan(GridDesc *__restrict__ grids, RunParams rp, const unsigned long long *__restrict__ bm,
												  uint32_t *__restrict__ pf)
{
	const uint32_t cls = blockIdx.x % MULLS_NC;
	if (!rp.used[cls])
		return;
	const GridDesc g = grids[blockIdx.x];
	const unsigned long long *b = bm + g.cell_off;
	uint32_t *p = pf + g.cell_off;
	const uint32_t nocc = block_scan_1024(
		g.ncell, [&](uint32_t w) { return (uint32_t)__popcll(b[w]); }, [&](uint32_t w, uint32_t ex) { p[w] = ex; });
	if (threadIdx.x == 0)
		grids[blockIdx.x].nocc = nocc;
}

__global__ __launch_bounds__(MULLS_BLOCK) void k_bm_count(const Job *__restrict__ tjobs, const CloudDesc *__restrict__ descs,
														   const GridDesc *__restrict__ grids, const float4 *__restrict__ tpos,
														   const unsigned long long *__restrict__ bm, const uint32_t *__restrict__ pf,
														   uint32_t *__restrict__ cnt)
{
	const Job job = tjobs[blockIdx.x];
	const uint32_t ci = job.pair * MULLS_NC + job.cls;
	const CloudDesc &d = descs[ci];
	const uint32_t t = job.start + threadIdx.x;
	if (t >= d.tgt_n)
		return;
	const GridDesc g = grids[ci];
	const float4 p = tpos[d.tgt_off + t];
	atomicAdd(&cnt[d.tgt_off + ci + bm_rank(bm + g.cell_off, pf + g.cell_off, bm_bit(g, p.x, p.y, p.z))], 1u);
}

// counts -> start positions; the counters are left at zero so that k_bm_scatter can reuse them as insertion cursors
__global__ __launch_bounds__(1024) void k_bm_starts(const CloudDesc *__restrict__ descs, const GridDesc *__restrict__ grids, RunParams rp,
													uint32_t *__restrict__ cnt, uint32_t *__restrict__ cs)
{
	const uint32_t cls = blockIdx.x % MULLS_NC;
	if (!rp.used[cls])
		return;
	const GridDesc g = grids[blockIdx.x];
	const uint32_t off = descs[blockIdx.x].tgt_off + blockIdx.x;
	uint32_t *c = cnt + off, *s = cs + off;
	const uint32_t total = block_scan_1024(
		g.nocc, [&](uint32_t r) { return c[r]; },
		[&](uint32_t r, uint32_t ex) {
			s[r] = ex;
			c[r] = 0u;
		});
	if (threadIdx.x == 0)
		s[g.nocc] = total;
}

// counting-sort scatter: target positions ordered by cell, original index carried in .w
__global__ __launch_bounds__(MULLS_BLOCK) void k_bm_scatter(const Job *__restrict__ tjobs, const CloudDesc *__restrict__ descs,
															 const GridDesc *__restrict__ grids, const float4 *__restrict__ tpos,
															 const unsigned long long *__restrict__ bm, const uint32_t *__restrict__ pf,
															 uint32_t *__restrict__ cnt, const uint32_t *__restrict__ cs, float4 *__restrict__ tsorted)
{
	const Job job = tjobs[blockIdx.x];
	const uint32_t ci = job.pair * MULLS_NC + job.cls;
	const CloudDesc &d = descs[ci];
	const uint32_t t = job.start + threadIdx.x;
	if (t >= d.tgt_n)
		return;
	const GridDesc g = grids[ci];
	const float4 p = tpos[d.tgt_off + t];
	const uint32_t r = d.tgt_off + ci + bm_rank(bm + g.cell_off, pf + g.cell_off, bm_bit(g, p.x, p.y, p.z));
	const uint32_t slot = cs[r] + atomicAdd(&cnt[r], 1u);
	tsorted[d.tgt_off + slot] = make_float4(p.x, p.y, p.z, __int_as_float((int)t));
}

// ---------------------------------------------------------------------------------------------------------------
// LDS tier grid build: one workgroup per target class cloud (<= MULLS_LDS_MAXPTS points, <= MULLS_MAXCELLS cells).  Grids of
// fewer than 16384 cells (the usual case) are counting-sorted with LDS atomics; otherwise the
// cloud is sorted by (cell id, original index) with a bitonic network in LDS — packed 32-bit keys, cell id < 2^16,
// index < 2^14 — then the cell table is filled by binary search of every cell id in the sorted keys.  Replaces the
// count / scan / scatter kernels (global atomics on every point, 1.2 ms for 3072 clouds) for clouds that fit; the
// cell-sorted order also becomes deterministic (index order inside a cell).
__global__ __launch_bounds__(MULLS_LDS_BLOCK) void k_grid_build_sort(const CloudDesc *__restrict__ descs, const GridDesc *__restrict__ grids,
																	  RunParams rp, const float4 *__restrict__ tpos,
																	  uint32_t *__restrict__ cell_start, float4 *__restrict__ tsorted)
{
	__shared__ uint32_t K[16384];
	const uint32_t cls = blockIdx.x % MULLS_NC;
	if (!rp.used[cls])
		return;
	const CloudDesc &d = descs[blockIdx.x];
	const GridDesc g = grids[blockIdx.x];
	const uint32_t n = d.tgt_n;
	if (g.ncell == 0)
		return;
	if (g.ncell < 16384u)
	{
		// few enough cells for an on-chip histogram: counting sort with LDS atomics (a fraction of the bitonic network's passes).
		// The order inside a cell is whatever the atomics give; every consumer breaks distance ties by original index.
		__shared__ uint32_t wave_tot[MULLS_LDS_BLOCK / 64];
		uint32_t *cnt = K; // [ncell + 1]
		for (uint32_t c = threadIdx.x; c <= g.ncell; c += MULLS_LDS_BLOCK)
			cnt[c] = 0u;
		__syncthreads();
		for (uint32_t i = threadIdx.x; i < n; i += MULLS_LDS_BLOCK)
		{
			const float4 p = tpos[d.tgt_off + i];
			atomicAdd(&cnt[grid_cell_id(g, p.x, p.y, p.z)], 1u);
		}
		__syncthreads();
		// exclusive scan over the cells: consecutive cells per lane, wave scan, wave totals
		const uint32_t per = (g.ncell + 1u + MULLS_LDS_BLOCK - 1u) / MULLS_LDS_BLOCK;
		const uint32_t c0 = threadIdx.x * per, c1 = min(g.ncell + 1u, c0 + per);
		uint32_t sum = 0;
		for (uint32_t c = c0; c < c1; c++)
			sum += cnt[c];
		const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
		uint32_t incl = sum;
		for (int off = 1; off < 64; off <<= 1)
		{
			const uint32_t o = __shfl_up(incl, off);
			if (lane >= off)
				incl += o;
		}
		if (lane == 63)
			wave_tot[wave] = incl;
		__syncthreads();
		uint32_t before = incl - sum;
		for (int w = 0; w < wave; w++)
			before += wave_tot[w];
		for (uint32_t c = c0; c < c1; c++)
		{
			const uint32_t v = cnt[c];
			cnt[c] = before; // becomes the insertion cursor of the cell
			reinterpret_cast<uint16_t *>(cell_start)[g.cell_off + c] = (uint16_t)before; // LDS tier: 16-bit table (n <= MULLS_LDS_MAXPTS), half the bytes to stage
			before += v;
		}
		__syncthreads();
		for (uint32_t i = threadIdx.x; i < n; i += MULLS_LDS_BLOCK)
		{
			const float4 p = tpos[d.tgt_off + i];
			const uint32_t pos = atomicAdd(&cnt[grid_cell_id(g, p.x, p.y, p.z)], 1u);
			tsorted[d.tgt_off + pos] = make_float4(p.x, p.y, p.z, __int_as_float((int)i));
		}
		return;
	}
	uint32_t npow = 64;
	while (npow < n)
		npow <<= 1;
	for (uint32_t i = threadIdx.x; i < npow; i += MULLS_LDS_BLOCK)
	{
		uint32_t key = 0xffffffffu;
		if (i < n)
		{
			const float4 p = tpos[d.tgt_off + i];
			key = (grid_cell_id(g, p.x, p.y, p.z) << 14) | i;
		}
		K[i] = key;
	}
	__syncthreads();
	for (uint32_t kk = 2; kk <= npow; kk <<= 1)
		for (uint32_t j = kk >> 1; j > 0; j >>= 1)
		{
			for (uint32_t i = threadIdx.x; i < (npow >> 1); i += MULLS_LDS_BLOCK)
			{
				const uint32_t a = ((i & ~(j - 1u)) << 1) | (i & (j - 1u)), b = a | j;
				const uint32_t x = K[a], y = K[b];
				if ((x > y) == ((a & kk) == 0u))
				{
					K[a] = y;
					K[b] = x;
				}
			}
			__syncthreads();
		}
	for (uint32_t pos = threadIdx.x; pos < n; pos += MULLS_LDS_BLOCK)
	{
		const uint32_t idx = K[pos] & 16383u;
		const float4 p = tpos[d.tgt_off + idx];
		tsorted[d.tgt_off + pos] = make_float4(p.x, p.y, p.z, __int_as_float((int)idx));
	}
	// cell_start[c] = first sorted position whose cell id is >= c (c = ncell gives n)
	for (uint32_t c = threadIdx.x; c <= g.ncell; c += MULLS_LDS_BLOCK)
	{
		uint32_t lo = 0, hi = n;
		while (lo < hi)
		{
			const uint32_t mid = (lo + hi) >> 1;
			if ((K[mid] >> 14) < c)
				lo = mid + 1u;
			else
				hi = mid;
		}
		reinterpret_cast<uint16_t *>(cell_start)[g.cell_off + c] = (uint16_t)lo;
	}
}

// Correspondence search, global-memory grid tier (target class clouds too large for LDS).  Phase 1: the workgroup applies
// this iteration's rigid step to its slice of a 512-point job (coalesced 16-B traffic, double math once per point) and
// parks the transformed positions in LDS.  Phase 2: a 16-lane sub-group owns one query at a time: own cell, then the cube
// of min(cell edge, distance found), then — while nothing lies inside the probed radius — one last cube of the distance
// found, or cubes of twice the radius up to the rejection radius; rows are swept 16 at a time with coalesced candidate
// loads, 4 xor-shuffles reduce (distance, index).  `split` workgroups share one job (batches with few jobs would leave
// most CUs idle otherwise).  Outputs are identical to k_nn.
__global__ __launch_bounds__(MULLS_BLOCK) void k_nn_grid(const Job *__restrict__ jobs, CloudDesc *__restrict__ descs,
														  const PairState *__restrict__ states, RunParams rp, float4 *__restrict__ spos,
														  float4 *__restrict__ snrm, const GridDesc *__restrict__ grids,
														  const unsigned long long *__restrict__ bm, const uint32_t *__restrict__ pf,
														  const uint32_t *__restrict__ cs, const float4 *__restrict__ tsorted,
														  const uint8_t *__restrict__ flag, int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2,
														  unsigned long long *__restrict__ winner, uint32_t split)
{
	__shared__ float4 qpos[MULLS_SRC_PER_BLOCK]; // transformed query positions; w = 1 for live points, 0 for dead / out of range
	const uint32_t wg = xcd_job(blockIdx.x, gridDim.x);
	const Job job = jobs[wg / split];
	const uint32_t per = MULLS_SRC_PER_BLOCK / split, q0 = (wg % split) * per; // this workgroup's queries of the job: [q0, q0 + per)
	const PairState &ps = states[job.pair];
	if (!ps.active || (rp.normal_shooting && (job.cls == 0 || job.cls == 2 || job.cls == 4)))
		return;
	const uint32_t ci = job.pair * MULLS_NC + job.cls;
	CloudDesc &d = descs[ci];
	const uint32_t src_n = d.src_n, alive_cur = d.alive_cur;
	const bool called = class_called(rp, d, job.cls);
	{
		const double *T = ps.T;
		for (uint32_t k = threadIdx.x; k < per; k += MULLS_BLOCK)
		{
			const uint32_t s = job.start + q0 + k;
			float4 out = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			if (s < src_n && (flag[d.src_off + s] & MULLS_F_ALIVE))
			{
				// pcl::transformPointCloudWithNormals<PointT,double> (cregistration.hpp:1690-1695)
				const float4 p = spos[d.src_off + s], n = snrm[d.src_off + s];
				const double x = p.x, y = p.y, z = p.z, nx = n.x, ny = n.y, nz = n.z;
				out.x = (float)(T[0] * x + T[1] * y + T[2] * z + T[3]);
				out.y = (float)(T[4] * x + T[5] * y + T[6] * z + T[7]);
				out.z = (float)(T[8] * x + T[9] * y + T[10] * z + T[11]);
				out.w = 1.0f;
				const float onx = (float)(T[0] * nx + T[1] * ny + T[2] * nz);
				const float ony = (float)(T[4] * nx + T[5] * ny + T[6] * nz);
				const float onz = (float)(T[8] * nx + T[9] * ny + T[10] * nz);
				spos[d.src_off + s] = make_float4(out.x, out.y, out.z, p.w);
				snrm[d.src_off + s] = make_float4(onx, ony, onz, n.w);
			}
			qpos[k] = out;
		}
	}
	if (!called)
		return; // correspondences of the previous iteration stay in force
	__syncthreads();

	const GridDesc g = grids[ci];
	const BmGrid B = {bm + g.cell_off, pf + g.cell_off, cs + d.tgt_off + ci};
	const float4 *__restrict__ ts = tsorted + d.tgt_off;
	const float r = 2.5f * ps.thr[job.cls]; // filter_dis_times * dis_thre (float), cregistration.hpp:1745
	const double maxd = (double)r;
	const double max_dist_sqr = maxd * maxd;
	const bool gate = alive_cur >= 500u;
	const unsigned long long key_hi = (unsigned long long)(0xffffffffu - (rp.tick_base + (uint32_t)ps.iter)) << 32;
	const float m = fminf(r, 0.999f * g.h - 2e-4f); // first-probe radius: its margin-inflated cube spans at most 3 cells per axis
	const uint32_t sub = threadIdx.x & (MULLS_GRID_GROUP - 1u), grp = threadIdx.x / MULLS_GRID_GROUP;
	uint32_t matched_cnt = 0;
	for (uint32_t k = grp; k < per; k += MULLS_BLOCK / MULLS_GRID_GROUP)
	{
		const uint32_t s = job.start + q0 + k;
		if (s >= src_n)
			break;
		const float4 q = qpos[k];
		if (q.w == 0.0f)
			continue;
		float best = __builtin_inff();
		int bi = -1;
		// probe 0: the query's own cell
		const int ocx = grid_cell(q.x, g.ox, g.inv_h, g.nx), ocy = grid_cell(q.y, g.oy, g.inv_h, g.ny), ocz = grid_cell(q.z, g.oz, g.inv_h, g.nz);
		{
			uint32_t lo, hi;
			bm_row_range(B, ((uint32_t)ocz * g.ny + (uint32_t)ocy) * g.wpr, (uint32_t)ocx, (uint32_t)ocx, lo, hi);
			for (uint32_t t = lo + sub; t < hi; t += 4 * MULLS_GRID_GROUP)
			{
				float4 c[4];
				bool v[4];
#pragma unroll
				for (int w = 0; w < 4; w++)
				{
					v[w] = t + w * MULLS_GRID_GROUP < hi;
					if (v[w])
						c[w] = ts[t + w * MULLS_GRID_GROUP];
				}
#pragma unroll
				for (int w = 0; w < 4; w++)
					if (v[w])
					{
						const float dx = q.x - c[w].x, dy = q.y - c[w].y, dz = q.z - c[w].z;
						const float dist = (dx * dx + dy * dy) + dz * dz;
						const int idx = __float_as_int(c[w].w);
						if (dist < best || (dist == best && idx < bi))
						{
							best = dist;
							bi = idx;
						}
					}
			}
		}
		group_min(best, bi);
		// probe 1: the cells within min(first-probe radius, current best distance); skipped when that box is the own cell
		{
			const float R1 = bi >= 0 ? fminf(m, sqrtf(best)) : m;
			const float Rm = R1 * 1.0001f + 1e-4f;
			const bool own_only = grid_cell(q.x - Rm, g.ox, g.inv_h, g.nx) == ocx && grid_cell(q.x + Rm, g.ox, g.inv_h, g.nx) == ocx &&
								  grid_cell(q.y - Rm, g.oy, g.inv_h, g.ny) == ocy && grid_cell(q.y + Rm, g.oy, g.inv_h, g.ny) == ocy &&
								  grid_cell(q.z - Rm, g.oz, g.inv_h, g.nz) == ocz && grid_cell(q.z + Rm, g.oz, g.inv_h, g.nz) == ocz;
			if (!own_only)
			{
				grid_scan_box(g, B, ts, q.x, q.y, q.z, R1, sub, best, bi);
				group_min(best, bi);
			}
		}
		// nothing inside the probed radius yet: one last probe at the distance found, else double the radius (up to r)
		float Rc = m;
		while (!(bi >= 0 && best <= Rc * Rc) && Rc < r)
		{
			const bool last = bi >= 0;
			Rc = last ? fminf(r, sqrtf(best)) : fminf(r, 2.0f * Rc);
			grid_scan_box(g, B, ts, q.x, q.y, q.z, Rc, sub, best, bi);
			group_min(best, bi);
			if (last)
				break;
		}
		if (sub == 0)
		{
			const bool matched = bi >= 0 && !((double)best > max_dist_sqr);
			nn_idx[d.src_off + s] = matched ? bi : -1;
			nn_d2[d.src_off + s] = best;
			if (matched)
			{
				matched_cnt++;
				if (gate)
					atomicMin(&winner[d.tgt_off + bi], key_hi | (unsigned long long)s);
			}
		}
	}
	for (int off = 32; off > 0; off >>= 1)
		matched_cnt += __shfl_down(matched_cnt, off);
	if ((threadIdx.x & 63) == 0 && matched_cnt)
		atomicAdd(&d.n_matched, matched_cnt);
}

// ---------------------------------------------------------------------------------------------------------------
// Rejection chain of determine_corres after the search (cregistration.hpp:1755-1830), one source point.
// `dedup_done`: the duplicate rule has been applied already (losers carry nn_idx = -1, k_nn_lds with rp.lds_dedup).
struct FilterCtx
{
	bool gate, any_match, normal_check, dedup_done;
	float max_sqr;	 // CorrespondenceRejectorDistance::setMaximumDistance (float)
	double cos_thre; // cos(angle_thre_degree / 180.0 * M_PI), evaluated on the host (:1818)
	unsigned long long key_hi;
};
__device__ __forceinline__ void filter_point(const FilterCtx &F, const CloudDesc &d, uint32_t s, const float4 *__restrict__ snrm,
											  const float4 *__restrict__ tnrm, uint8_t *__restrict__ flag, const int32_t *__restrict__ nn_idx,
											  const float *__restrict__ nn_d2, int32_t *__restrict__ match, float *__restrict__ wd,
											  const unsigned long long *__restrict__ winner, const float4 *__restrict__ tpos,
											  float4 *__restrict__ mq, uint32_t &n_alive, uint32_t &n_valid)
{
	const uint32_t g = d.src_off + s;
	const uint32_t f = flag[g];
	if (!(f & MULLS_F_ALIVE))
		return;
	bool alive = true, valid, fresh = false;
	int m;
	float4 n2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	if (F.any_match)
	{
		m = nn_idx[g];
		valid = m >= 0;
		if (F.gate && (m < 0 || (!F.dedup_done && winner[d.tgt_off + m] != (F.key_hi | (unsigned long long)s))))
		{
			alive = false; // unmatched or duplicate-losing source points vanish for good (:1762-1789)
			valid = false;
		}
		if (valid)
		{
			const float dist = nn_d2[g];
			valid = !(dist > F.max_sqr);
			if (valid)
			{
				wd[g] = dist; // pcl::Correspondence::distance (shares storage with ::weight)
				// the matched target travels with the source point from here on: k_accum streams (position, direction) records
				// instead of gathering two cache lines per correspondence (its launches were bound by exactly that traffic).
				// From the second iteration on most points keep their target: the record is already there (it is only ever
				// written together with match[]), so neither gather nor store is needed — one coalesced 16-B read instead.
				if (match[g] == m)
					n2 = mq[2u * g + 1u];
				else
				{
					match[g] = m;
					n2 = tnrm[d.tgt_off + m];
					mq[2u * g] = tpos[d.tgt_off + m];
					mq[2u * g + 1u] = n2;
				}
				fresh = true;
			}
		}
	}
	else if (F.gate)
	{
		alive = false; // the whole cloud was swapped for an empty one; reference behaviour undefined, see oracle
		valid = false;
		m = -1;
	}
	else
	{
		// CorrespondenceRejectorDistance::getCorrespondences returned early on the empty input: the previous
		// Corr_f is still in place (SURVEY B-4) and goes through the direction check again.
		valid = (f & MULLS_F_VALID) != 0;
		m = match[g];
	}
	if (valid && F.normal_check)
	{
		const float4 n1 = snrm[g];
		if (!fresh)
			n2 = mq[2u * g + 1u]; // the standing correspondence's target direction
		const double dot = (double)n1.x * (double)n2.x + (double)n1.y * (double)n2.y + (double)n1.z * (double)n2.z;
		const float c = (float)fabs(dot);
		if ((double)c < F.cos_thre)
			valid = false;
	}
	flag[g] = (uint8_t)((alive ? MULLS_F_ALIVE : 0u) | (valid ? MULLS_F_VALID : 0u));
	n_alive += alive ? 1u : 0u;
	n_valid += valid ? 1u : 0u;
}

__global__ __launch_bounds__(MULLS_BLOCK) void k_filter(const Job *__restrict__ jobs, CloudDesc *__restrict__ descs,
														 const PairState *__restrict__ states, RunParams rp,
														 const float4 *__restrict__ snrm, const float4 *__restrict__ tnrm, uint8_t *__restrict__ flag,
														 const int32_t *__restrict__ nn_idx, const float *__restrict__ nn_d2,
														 int32_t *__restrict__ match, float *__restrict__ wd,
														 const unsigned long long *__restrict__ winner, const float4 *__restrict__ tpos,
														 float4 *__restrict__ mq)
{
	__shared__ uint32_t red4[4];
	const Job job = jobs[xcd_job(blockIdx.x, gridDim.x)];
	const PairState &ps = states[job.pair];
	if (!ps.active)
		return;
	CloudDesc &d = descs[job.pair * MULLS_NC + job.cls];
	if (!class_called(rp, d, job.cls))
		return;
	const float thr = ps.thr[job.cls];
	// vertex correspondences skip the direction check (cregistration.hpp:1292)
	const FilterCtx F = {d.alive_cur >= 500u, d.n_matched > 0u, job.cls != 5, rp.lds_dedup != 0u, thr * thr, rp.cos_bearing,
						 (unsigned long long)(0xffffffffu - (rp.tick_base + (uint32_t)ps.iter)) << 32};
	uint32_t n_alive = 0, n_valid = 0;
#pragma unroll
	for (int u = 0; u < MULLS_SRC_PER_THREAD; u++)
	{
		const uint32_t s = job.start + threadIdx.x + u * MULLS_BLOCK;
		if (s < d.src_n)
			filter_point(F, d, s, snrm, tnrm, flag, nn_idx, nn_d2, match, wd, winner, tpos, mq, n_alive, n_valid);
	}
	const uint32_t ta = block_sum_u32(n_alive, red4);
	const uint32_t tv = block_sum_u32(n_valid, red4);
	if (threadIdx.x == 0)
	{
		if (ta)
			atomicAdd(&d.alive_next, ta);
		if (tv)
			atomicAdd(&d.valid_next, tv);
	}
}

// ---------------------------------------------------------------------------------------------------------------
// Correspondence search, LDS grid tier — the default whenever every searched target class cloud holds at most
// MULLS_LDS_MAXPTS points (the reference-default KITTI sizes).  Rationale (profiles/r01_c_pmc_grid.txt): the global-memory
// grid tier is neither HBM- nor VALU-bound, it waits (69 % s_waitcnt) on chains of 64-B sector gathers with ~2 us
// latency.  Here one 1024-lane workgroup per (pair, class) — class-level jobs; 512-query jobs when a batch has too few
// class clouds to fill the chip — brings the whole cell-sorted target class cloud on chip once with coalesced 16-B
// loads: 12-B position records, a uint16 original index per point, a uint16 cell table (as many cells as the rest of the
// 160 KiB allows), and then searches the source class cloud against it in equal chunks of at most MULLS_LDS_QCHUNK queries:
//   rigid step   one lane per query: this iteration's transform (loads issued one chunk ahead), and the distance to the
//                target the point found in the previous iteration — an exact upper bound that travels as a 4-B hint;
//   cost order   counting sort of the chunk's queries by the candidate trips they took last time (kept next to the hint):
//                the eight sub-groups of a wave run in lock step, so neighbours should cost the same;
//   search       8-lane sub-groups, one query each: the cube of the bound's radius, two rows of cells per step, their
//                candidate ranges laid end to end, two candidates per lane in flight, state = one 64-bit
//                (distance bits, index) key, DPP minima.  Unhinted queries probe their own cell first; the 2.5*thr ball
//                is swept only if nothing lies within one cell edge;
//   tail         with class-level jobs: duplicate rule in an LDS table, then the rejection chain (filter_point) on the
//                results while they are still in cache, and the matched target's record for k_accum.
// Same exactness argument, same outputs as k_nn / k_nn_grid.  The time of this kernel follows its VALU instruction
// count (profiles/r01_h_pmc_sq.txt); profiles/r01_m_search_steps.txt lists what each of the choices above bought.
namespace
{
struct LdsGrid
{
	const float *P;	// staged target positions, 12-B records (x, y, z): one address, three immediate offsets, conflict-free stride
	const uint16_t *IDX, *CS;
};

// Search state of a query: (distance bits << 32) | target index.  Distances are sums of squares (>= +0, or NaN), so the
// unsigned order of the key is the lexicographic (distance, index) order the tie rule asks for, NaN keys sort after
// NNKEY_NONE and are never taken, and one 64-bit compare + two selects update the running minimum.
typedef unsigned long long nnkey;
#define NNKEY_NONE 0x7f800000ffffffffull
__device__ __forceinline__ nnkey nn_key(float dist, uint32_t idx) { return ((nnkey)__float_as_uint(dist) << 32) | idx; }
__device__ __forceinline__ float key_dist(nnkey k) { return __uint_as_float((uint32_t)(k >> 32)); }
__device__ __forceinline__ bool key_found(nnkey k) { return (uint32_t)k != 0xffffffffu; }

// minimum over the lanes of a sub-group, VALU only (DPP; no LDS-crossbar shuffles): quad xor-1, quad xor-2, half-row
// mirror (8 lanes), row mirror (16 lanes).  Result in every lane.
template <int CTRL>
__device__ __forceinline__ void dpp_min_step(nnkey &bk)
{
	const uint32_t oh = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(bk >> 32), CTRL, 0xf, 0xf, false);
	const uint32_t ol = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)bk, CTRL, 0xf, 0xf, false);
	const nnkey o = ((nnkey)oh << 32) | ol;
	bk = o < bk ? o : bk;
}
__device__ __forceinline__ void row16_min(nnkey &bk)
{
	dpp_min_step<0xB1>(bk);	 // quad_perm [1,0,3,2]
	dpp_min_step<0x4E>(bk);	 // quad_perm [2,3,0,1]
#if MULLS_LDS_GROUP >= 8
	dpp_min_step<0x141>(bk); // row_half_mirror: lanes i <-> 7 - i of each 8-lane half
#endif
#if MULLS_LDS_GROUP == 16
	dpp_min_step<0x140>(bk); // row_mirror: lanes i <-> 15 - i
#endif
}

// Evaluate every staged target in the cells intersecting the cube [p - R, p + R] (same exactness argument as
// grid_scan_box).  The rows (x-runs of cells, contiguous in the sorted cloud) are taken four at a time: every lane of the
// sub-group reads their bounds (same addresses: LDS broadcast), the four candidate ranges are laid end to end and the
// sub-group strides over the concatenation, two candidates per lane in flight — the trip count is that of the total, not
// the sum of the per-row round-ups, and the only per-row work is two table reads and a running sum.
#define MULLS_LDS_CHUNK 2
__device__ __forceinline__ void lds_scan_box(const GridDesc &g, const LdsGrid &L, float px, float py, float pz, float R, uint32_t sub,
											  nnkey &bk, uint32_t &trips, bool own_done = false)
{
	const float Rm = R * 1.0001f + 1e-4f;
	const uint32_t x0 = (uint32_t)grid_cell(px - Rm, g.ox, g.inv_h, g.nx), x1 = (uint32_t)grid_cell(px + Rm, g.ox, g.inv_h, g.nx) + 1u;
	const int y0 = grid_cell(py - Rm, g.oy, g.inv_h, g.ny), y1 = grid_cell(py + Rm, g.oy, g.inv_h, g.ny);
	const int z0 = grid_cell(pz - Rm, g.oz, g.inv_h, g.nz), z1 = grid_cell(pz + Rm, g.oz, g.inv_h, g.nz);
	if (own_done && x1 - x0 == 1u && y0 == y1 && z0 == z1)
		return; // the cube stays inside the query's own cell, which has been swept already
	int cy = y0, cz = z0;
	while (cz <= z1)
	{
		uint32_t lo[MULLS_LDS_CHUNK], pre[MULLS_LDS_CHUNK], acc = 0;
#pragma unroll
		for (int jj = 0; jj < MULLS_LDS_CHUNK; jj++)
		{
			const bool valid = cz <= z1;
			const uint32_t row = ((uint32_t)(valid ? cz : z0) * g.ny + (uint32_t)cy) * g.nx;
			const uint32_t a = L.CS[row + x0], e = L.CS[row + x1];
			lo[jj] = a - acc; // candidate f of the concatenation lives at lo[jj] + f while f < pre[jj]
			acc += valid ? e - a : 0u;
			pre[jj] = acc;
			if (++cy > y1)
			{
				cy = y0;
				cz++;
			}
		}
		trips += (acc + 2u * MULLS_LDS_GROUP - 1u) / (2u * MULLS_LDS_GROUP);
		for (uint32_t f = sub; f < acc; f += 2 * MULLS_LDS_GROUP)
		{
			const uint32_t f2 = f + MULLS_LDS_GROUP;
			const bool ok2 = f2 < acc;
			const uint32_t ff = ok2 ? f2 : f;
#if MULLS_LDS_CHUNK == 1
			const uint32_t ta = f + lo[0], tb = ff + lo[0];
#else
			const uint32_t ta = f + (f < pre[0] ? lo[0] : lo[1]);
			const uint32_t tb = ff + (ff < pre[0] ? lo[0] : lo[1]);
#endif
			const float *pa = L.P + 3u * ta, *pb = L.P + 3u * tb;
			const float ax = pa[0], ay = pa[1], az = pa[2], bx = pb[0], by = pb[1], bz = pb[2];
			const uint32_t ia = L.IDX[ta], ib = L.IDX[tb];
			float dx = px - ax, dy = py - ay, dz = pz - az;
			const nnkey ka = nn_key((dx * dx + dy * dy) + dz * dz, ia); // L2_Simple<float>, no FMA
			dx = px - bx, dy = py - by, dz = pz - bz;
			const nnkey kb = nn_key((dx * dx + dy * dy) + dz * dz, ib); // !ok2: the same candidate again, no effect
			bk = ka < bk ? ka : bk;
			bk = kb < bk ? kb : bk;
		}
	}
}
} // namespace

__global__ __launch_bounds__(MULLS_LDS_BLOCK) void k_nn_lds(const Job *__restrict__ cjobs, CloudDesc *__restrict__ descs,
															 const PairState *__restrict__ states, RunParams rp, float4 *__restrict__ spos,
															 float4 *__restrict__ snrm, const GridDesc *__restrict__ grids,
															 const uint32_t *__restrict__ cell_start, const float4 *__restrict__ tsorted,
															 uint8_t *flag, int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2,
															 unsigned long long *__restrict__ winner, const float4 *__restrict__ tnrm, int32_t *__restrict__ match,
															 float *__restrict__ wd, const float4 *__restrict__ tpos, int32_t *__restrict__ nn_hint, float4 *__restrict__ mq, uint32_t cap)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	float4 *qpos = reinterpret_cast<float4 *>(lds_raw);					  // [512] transformed queries, w = 1 live / 0 dead
	uint32_t *HIST = reinterpret_cast<uint32_t *>(qpos + MULLS_LDS_QCHUNK); // [32] cost histogram, [32] bucket bases, [64] live queries of the chunk
	uint16_t *ORDER = reinterpret_cast<uint16_t *>(HIST + 80);				  // [MULLS_LDS_QCHUNK] query slots, most expensive first
	float *P = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(HIST) + MULLS_LDS_AUX); // [3 * cap] x, y, z records
	uint16_t *IDX = reinterpret_cast<uint16_t *>(P + 3u * cap);			  // [cap]
	uint16_t *CS = IDX + cap;											  // [rp.grid_maxcells + 1]
	uint32_t *W = reinterpret_cast<uint32_t *>(CS + ((rp.grid_maxcells + 8u) & ~1u)); // [cap] lowest source index matched to each target (lds_dedup)

	// one workgroup per (pair, class): the target class cloud is staged ONCE and every 512-query chunk of the source
	// class cloud is searched against it (the first version staged it once per chunk: 2.3x the algorithmic HBM bytes,
	// profiles/r01_f_pmc_traffic.txt)
	const Job job = cjobs[blockIdx.x]; // host order: most expensive class clouds first (round-robin over the XCDs)
	const PairState &ps = states[job.pair];
	if (!ps.active || (rp.normal_shooting && (job.cls == 0 || job.cls == 2 || job.cls == 4)))
		return;
	CloudDesc &d = descs[job.pair * MULLS_NC + job.cls];
	const uint32_t src_n = d.src_n, tgt_n = d.tgt_n, alive_cur = d.alive_cur;
	const bool called = class_called(rp, d, job.cls);
	const GridDesc g = grids[job.pair * MULLS_NC + job.cls];

	// The rigid-step phase of a chunk needs four global loads per lane and, with a hint, a dependent gather: they are issued one
	// chunk ahead (the first chunk's before the target cloud is staged) so that their latency hides behind the staging / the search.
	const bool use_hint = called && ps.iter > 0 && rp.debug_stop != 6u; // hints of this run exist from its second iteration on
	const uint32_t q_end = min(src_n, job.start + (job.count ? job.count : (uint32_t)MULLS_SRC_PER_BLOCK));
	// chunks of equal size (1200 queries: 2 x 600, not 1024 + 176: the last chunk would leave most sub-groups idle)
	const uint32_t q_cnt = q_end > job.start ? q_end - job.start : 0u, n_chunks = (q_cnt + MULLS_LDS_QCHUNK - 1u) / MULLS_LDS_QCHUNK;
	const uint32_t q_step = n_chunks ? (q_cnt + n_chunks - 1u) / n_chunks : 1u;
	float4 pf_p = make_float4(0.0f, 0.0f, 0.0f, 0.0f), pf_n = pf_p, pf_t = pf_p;
	uint32_t pf_hv = 0xffffu, pf_f = 0u, pf_g = 0u;
	int32_t pf_m = -1;
	auto prefetch = [&](uint32_t chunk) {
		const uint32_t s = chunk + threadIdx.x;
		pf_f = 0u;
		pf_hv = 0xffffu;
		pf_m = -1;
		pf_g = d.src_off + s;
		if (chunk < q_end && s < min(q_end, chunk + q_step))
		{
			pf_f = flag[pf_g];
			pf_p = spos[pf_g];
			pf_n = snrm[pf_g];
			if (use_hint)
			{
				pf_hv = (uint32_t)nn_hint[pf_g];
				pf_m = match[pf_g];
			}
		}
	};
	auto prefetch_hint = [&]() {
		// the hinted target's position: for a point whose hint is its standing correspondence it sits in the point's own
		// record (coalesced), otherwise it is gathered
		const uint32_t h = pf_hv & 0xffffu;
		if (h < tgt_n)
			pf_t = (int32_t)h == pf_m ? mq[2u * pf_g] : tpos[d.tgt_off + h];
	};
	prefetch(job.start);

	if (called)
	{
		// stage the cell-sorted target cloud and its cell table (coalesced reads).  Loads are issued in batches of 8 / 4
		// per lane before the first LDS write: one memory latency per batch instead of one per element.
		const float4 *__restrict__ ts = tsorted + d.tgt_off;
		for (uint32_t k0 = threadIdx.x; k0 < tgt_n; k0 += 8 * MULLS_LDS_BLOCK)
		{
			float4 t[8];
#pragma unroll
			for (int u = 0; u < 8; u++)
			{
				const uint32_t k = k0 + u * MULLS_LDS_BLOCK;
				if (k < tgt_n)
					t[u] = ts[k];
			}
#pragma unroll
			for (int u = 0; u < 8; u++)
			{
				const uint32_t k = k0 + u * MULLS_LDS_BLOCK;
				if (k < tgt_n)
				{
					P[3u * k] = t[u].x;
					P[3u * k + 1u] = t[u].y;
					P[3u * k + 2u] = t[u].z;
					IDX[k] = (uint16_t)__float_as_int(t[u].w);
				}
			}
		}
		prefetch_hint(); // the first chunk's hinted targets, in flight while the cell table is staged
		// cell table: (ncell + 1) uint16 entries written by k_grid_build_sort, moved as uint4 words of 8 (the table slot of a
		// cloud is uint4-aligned and padded)
		const uint4 *__restrict__ cs4 = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(cell_start) + g.cell_off);
		const uint32_t nw = (g.ncell + 1u + 7u) >> 3;
		for (uint32_t w0 = threadIdx.x; w0 < nw; w0 += 4 * MULLS_LDS_BLOCK)
		{
			uint4 v[4];
#pragma unroll
			for (int u = 0; u < 4; u++)
			{
				const uint32_t w = w0 + u * MULLS_LDS_BLOCK;
				if (w < nw)
					v[u] = cs4[w];
			}
#pragma unroll
			for (int u = 0; u < 4; u++)
			{
				const uint32_t w = w0 + u * MULLS_LDS_BLOCK;
				if (w < nw)
					reinterpret_cast<uint4 *>(CS)[w] = v[u];
			}
		}
	}

	const bool dedup = rp.lds_dedup != 0u && called && alive_cur >= 500u;
	if (dedup)
		for (uint32_t t = threadIdx.x; t < tgt_n; t += MULLS_LDS_BLOCK)
			W[t] = 0xffffffffu;
	const LdsGrid L = {P, IDX, CS};
	const float r = 2.5f * ps.thr[job.cls]; // filter_dis_times * dis_thre (float), cregistration.hpp:1745
	const double maxd = (double)r;
	const double max_dist_sqr = maxd * maxd;
	const bool gate = alive_cur >= 500u;
	const unsigned long long key_hi = (unsigned long long)(0xffffffffu - (rp.tick_base + (uint32_t)ps.iter)) << 32;
	const float m = fminf(r, 0.999f * g.h - 2e-4f); // first-probe radius: its margin-inflated cube spans at most 3 cells per axis
	const uint32_t sub = threadIdx.x & (MULLS_LDS_GROUP - 1u), grp = threadIdx.x / MULLS_LDS_GROUP;
	uint32_t matched_cnt = 0;

	if (threadIdx.x < 32u)
		HIST[threadIdx.x] = 0u;
	for (uint32_t chunk = job.start; chunk < q_end; chunk += q_step)
	{
		const uint32_t c_end = min(q_end, chunk + q_step);
		__syncthreads(); // the previous chunk's queries have been consumed (and, first trip, the staging stores are visible below)
		uint32_t bucket = 0, rank = 0xffffffffu; // cost class of this lane's query (0 = most expensive) and its rank inside the class
		// phase 1: one source point per lane (lanes 0..511) — fused rigid step (cregistration.hpp:1690-1695), coalesced 16-B traffic
		if (threadIdx.x < MULLS_LDS_QCHUNK)
		{
			const uint32_t s = chunk + threadIdx.x;
			float4 out = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
			if (s < c_end && (pf_f & MULLS_F_ALIVE))
			{
				const float4 p = pf_p, n = pf_n;
				const double *T = ps.T;
				const double x = p.x, y = p.y, z = p.z, nx = n.x, ny = n.y, nz = n.z;
				out.x = (float)(T[0] * x + T[1] * y + T[2] * z + T[3]);
				out.y = (float)(T[4] * x + T[5] * y + T[6] * z + T[7]);
				out.z = (float)(T[8] * x + T[9] * y + T[10] * z + T[11]);
				const float onx = (float)(T[0] * nx + T[1] * ny + T[2] * nz);
				const float ony = (float)(T[4] * nx + T[5] * ny + T[6] * nz);
				const float onz = (float)(T[8] * nx + T[9] * ny + T[10] * nz);
				spos[d.src_off + s] = make_float4(out.x, out.y, out.z, p.w);
				snrm[d.src_off + s] = make_float4(onx, ony, onz, n.w);
				// temporal coherence: the target this point found in the previous iteration is very likely still its nearest one.
				// Its distance is an exact upper bound (any target point gives one), so the search below can skip the own-cell
				// probe and sweep the cube of that radius straight away.  w: bound (squared), +inf = none, -1 = dead point.
				out.w = __builtin_inff();
				bucket = 31u - ((pf_hv >> 16) & 31u);
				if ((pf_hv & 0xffffu) < tgt_n)
				{
					const float dx = out.x - pf_t.x, dy = out.y - pf_t.y, dz = out.z - pf_t.z;
					const float d0 = (dx * dx + dy * dy) + dz * dz;
					if (d0 >= 0.0f)
						out.w = d0;
				}
				if (called)
					rank = atomicAdd(&HIST[bucket], 1u);
			}
			qpos[threadIdx.x] = out;
		}
		prefetch(chunk + q_step); // the next chunk's loads, consumed after this chunk's search
		if (!called || rp.debug_stop == 1u)
			continue; // correspondences of the previous iteration stay in force; the points still move
		__syncthreads();
		if (rp.debug_stop == 2u)
			continue;
		// queries of the chunk in order of the work they took in the previous iteration (candidate trips, kept next to the hint):
		// the eight sub-groups of a wave run in lock step, so a wave is as slow as its most expensive query — neighbours in
		// this order cost about the same.  Counting sort over 32 classes; dead points drop out.
		if (threadIdx.x < 32u)
		{
			const uint32_t v = HIST[threadIdx.x];
			uint32_t incl = v;
			for (int off = 1; off < 32; off <<= 1)
			{
				const uint32_t o = __shfl_up(incl, off);
				if ((int)threadIdx.x >= off)
					incl += o;
			}
			HIST[32u + threadIdx.x] = incl - v;
			HIST[threadIdx.x] = 0u; // ready for the next chunk
			if (threadIdx.x == 31u)
				HIST[64] = incl;
		}
		__syncthreads();
		if (rank != 0xffffffffu)
			ORDER[HIST[32u + bucket] + rank] = (uint16_t)threadIdx.x;
		__syncthreads();
		const uint32_t n_live = HIST[64];
		prefetch_hint(); // gather of the next chunk's hinted targets (its hint words have landed during the sort above)

		// phase 2: sub-groups of MULLS_LDS_GROUP lanes, one query at a time each
		for (uint32_t i = grp; i < n_live; i += MULLS_LDS_BLOCK / MULLS_LDS_GROUP)
		{
			const uint32_t k = ORDER[i], s = chunk + k;
			const float4 q = qpos[k];
			nnkey bk = NNKEY_NONE;
			uint32_t trips = 0;
			if (rp.debug_stop == 5u)
				continue;
			if (q.w < __builtin_inff())
			{
				// bounded by last iteration's correspondence: one sweep of the cube of that radius (it contains that target)
				lds_scan_box(g, L, q.x, q.y, q.z, fminf(m, sqrtf(q.w)), sub, bk, trips);
				row16_min(bk);
			}
			else
			{
				// probe 0: the query's own cell.  In dense regions (tens of targets per cell) this already yields a tight bound.
				const int cx = grid_cell(q.x, g.ox, g.inv_h, g.nx), cy = grid_cell(q.y, g.oy, g.inv_h, g.ny), cz = grid_cell(q.z, g.oz, g.inv_h, g.nz);
				const uint32_t cell = ((uint32_t)cz * g.ny + (uint32_t)cy) * g.nx + (uint32_t)cx;
				const uint32_t lo = L.CS[cell], hi = L.CS[cell + 1u];
				trips += (hi - lo + 2u * MULLS_LDS_GROUP - 1u) / (2u * MULLS_LDS_GROUP);
				for (uint32_t t = lo + sub; t < hi; t += 2 * MULLS_LDS_GROUP) // two candidates in flight per lane and trip
				{
					const uint32_t t2 = t + MULLS_LDS_GROUP;
					const uint32_t tt2 = t2 < hi ? t2 : t;
					const float *pa = L.P + 3u * t, *pb = L.P + 3u * tt2;
					const float ax = pa[0], ay = pa[1], az = pa[2], bx = pb[0], by = pb[1], bz = pb[2];
					const uint32_t ia = L.IDX[t], ib = L.IDX[tt2];
					float dx = q.x - ax, dy = q.y - ay, dz = q.z - az;
					const nnkey ka = nn_key((dx * dx + dy * dy) + dz * dz, ia);
					dx = q.x - bx, dy = q.y - by, dz = q.z - bz;
					const nnkey kb = nn_key((dx * dx + dy * dy) + dz * dz, ib);
					bk = ka < bk ? ka : bk;
					bk = kb < bk ? kb : bk;
				}
				row16_min(bk);
				// probe 1: every cell within min(first-probe radius, current best distance) of the query
				if (rp.debug_stop != 3u)
				{
					const float R1 = key_found(bk) ? fminf(m, sqrtf(key_dist(bk))) : m;
					lds_scan_box(g, L, q.x, q.y, q.z, R1, sub, bk, trips, true);
					row16_min(bk);
				}
			}
			if (rp.debug_stop < 3u && !(key_found(bk) && key_dist(bk) <= m * m))
			{
				// nothing within the first-probe radius: widen to the current best distance, or to the rejection radius
				const float R = key_found(bk) ? fminf(r, sqrtf(key_dist(bk))) : r;
				lds_scan_box(g, L, q.x, q.y, q.z, R, sub, bk, trips);
				row16_min(bk);
			}
			if (sub == 0)
			{
				const float best = key_dist(bk);
				const int bi = (int)(uint32_t)bk; // -1: nothing found
				const bool matched = bi >= 0 && !((double)best > max_dist_sqr);
				nn_idx[d.src_off + s] = matched ? bi : -1;
				nn_d2[d.src_off + s] = best;
				nn_hint[d.src_off + s] = (int32_t)(((uint32_t)bi & 0xffffu) | (min(trips, 31u) << 16)); // hint and cost class of the next iteration
				if (matched)
				{
					matched_cnt++;
					if (dedup)
						atomicMin(&W[bi], s); // this workgroup sees every query of the class cloud: the duplicate table stays on chip
					else if (gate)
						atomicMin(&winner[d.tgt_off + bi], key_hi | (unsigned long long)s);
				}
			}
		}
	}
	if (dedup)
	{
		// first source (lowest index) matched to a target keeps it (cregistration.hpp:1762-1789); the others become unmatched,
		// which is what k_filter does with them anyway (it skips its winner-table check when rp.lds_dedup is set)
		__threadfence_block();
		__syncthreads();
		for (uint32_t s = job.start + threadIdx.x; s < q_end; s += MULLS_LDS_BLOCK)
			if (flag[d.src_off + s] & MULLS_F_ALIVE)
			{
				const int m = nn_idx[d.src_off + s];
				if (m >= 0 && W[m] != s)
					nn_idx[d.src_off + s] = -1;
			}
	}
	for (int off = 32; off > 0; off >>= 1)
		matched_cnt += __shfl_down(matched_cnt, off);
	if (!rp.lds_dedup)
	{
		if ((threadIdx.x & 63) == 0 && matched_cnt)
			atomicAdd(&d.n_matched, matched_cnt);
		return;
	}
	// Class-level jobs: this workgroup holds every correspondence of the class cloud, so the rejection chain (k_filter) runs
	// right here while the results are still in cache — one launch and one pass over nn_idx / nn_d2 less per iteration.
	if (!called)
		return;
	uint32_t *red = reinterpret_cast<uint32_t *>(lds_raw); // the query block is free now: [0..15] matched, [16..31] alive, [32..47] valid
	__threadfence_block(); // this workgroup's nn_idx / nn_d2 / snrm stores, read back below by other lanes
	__syncthreads();
	if ((threadIdx.x & 63) == 0)
		red[threadIdx.x >> 6] = matched_cnt;
	__syncthreads();
	uint32_t total_matched = 0;
	for (int w = 0; w < MULLS_LDS_BLOCK / 64; w++)
		total_matched += red[w];
	const float thr = ps.thr[job.cls];
	const FilterCtx F = {gate, total_matched > 0u, job.cls != 5, true, thr * thr, rp.cos_bearing, key_hi};
	uint32_t n_alive = 0, n_valid = 0;
	for (uint32_t s = job.start + threadIdx.x; s < q_end; s += MULLS_LDS_BLOCK)
		filter_point(F, d, s, snrm, tnrm, flag, nn_idx, nn_d2, match, wd, winner, tpos, mq, n_alive, n_valid);
	for (int off = 32; off > 0; off >>= 1)
	{
		n_alive += __shfl_down(n_alive, off);
		n_valid += __shfl_down(n_valid, off);
	}
	if ((threadIdx.x & 63) == 0)
	{
		red[16 + (threadIdx.x >> 6)] = n_alive;
		red[32 + (threadIdx.x >> 6)] = n_valid;
	}
	__syncthreads();
	if (threadIdx.x == 0)
	{
		uint32_t ta = 0, tv = 0;
		for (int w = 0; w < MULLS_LDS_BLOCK / 64; w++)
		{
			ta += red[16 + w];
			tv += red[32 + w];
		}
		d.n_matched = total_matched; // k_finish reset it to 0 after the previous iteration
		d.alive_next = ta;
		d.valid_next = tv;
	}
}

// ---------------------------------------------------------------------------------------------------------------
// Normal-shooting correspondence search (normal_shooting_on; planar classes only: cregistration.hpp:1730-1739, PCL's
// CorrespondenceEstimationNormalShooting with k = 10).  Among the 10 nearest targets (ascending (d^2, index)) the one
// minimising |n_s x (p_t - p_s)|^2 (double) wins; it is rejected if that minimum exceeds max_distance (compared with
// r = 2.5*thr, not r^2 — PCL quirk); the stored distance is that candidate's squared Euclidean distance.  No shipped
// configuration enables the option, so this kernel is written for exactness, not speed: one query per lane, targets
// broadcast from an LDS tile, a sorted 10-entry list per lane.
__global__ __launch_bounds__(MULLS_BLOCK) void k_nn_shoot(const Job *__restrict__ jobs, CloudDesc *__restrict__ descs,
														   const PairState *__restrict__ states, RunParams rp, float4 *__restrict__ spos,
														   float4 *__restrict__ snrm, const float4 *__restrict__ tpos,
														   const uint8_t *__restrict__ flag, int32_t *__restrict__ nn_idx, float *__restrict__ nn_d2,
														   unsigned long long *__restrict__ winner)
{
	__shared__ float4 tile[1024];
	const Job job = jobs[blockIdx.x];
	if (!(job.cls == 0 || job.cls == 2 || job.cls == 4))
		return; // pillar / beam / vertex always use the plain nearest neighbour
	const PairState &ps = states[job.pair];
	if (!ps.active)
		return;
	CloudDesc &d = descs[job.pair * MULLS_NC + job.cls];
	const uint32_t src_n = d.src_n, tgt_n = d.tgt_n, alive_cur = d.alive_cur;
	const bool called = class_called(rp, d, job.cls);
	const float r = 2.5f * ps.thr[job.cls];
	const double max_distance = (double)r;
	const bool gate = alive_cur >= 500u;
	const unsigned long long key_hi = (unsigned long long)(0xffffffffu - (rp.tick_base + (uint32_t)ps.iter)) << 32;
	uint32_t matched_cnt = 0;
	for (int u = 0; u < MULLS_SRC_PER_THREAD; u++) // uniform trip count: the tile loop below contains barriers
	{
		const uint32_t s = job.start + threadIdx.x + u * MULLS_BLOCK;
		const bool alive = s < src_n && (flag[d.src_off + s] & MULLS_F_ALIVE);
		float px = 0, py = 0, pz = 0, nx = 0, ny = 0, nz = 0;
		if (alive)
		{
			const float4 p = spos[d.src_off + s], n = snrm[d.src_off + s];
			const double *T = ps.T;
			const double x = p.x, y = p.y, z = p.z, ax = n.x, ay = n.y, az = n.z;
			px = (float)(T[0] * x + T[1] * y + T[2] * z + T[3]);
			py = (float)(T[4] * x + T[5] * y + T[6] * z + T[7]);
			pz = (float)(T[8] * x + T[9] * y + T[10] * z + T[11]);
			nx = (float)(T[0] * ax + T[1] * ay + T[2] * az);
			ny = (float)(T[4] * ax + T[5] * ay + T[6] * az);
			nz = (float)(T[8] * ax + T[9] * ay + T[10] * az);
			spos[d.src_off + s] = make_float4(px, py, pz, p.w);
			snrm[d.src_off + s] = make_float4(nx, ny, nz, n.w);
		}
		if (!called)
			continue;
		float kd[10];
		int ki[10];
#pragma unroll
		for (int j = 0; j < 10; j++)
		{
			kd[j] = __builtin_inff();
			ki[j] = 0x7fffffff;
		}
		for (uint32_t base = 0; base < tgt_n; base += 1024)
		{
			const uint32_t nt = min(1024u, tgt_n - base);
			__syncthreads();
			for (uint32_t k = threadIdx.x; k < nt; k += MULLS_BLOCK)
				tile[k] = tpos[d.tgt_off + base + k];
			__syncthreads();
			for (uint32_t j = 0; j < nt; j++)
			{
				const float4 t = tile[j];
				const float dx = px - t.x, dy = py - t.y, dz = pz - t.z;
				float dist = (dx * dx + dy * dy) + dz * dz;
				int idx = (int)(base + j);
				if (dist < kd[9]) // indices arrive in ascending order: an equal distance never displaces an earlier index
				{
#pragma unroll
					for (int q = 0; q < 10; q++) // sorted insertion by one pass of compare-exchange
					{
						const bool lt = dist < kd[q];
						const float td = kd[q];
						const int ti = ki[q];
						kd[q] = lt ? dist : td;
						ki[q] = lt ? idx : ti;
						dist = lt ? td : dist;
						idx = lt ? ti : idx;
					}
				}
			}
		}
		if (!alive)
			continue;
		double min_dist = 1.7976931348623157e308;
		int min_index = 0;
		// entries are filled from the front; with a NaN query (a singular solve upstream propagates NaNs, SURVEY B-11) no distance
		// compares below infinity and nothing is found — the oracle's kd-tree returns an empty list there, too
		int found = 0;
#pragma unroll
		for (int j = 0; j < 10; j++)
			found += ki[j] != 0x7fffffff ? 1 : 0;
#pragma unroll
		for (int j = 0; j < 10; j++)
			if (j < found)
			{
				const float4 t = tpos[d.tgt_off + (uint32_t)ki[j]];
				const float ptx = px - t.x, pty = py - t.y, ptz = pz - t.z; // PCL forms the difference in float
				const double Vx = ptx, Vy = pty, Vz = ptz, Nx = nx, Ny = ny, Nz = nz;
				const double cx = Ny * Vz - Nz * Vy, cy = Nz * Vx - Nx * Vz, cz = Nx * Vy - Ny * Vx;
				const double dist = cx * cx + cy * cy + cz * cz;
				if (dist < min_dist)
				{
					min_dist = dist;
					min_index = j;
				}
			}
		float sel_d = 0.0f;
		int sel_i = -1;
#pragma unroll
		for (int j = 0; j < 10; j++)
			if (j == min_index)
			{
				sel_d = kd[j];
				sel_i = ki[j];
			}
		const bool matched = found > 0 && !(min_dist > max_distance);
		nn_idx[d.src_off + s] = matched ? sel_i : -1;
		nn_d2[d.src_off + s] = sel_d;
		if (matched)
		{
			matched_cnt++;
			if (gate)
				atomicMin(&winner[d.tgt_off + sel_i], key_hi | (unsigned long long)s);
		}
	}
	for (int off = 32; off > 0; off >>= 1)
		matched_cnt += __shfl_down(matched_cnt, off);
	if ((threadIdx.x & 63) == 0 && matched_cnt)
		atomicAdd(&d.n_matched, matched_cnt);
}

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
														 uint32_t pair_base)
{
	const uint32_t pair = pair_base + blockIdx.x;
	const int active = states[pair].active, want_residual = states[pair].want_residual;
	if (active || want_residual) // uniform per workgroup
		finish_pair(descs + pair * MULLS_NC, states[pair], rp, partial, out[pair], bbox + pair * 6);
}

// Results to the host: the records of pairs [pair_base, pair_base + npairs) go from HBM to pinned host memory as
// coalesced 16-B stores over PCIe (no copy-engine command, no stream synchronisation per iteration), packed to the
// counter block + the used classes.  Completion is published through a host-visible epoch word: every workgroup makes
// its stores system-visible, takes a ticket, and the last one to arrive writes the epoch the host is spinning on.
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

void launch_clone_src(hipStream_t st, uint32_t njobs, const Job *jobs, const CloudDesc *descs, const PairSetup *setup, const float4 *stage,
					  float4 *tmp_pos, float4 *tmp_nrm, uint32_t *bbox, const RunParams &rp)
{
	if (njobs)
		hipLaunchKernelGGL(k_clone_src, dim3(njobs), dim3(MULLS_BLOCK), 0, st, jobs, descs, setup, stage, tmp_pos, tmp_nrm, bbox, rp);
}
void launch_crop(hipStream_t st, uint32_t npairs, CloudDesc *descs, const PairSetup *setup, const uint32_t *bbox, const float4 *stage,
				 const float4 *tmp_pos, const float4 *tmp_nrm, float4 *spos, float4 *snrm, float4 *tpos, float4 *tnrm, uint8_t *flag,
				 int32_t *match, float *wd, const RunParams &rp, GridDesc *grids, uint32_t nbig_segs, const Job *big_segs, uint32_t nbig_clouds,
				 const Job *big_clouds, uint32_t *seg_cnt, uint32_t *big_box)
{
	if (!npairs)
		return;
	hipLaunchKernelGGL(k_crop, dim3(npairs * MULLS_NC * 2), dim3(MULLS_BLOCK), 0, st, descs, setup, bbox, stage, tmp_pos, tmp_nrm, spos, snrm, tpos,
					   tnrm, flag, match, wd, rp, grids, big_box);
	if (nbig_clouds)
	{
		hipLaunchKernelGGL(k_crop_big_count, dim3(nbig_segs), dim3(MULLS_BLOCK), 0, st, big_segs, descs, setup, bbox, stage, rp, seg_cnt, big_box);
		hipLaunchKernelGGL(k_crop_big_scan, dim3(nbig_clouds), dim3(64), 0, st, big_clouds, descs, rp, seg_cnt, big_box, grids);
		hipLaunchKernelGGL(k_crop_big_scatter, dim3(nbig_segs), dim3(MULLS_BLOCK), 0, st, big_segs, descs, setup, bbox, stage, rp, seg_cnt, tpos,
						   tnrm);
	}
}
void launch_thin(hipStream_t st, uint32_t npairs, CloudDesc *descs, const uint8_t *src_keep, const uint8_t *tgt_keep, float4 *spos, float4 *snrm,
				 float4 *tpos, float4 *tnrm)
{
	if (npairs)
		hipLaunchKernelGGL(k_thin, dim3(npairs * MULLS_NC * 2), dim3(MULLS_BLOCK), 0, st, descs, src_keep, tgt_keep, spos, snrm, tpos, tnrm);
}
void launch_grid_build(hipStream_t st, uint32_t npairs, uint32_t ntjobs, const Job *tjobs, const CloudDesc *descs, GridDesc *grids,
					   const RunParams &rp, const float4 *tpos, unsigned long long *bm, uint32_t *pf, uint32_t *cnt, uint32_t *cell_start,
					   float4 *tsorted, bool lds_tier)
{
	if (!ntjobs || !npairs)
		return;
	if (lds_tier)
	{
		hipLaunchKernelGGL(k_grid_build_sort, dim3(npairs * MULLS_NC), dim3(MULLS_LDS_BLOCK), 0, st, descs, grids, rp, tpos, cell_start, tsorted);
		return;
	}
	// global-memory tier: occupancy bitmap + ranks + counting sort by rank (cell_start holds the start positions)
	hipLaunchKernelGGL(k_bm_clear, dim3(npairs * MULLS_NC, npairs >= 64 ? 4 : 64), dim3(MULLS_BLOCK), 0, st, grids, rp, bm);
	hipLaunchKernelGGL(k_bm_mark, dim3(ntjobs), dim3(MULLS_BLOCK), 0, st, tjobs, descs, grids, tpos, bm);
	hipLaunchKernelGGL(k_bm_scan, dim3(npairs * MULLS_NC), dim3(1024), 0, st, grids, rp, bm, pf);
	hipLaunchKernelGGL(k_bm_count, dim3(ntjobs), dim3(MULLS_BLOCK), 0, st, tjobs, descs, grids, tpos, bm, pf, cnt);
	hipLaunchKernelGGL(k_bm_starts, dim3(npairs * MULLS_NC), dim3(1024), 0, st, descs, grids, rp, cnt, cell_start);
	hipLaunchKernelGGL(k_bm_scatter, dim3(ntjobs), dim3(MULLS_BLOCK), 0, st, tjobs, descs, grids, tpos, bm, pf, cnt, cell_start, tsorted);
}
size_t nn_lds_bytes(uint32_t cap, uint32_t maxcells, bool dedup)
{
	// query block, cost-sort tables, position records + index, cell table, and (lds_dedup) the on-chip duplicate table
	return (size_t)MULLS_LDS_QCHUNK * 16u + (size_t)MULLS_LDS_AUX + (size_t)cap * 14u + (((size_t)maxcells + 8u) & ~(size_t)1) * 2u + (dedup ? (size_t)cap * 4u : 0u);
}
int launch_nn_lds(hipStream_t st, uint32_t njobs, const Job *jobs, CloudDesc *descs, const PairState *states, const RunParams &rp, float4 *spos,
				  float4 *snrm, const GridDesc *grids, const uint32_t *cell_start, const float4 *tsorted, uint8_t *flag, int32_t *nn_idx,
				  float *nn_d2, unsigned long long *winner, const float4 *tnrm, int32_t *match, float *wd, const float4 *tpos, int32_t *nn_hint, float4 *mq, uint32_t cap, uint32_t maxcells)
{
	static bool attr_set = false;
	if (!attr_set)
	{
		if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_nn_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
			return -1;
		attr_set = true;
	}
	if (njobs)
		hipLaunchKernelGGL(k_nn_lds, dim3(njobs), dim3(MULLS_LDS_BLOCK), nn_lds_bytes(cap, maxcells, rp.lds_dedup != 0u), st, jobs, descs, states, rp, spos, snrm, grids,
						   cell_start, tsorted, flag, nn_idx, nn_d2, winner, tnrm, match, wd, tpos, nn_hint, mq, cap);
	return 0;
}
void launch_nn_grid(hipStream_t st, uint32_t njobs, const Job *jobs, CloudDesc *descs, const PairState *states, const RunParams &rp,
					float4 *spos, float4 *snrm, const GridDesc *grids, const unsigned long long *bm, const uint32_t *pf, const uint32_t *cs,
					const float4 *tsorted, const uint8_t *flag, int32_t *nn_idx, float *nn_d2, unsigned long long *winner)
{
	if (!njobs)
		return;
	// few jobs (one scan against a big map): several workgroups share a 512-query job so that the chip is not left idle
	uint32_t split = 1;
	while (split < 16 && njobs * split < 1024u)
		split <<= 1;
	hipLaunchKernelGGL(k_nn_grid, dim3(njobs * split), dim3(MULLS_BLOCK), 0, st, jobs, descs, states, rp, spos, snrm, grids, bm, pf, cs, tsorted,
					   flag, nn_idx, nn_d2, winner, split);
}
void launch_nn(hipStream_t st, uint32_t njobs, const Job *jobs, CloudDesc *descs, const PairState *states, const RunParams &rp, float4 *spos,
			   float4 *snrm, const float4 *tpos, const uint8_t *flag, int32_t *nn_idx, float *nn_d2, unsigned long long *winner)
{
	if (njobs)
		hipLaunchKernelGGL(k_nn, dim3(njobs), dim3(MULLS_NN_BLOCK), 0, st, jobs, descs, states, rp, spos, snrm, tpos, flag, nn_idx, nn_d2, winner);
}
void launch_nn_shoot(hipStream_t st, uint32_t njobs, const Job *jobs, CloudDesc *descs, const PairState *states, const RunParams &rp,
					 float4 *spos, float4 *snrm, const float4 *tpos, const uint8_t *flag, int32_t *nn_idx, float *nn_d2, unsigned long long *winner)
{
	if (njobs)
		hipLaunchKernelGGL(k_nn_shoot, dim3(njobs), dim3(MULLS_BLOCK), 0, st, jobs, descs, states, rp, spos, snrm, tpos, flag, nn_idx, nn_d2,
						   winner);
}
void launch_filter(hipStream_t st, uint32_t njobs, const Job *jobs, CloudDesc *descs, const PairState *states, const RunParams &rp,
				   const float4 *snrm, const float4 *tnrm, uint8_t *flag, const int32_t *nn_idx, const float *nn_d2, int32_t *match, float *wd,
				   const unsigned long long *winner, const float4 *tpos, float4 *mq)
{
	if (njobs)
		hipLaunchKernelGGL(k_filter, dim3(njobs), dim3(MULLS_BLOCK), 0, st, jobs, descs, states, rp, snrm, tnrm, flag, nn_idx, nn_d2, match, wd,
						   winner, tpos, mq);
}
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
	hipLaunchKernelGGL(k_finish, dim3(npairs), dim3(MULLS_BLOCK), 0, st, descs, states, rp, partial, out, bbox, pair_base);
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
