// k_setup.hip — once per registration: clone + initial guess, intersection crop, keep-less thinning
// (gfx950 / CDNA4, wave64; numerics policy and launch geometry: device_util.h)
#include "device_util.h"
#include "crop_grid.h"
#include "detmath.h"

// ---------------------------------------------------------------------------------------------------------------
// Setup 1: clone the staged source clouds into SoA and apply the initial guess (double math, float store); reduce
// the bounding box of the transformed ground / pillar / facade source clouds (cregistration.hpp:2912-2915).
__global__ __launch_bounds__(MULLS_BLOCK) void k_clone_src(const Job *__restrict__ jobs, const CloudDesc *__restrict__ descs,
															const PairSetup *__restrict__ setup, const float4 *__restrict__ stage,
															float4 *__restrict__ tmp_pos, float4 *__restrict__ tmp_nrm,
															uint32_t *__restrict__ bbox /* [pair][6] ordered keys */, RunParams rp)
{
	__shared__ uint32_t box_red[MULLS_BLOCK / 64][6];
	const Job job = jobs[blockIdx.x];
	const CloudDesc &d = descs[job.pair * MULLS_NC + job.cls];
	// motion undistortion regenerates the five non-vertex clouds from block2->pc_*_down (cregistration.hpp:1251-1253)
	const bool regen = rp.undistort && job.cls != 5;
	const uint32_t n_in = regen ? d.sd_n0 : d.src_n0;
	const PairSetup &su = setup[job.pair];
	const double *G = su.guess;
	uint32_t k[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
	// job.count consecutive 256-point chunks (0 = one): a dense scan's clouds take eight per workgroup, so that the box below costs a few hundred atomics on
	// the pair's six words instead of tens of thousands (250 us of a 236 k-point pair's setup, profiles/r04_large_base.txt)
	const uint32_t reps = job.count ? job.count : 1u;
	for (uint32_t rep_chunk = 0; rep_chunk < reps; rep_chunk++)
	{
		const uint32_t s = job.start + rep_chunk * MULLS_BLOCK + threadIdx.x;
		if (s >= n_in)
			continue;
		float x = 0, y = 0, z = 0;
		float4 a, b; // a = x y z intensity, b = nx ny nz curvature
		if (regen)
			load_staged(stage, d.sd_stage, (d.stage_fmt >> 4) & 3u, d.sd_n0, s, a, b);
		else
			load_staged(stage, d.src_stage, d.stage_fmt & 3u, d.src_n0, s, a, b);
		if (regen)
		{
			// CFilter::apply_motion_compensation(in, out, inverse(initial_guess)) (cfilter.hpp:493-516): the point is moved
			// by the fraction `curvature` (its time stamp in [0,1]) of the inverse guess — slerp from the identity
			// quaternion, linear translation — in double, stored as float; normals are copied unrotated.
			const float sc = b.w;
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
					const double theta = su.inv_t[3], sinTheta = mulls::det::sin_cr(theta); // (theta: the host's acos of |q.w|, batch_fill)
					s0 = mulls::det::sin_cr((1.0 - t) * theta) / sinTheta;
					s1 = mulls::det::sin_cr((t * theta)) / sinTheta;
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
		const int reps_g = (rp.undistort && job.cls == 5) ? 2 : 1; // the vertex cloud is not regenerated: it receives the guess twice
		float onx = b.x, ony = b.y, onz = b.z;						 // (cregistration.hpp:1183 and :1257)
		x = a.x, y = a.y, z = a.z;
		for (int rep = 0; rep < reps_g; rep++)
		{
			const double px = x, py = y, pz = z, nx = onx, ny = ony, nz = onz;
			x = (float)(G[0] * px + G[1] * py + G[2] * pz + G[3]);
			y = (float)(G[4] * px + G[5] * py + G[6] * pz + G[7]);
			z = (float)(G[8] * px + G[9] * py + G[10] * pz + G[11]);
			onx = (float)(G[0] * nx + G[1] * ny + G[2] * nz);
			ony = (float)(G[4] * nx + G[5] * ny + G[6] * nz);
			onz = (float)(G[8] * nx + G[9] * ny + G[10] * nz);
		}
		tmp_pos[d.src_off + s] = make_float4(x, y, z, a.w);
		tmp_nrm[d.src_off + s] = make_float4(onx, ony, onz, b.w);
		k[0] = min(k[0], f2ord(x)), k[1] = min(k[1], f2ord(y)), k[2] = min(k[2], f2ord(z));
		k[3] = max(k[3], f2ord(x)), k[4] = max(k[4], f2ord(y)), k[5] = max(k[5], f2ord(z));
	}
	if (job.cls == 0 || job.cls == 1 || job.cls == 2)
	{
		// the bounding box of the transformed ground / pillar / facade source clouds (cregistration.hpp:2912-2915): wave, workgroup, then six atomics
		for (int off = 32; off > 0; off >>= 1)
			for (int j = 0; j < 3; j++)
			{
				k[j] = min(k[j], (uint32_t)__shfl_down(k[j], off));
				k[3 + j] = max(k[3 + j], (uint32_t)__shfl_down(k[3 + j], off));
			}
		if ((threadIdx.x & 63) == 0)
			for (int j = 0; j < 6; j++)
				box_red[threadIdx.x >> 6][j] = k[j];
		__syncthreads();
		if (threadIdx.x < 6)
		{
			uint32_t v = box_red[0][threadIdx.x];
			for (int w = 1; w < MULLS_BLOCK / 64; w++)
				v = threadIdx.x < 3 ? min(v, box_red[w][threadIdx.x]) : max(v, box_red[w][threadIdx.x]);
			if (threadIdx.x < 3 && v != 0xffffffffu)
				atomicMin(&bbox[job.pair * 6 + threadIdx.x], v);
			else if (threadIdx.x >= 3 && v != 0u)
				atomicMax(&bbox[job.pair * 6 + threadIdx.x], v);
		}
	}
}

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
	if (side && rp.tgt_map && d.tier == MULLS_TIER_LDS)
		return; // k_tgt_grid crops the LDS tier's target clouds (no working copy: rank -> staged index map) and builds their grids in one pass
	if (side && d.big_slot)
	{
		// cropped by k_crop_big_* (many workgroups); this one only arms the cloud's bounding-box keys
		if (threadIdx.x < 6)
			big_box[(d.big_slot - 1u) * 6u + threadIdx.x] = threadIdx.x < 3 ? 0xffffffffu : 0u;
		return;
	}
	if (!side && d.src_big_slot)
		return; // likewise
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
				load_staged(stage, d.tgt_stage, (d.stage_fmt >> 2) & 3u, d.tgt_n0, i, p, q);
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
				identity_step(p, q); // iteration 0's rigid step (TempTran = I), so that its search may start from these records (device_util.h)
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
			grids[pair * MULLS_NC + cls] = make_grid(lo3, hi3, running, rp, d, cls);
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
																 const float4 *__restrict__ stage, const float4 *__restrict__ tmp_pos, RunParams rp,
																 uint32_t *__restrict__ seg_cnt, uint32_t *__restrict__ seg_box)
{
	__shared__ uint32_t red4[4];
	const Job sg = segs[blockIdx.x]; // count = big slot
	const uint32_t cls = sg.cls & 0xffu;
	const bool src_side = (sg.cls & MULLS_BIG_SRC_SIDE) != 0u;
	const CloudDesc &d = descs[sg.pair * MULLS_NC + cls];
	const uint32_t n0 = src_side ? ((rp.undistort && cls != 5u) ? d.sd_n0 : d.src_n0) : d.tgt_n0;
	double lo[3], hi[3];
	if (rp.crop)
		crop_box(sg.pair, bbox, setup, lo, hi);
	uint32_t mine = 0;
	float bmin[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, bmax[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
	for (uint32_t k = 0; k < MULLS_SEG; k += MULLS_BLOCK)
	{
		const uint32_t i = sg.start + k + threadIdx.x;
		if (i < n0)
		{
			const float4 p = src_side ? tmp_pos[d.src_off + i] : load_staged_pos(stage, d.tgt_stage, (d.stage_fmt >> 2) & 3u, i);
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
	// the segment's box (ordered keys; 0xffffffff / 0 = nothing) into ITS slot of seg_box — k_crop_big_scan takes the minimum / maximum over a cloud's segments.
	// (They were atomics on the cloud's six words: a 1 M-point map has 235 segments of four waves each, and atomics on one address are served one after the other.)
	__shared__ uint32_t box4[4][6];
	for (int k = 0; k < 3; k++)
	{
		for (int off = 32; off > 0; off >>= 1)
		{
			bmin[k] = fminf(bmin[k], __shfl_down(bmin[k], off));
			bmax[k] = fmaxf(bmax[k], __shfl_down(bmax[k], off));
		}
		if ((threadIdx.x & 63) == 0)
		{
			const bool any = bmin[k] <= bmax[k];
			box4[threadIdx.x >> 6][k] = any ? f2ord(bmin[k]) : 0xffffffffu;
			box4[threadIdx.x >> 6][3 + k] = any ? f2ord(bmax[k]) : 0u;
		}
	}
	const uint32_t total = block_sum_u32(mine, red4); // (barriers inside: box4 is complete behind it)
	if (threadIdx.x == 0)
		seg_cnt[blockIdx.x] = total;
	if (threadIdx.x < 6 && !src_side) // (the box sizes a target's grid)
	{
		const uint32_t k = threadIdx.x;
		const uint32_t a = box4[0][k], b = box4[1][k], c = box4[2][k], e = box4[3][k];
		seg_box[(size_t)blockIdx.x * 6u + k] = k < 3 ? min(min(a, b), min(c, e)) : max(max(a, b), max(c, e));
	}
}

// one wave per big cloud: segment counts -> segment bases, cloud size, grid descriptor
__global__ __launch_bounds__(64) void k_crop_big_scan(const Job *__restrict__ clouds, CloudDesc *__restrict__ descs, RunParams rp,
													   uint32_t *__restrict__ seg_cnt, const uint32_t *__restrict__ seg_box,
													   GridDesc *__restrict__ grids)
{
	const Job bc = clouds[blockIdx.x]; // start = first segment, count = number of segments
	const bool want_box = !(bc.cls & MULLS_BIG_SRC_SIDE) && grids != nullptr;
	uint32_t running = 0;
	uint32_t box[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u}; // ordered keys of the cloud's box: this lane's segments
	for (uint32_t base = 0; base < bc.count; base += 64)
	{
		const uint32_t s = base + threadIdx.x;
		if (want_box && s < bc.count)
			for (int k = 0; k < 6; k++)
			{
				const uint32_t b = seg_box[(size_t)(bc.start + s) * 6u + k];
				box[k] = k < 3 ? min(box[k], b) : max(box[k], b);
			}
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
	if (want_box)
		for (int k = 0; k < 6; k++)
			for (int off = 32; off > 0; off >>= 1)
			{
				const uint32_t o = (uint32_t)__shfl_down((int)box[k], off);
				box[k] = k < 3 ? min(box[k], o) : max(box[k], o);
			}
	if (threadIdx.x == 0 && (bc.cls & MULLS_BIG_SRC_SIDE))
	{
		CloudDesc &d = descs[bc.pair * MULLS_NC + (bc.cls & 0xffu)];
		d.src_n = running;
		d.alive_cur = running;
		d.alive_next = 0;
		d.n_matched = 0;
		d.valid_next = 0;
		d.n_valid = 0;
	}
	else if (threadIdx.x == 0)
	{
		descs[bc.pair * MULLS_NC + bc.cls].tgt_n = running;
		if (grids)
		{
			float lo3[3], hi3[3];
			for (int k = 0; k < 3; k++)
			{
				lo3[k] = running ? ord2f(box[k]) : __builtin_inff();
				hi3[k] = running ? ord2f(box[3 + k]) : -__builtin_inff();
			}
			grids[bc.pair * MULLS_NC + bc.cls] = make_grid(lo3, hi3, running, rp, descs[bc.pair * MULLS_NC + bc.cls], bc.cls);
		}
	}
}

__global__ __launch_bounds__(MULLS_BLOCK) void k_crop_big_scatter(const Job *__restrict__ segs, const CloudDesc *__restrict__ descs,
																   const PairSetup *__restrict__ setup, const uint32_t *__restrict__ bbox,
																   const float4 *__restrict__ stage, RunParams rp,
																   const uint32_t *__restrict__ seg_base, float4 *__restrict__ tpos,
																   float4 *__restrict__ tnrm, const float4 *__restrict__ tmp_pos, const float4 *__restrict__ tmp_nrm,
																   float4 *__restrict__ spos, float4 *__restrict__ snrm, uint8_t *__restrict__ flag,
																   int32_t *__restrict__ match, float *__restrict__ wd)
{
	__shared__ uint32_t wave_cnt[4];
	const Job sg = segs[blockIdx.x];
	const uint32_t cls = sg.cls & 0xffu;
	const bool src_side = (sg.cls & MULLS_BIG_SRC_SIDE) != 0u;
	const CloudDesc &d = descs[sg.pair * MULLS_NC + cls];
	const uint32_t n0 = src_side ? ((rp.undistort && cls != 5u) ? d.sd_n0 : d.src_n0) : d.tgt_n0;
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
		if (i < n0)
		{
			if (src_side)
				p = tmp_pos[d.src_off + i], q = tmp_nrm[d.src_off + i];
			else
				load_staged(stage, d.tgt_stage, (d.stage_fmt >> 2) & 3u, d.tgt_n0, i, p, q);
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
		if (keep && src_side)
		{
			const uint32_t dst = d.src_off + running + wbase + before;
			identity_step(p, q); // (as k_crop)
			spos[dst] = p;
			snrm[dst] = q;
			flag[dst] = MULLS_F_ALIVE;
			match[dst] = -1;
			wd[dst] = 0.0f;
		}
		else if (keep)
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
// CFilter::apply_motion_compensation(pc_in_out, Tran, s_ambigous_thre) (cfilter.hpp:470-491) on a cloud of 48-byte records in device memory: the point
// with time stamp t = curvature in [thre, 1 - thre] moves by the fraction t of Tran — slerp from the identity quaternion (Eigen's
// QuaternionBase::slerp), linear translation — in double, stored as float; directions, intensities and time stamps stay.  What mulls_slam applies to a
// frame's clouds after its registration (test/mulls_slam.cpp:703-712); k_clone_src does the same inside a registration (cregistration.hpp:1251-1253).
struct MotionComp
{
	double q[4]; // Eigen::Quaterniond(Tran.block<3,3>(0,0)): w x y z
	double t[3]; // Tran.block<3,1>(0,3)
	double theta, sin_theta; // acos(|q.w|) — one value per transform, by the HOST's libm, the reference's own (launch_motion_comp) — and its sine (detmath.h)
	float thre;
};
__global__ __launch_bounds__(MULLS_BLOCK) void k_motion_comp(float4 *__restrict__ recs, uint32_t n, MotionComp M)
{
	const uint32_t i = blockIdx.x * MULLS_BLOCK + threadIdx.x;
	if (i >= n)
		return;
	float4 a = recs[(size_t)i * 3];
	const float sc = recs[(size_t)i * 3 + 2].y; // curvature
	if (sc < M.thre || (double)sc > 1.0 - M.thre)
		return;
	const double t = (double)sc, one = 1.0 - 2.220446049250313e-16;
	const double dq = M.q[0], absD = fabs(dq);
	double s0, s1;
	if (absD >= one)
	{
		s0 = 1.0 - t;
		s1 = t;
	}
	else
	{
		// (the two sines per point by detmath.h's correctly rounded sine: the same bits on every ROCm version and on the host, like the rest of the library's
		// trigonometry; the device library's acos / sin were the one place where a result depended on the toolchain — advisor, round 4)
		const double theta = M.theta, sinTheta = M.sin_theta;
		s0 = mulls::det::sin_cr((1.0 - t) * theta) / sinTheta;
		s1 = mulls::det::sin_cr((t * theta)) / sinTheta;
	}
	if (dq < 0)
		s1 = -s1;
	const double qw = s0 + s1 * M.q[0], qx = s1 * M.q[1], qy = s1 * M.q[2], qz = s1 * M.q[3];
	const double vx = a.x, vy = a.y, vz = a.z;
	const double uvx = 2.0 * (qy * vz - qz * vy), uvy = 2.0 * (qz * vx - qx * vz), uvz = 2.0 * (qx * vy - qy * vx);
	const double rx = vx + qw * uvx + (qy * uvz - qz * uvy);
	const double ry = vy + qw * uvy + (qz * uvx - qx * uvz);
	const double rz = vz + qw * uvz + (qx * uvy - qy * uvx);
	a.x = (float)(rx + t * M.t[0]);
	a.y = (float)(ry + t * M.t[1]);
	a.z = (float)(rz + t * M.t[2]);
	recs[(size_t)i * 3] = a;
}

// ---------------------------------------------------------------------------------------------------------------
// blockIdx.y: the segment; blockIdx.x: its 16-KiB chunk (256 lanes x 4 x 16 B).  A source in the host-mapped mailbox is read over PCIe by the loads themselves.
__global__ __launch_bounds__(256) void k_copy_segs(CopyArgs a)
{
	const CopySeg s = a.seg[blockIdx.y];
	const uint32_t w16 = s.bytes >> 4, first = blockIdx.x * 1024u;
	if (first >= w16 + 1u)
		return;
	const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(s.src);
	uint4 *__restrict__ dst = reinterpret_cast<uint4 *>(s.dst);
	uint4 v[4];
#pragma unroll
	for (int k = 0; k < 4; k++)
	{
		const uint32_t w = first + threadIdx.x + 256u * (uint32_t)k;
		if (w < w16)
			v[k] = src[w];
	}
#pragma unroll
	for (int k = 0; k < 4; k++)
	{
		const uint32_t w = first + threadIdx.x + 256u * (uint32_t)k;
		if (w < w16)
			dst[w] = v[k];
	}
	if (blockIdx.x == 0 && threadIdx.x < ((s.bytes & 15u) >> 2)) // the tail: up to three 4-byte words
		reinterpret_cast<uint32_t *>(s.dst)[(size_t)w16 * 4u + threadIdx.x] = reinterpret_cast<const uint32_t *>(s.src)[(size_t)w16 * 4u + threadIdx.x];
}

// host-callable launch wrappers (the driver is plain C++ and never sees <<< >>>)
#include "launch.h"

void launch_motion_comp(hipStream_t st, float4 *recs, uint32_t n, const double q[4], const double t[3], float thre)
{
	if (!n)
		return;
	MotionComp M;
	for (int k = 0; k < 4; k++)
		M.q[k] = q[k];
	for (int k = 0; k < 3; k++)
		M.t[k] = t[k];
	M.thre = thre;
	M.theta = std::acos(std::fabs(q[0]) < 1.0 ? std::fabs(q[0]) : 1.0);
	M.sin_theta = mulls::det::sin_cr(M.theta);
	hipLaunchKernelGGL(k_motion_comp, dim3((n + MULLS_BLOCK - 1) / MULLS_BLOCK), dim3(MULLS_BLOCK), 0, st, recs, n, M);
}

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
		hipLaunchKernelGGL(k_crop_big_count, dim3(nbig_segs), dim3(MULLS_BLOCK), 0, st, big_segs, descs, setup, bbox, stage, tmp_pos, rp, seg_cnt, big_box + (size_t)nbig_clouds * 6u);
		hipLaunchKernelGGL(k_crop_big_scan, dim3(nbig_clouds), dim3(64), 0, st, big_clouds, descs, rp, seg_cnt, big_box + (size_t)nbig_clouds * 6u, grids);
		hipLaunchKernelGGL(k_crop_big_scatter, dim3(nbig_segs), dim3(MULLS_BLOCK), 0, st, big_segs, descs, setup, bbox, stage, rp, seg_cnt, tpos,
						   tnrm, tmp_pos, tmp_nrm, spos, snrm, flag, match, wd);
	}
}

void launch_thin(hipStream_t st, uint32_t npairs, CloudDesc *descs, const uint8_t *src_keep, const uint8_t *tgt_keep, float4 *spos, float4 *snrm,
				 float4 *tpos, float4 *tnrm)
{
	if (npairs)
		hipLaunchKernelGGL(k_thin, dim3(npairs * MULLS_NC * 2), dim3(MULLS_BLOCK), 0, st, descs, src_keep, tgt_keep, spos, snrm, tpos, tnrm);
}

void launch_copy_segs(hipStream_t st, const CopySeg *segs, uint32_t n)
{
	for (uint32_t i0 = 0; i0 < n; i0 += MULLS_COPY_SEGS)
	{
		CopyArgs a;
		const uint32_t m = n - i0 < (uint32_t)MULLS_COPY_SEGS ? n - i0 : (uint32_t)MULLS_COPY_SEGS;
		uint32_t most = 0;
		for (uint32_t k = 0; k < (uint32_t)MULLS_COPY_SEGS; k++)
		{
			a.seg[k] = k < m ? segs[i0 + k] : CopySeg{0ull, 0ull, 0u, 0u};
			most = a.seg[k].bytes > most ? a.seg[k].bytes : most;
		}
		hipLaunchKernelGGL(k_copy_segs, dim3(((most >> 4) + 1024u) / 1024u, m), dim3(256), 0, st, a);
	}
}
